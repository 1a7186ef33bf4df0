#!/bin/bash
# Same-box pair of steady-state kernel summaries of the SD img2img trajectory with the eight-phase GEMM off / on (PD_LIN_P8 is read by the
# library at every dispatch; the variable is exported BEFORE rocprofv3 starts python3, so the program after `--` is python3 itself).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/ab_p8_insitu
mkdir -p $OUT
for v in 0 1 0 1; do
  tag=p8_${v}_$(ls $OUT | grep -c "p8_${v}_.*stats.csv" || true)
  export PD_LIN_P8_FORCE=$v
  if [ $v = 0 ]; then export PD_LIN_P8=0; else unset PD_LIN_P8; fi
  d=/tmp/prof_ab_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --workload sd_img2img --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-workloads > $OUT/$tag.log 2>$OUT/$tag.err
  python3 scripts/steady_stats.py "$(ls $d/*/*_kernel_trace.csv | head -1)" 2 1 $OUT/${tag}_stats.csv 2> $OUT/${tag}.note
  python3 -c "import json,sys; d=json.loads(open('$OUT/$tag.log').read().strip().splitlines()[-1]); print('$tag', d['value'], d['ms_per_step'])"
  rm -rf $d
done
