#!/bin/bash
# PMC passes over pd_attn_d64 forward at the 64^2 SD shape (B 32, 5 heads, N 4096): which pipe binds.
# Separate --pmc passes with --kernel-trace only; the program after `--` is python3 itself.
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
MODE=${1:-fwd}        # fwd | bwd (the two backward kernels: dq, dkv)
ARGS=""; [ "$MODE" = bwd ] && ARGS="--bwd"
OUT=gpurun_out/pmc_attn_d64_$MODE
mkdir -p $OUT
python3 scripts/bench_attn_d64.py 32 $ARGS > $OUT/op_bench.log 2>&1; cat $OUT/op_bench.log
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_d64_$i -- python3 scripts/bench_attn_d64.py 32 --launches 6 $ARGS > /dev/null 2>$OUT/pass$i.err || { echo "pass $i failed"; tail -3 $OUT/pass$i.err; continue; }
  python3 - "$(ls /tmp/pmc_d64_$i/*/*_counter_collection.csv | head -1)" <<'PY' | tee -a $OUT/counters.txt
import collections, csv, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "attn_d64" in r["Kernel_Name"]:
        d[r["Kernel_Name"].split("<")[0].split("::")[-1]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in d.items():
    for n, v in sorted(c.items()):
        print(k, n, sum(v) / len(v), len(v))
PY
  rm -rf /tmp/pmc_d64_$i
done
