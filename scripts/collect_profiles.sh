#!/bin/bash
# Round-end evidence on the GPU box: rocprofv3 kernel-trace summaries + bench lines, copied to gpurun_out/profiles_new/
# (then committed under profiles/).  Targets: head | train | sd | pmc_side.  The PMC summaries carry the hash of the kernel sources they were measured on; bench.py quotes
# them only while that hash matches the tree.  Usage: bash scripts/collect_profiles.sh head|train|sd
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
R=${R:-r5}          # round tag of the files written (profiles/${R}_*)
OUT=gpurun_out/profiles_new
mkdir -p $OUT
# Steady state only (round 5): every collection runs BATCHES = warm-up + timed steps of the workload and the summary is computed from
# the kernel trace with the first DISCARD batches cut off (scripts/steady_stats.py) -- rocprofv3's own --stats file, which averages
# over the cold first batches too, is kept beside it as *_whole_run.csv.
stats() {   # <tag> <log name> <batches> <discard> -- bench args (whose --warmup + --steps == batches)
  local tag=$1 log=$2 batches=$3 discard=$4; shift 4
  local d=/tmp/prof_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py "$@" --no-box > $OUT/$log 2>$OUT/$log.err
  cp "$(ls $d/*/*_kernel_stats.csv | head -1)" $OUT/${R}_${tag}_kernel_stats_whole_run.csv
  python3 scripts/steady_stats.py "$(ls $d/*/*_kernel_trace.csv | head -1)" $batches $discard $OUT/${R}_${tag}_kernel_stats.csv 2> $OUT/${R}_${tag}_kernel_stats.note
  cat $OUT/${R}_${tag}_kernel_stats.note
  echo "== $tag: $(tail -c 300 $OUT/$log | head -c 300)"
}
case "$1" in
head)
  # (6 replays of the 10 700-node graph queued back to back crashed rocprofv3 7.2 -- SIGSEGV in the tool 9 s in; 4 replays hold)
  stats b32 ${R}_bench_under_rocprof_b32.log ${HEAD_BATCHES:-4} ${HEAD_DISCARD:-2} --steps $(( ${HEAD_BATCHES:-4} - ${HEAD_DISCARD:-2} )) --warmup ${HEAD_DISCARD:-2} --batch 32 --no-cpu-baseline --no-roofline --no-sweep --no-side-workloads
  for x in kernel_stats.csv kernel_stats_whole_run.csv kernel_stats.note; do mv $OUT/${R}_b32_$x $OUT/${R}_kernel_stats_b32${x#kernel_stats}; done
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --batch 32 --inference-steps 2 --no-cpu-baseline --no-roofline --no-sweep --no-side-workloads --no-box > /dev/null 2>$OUT/pmc_$c.err
  done
  python3 scripts/collect_traffic.py $R "$(ls /tmp/pmc_FETCH_SIZE/*/*_counter_collection.csv | head -1)" "$(ls /tmp/pmc_WRITE_SIZE/*/*_counter_collection.csv | head -1)" > $OUT/traffic.txt
  cp profiles/${R}_hbm_traffic.json $OUT/
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_busy -- python3 bench.py --steps 1 --warmup 0 --batch 32 --inference-steps 2 --no-cpu-baseline --no-roofline --no-sweep --no-side-workloads --no-box > /dev/null 2>$OUT/pmc_busy.err
  python3 scripts/collect_mfma_busy.py $R "$(ls /tmp/pmc_busy/*/*_counter_collection.csv | head -1)" > $OUT/mfma_busy.txt
  cp profiles/${R}_mfma_busy.json $OUT/
  # the default line last: it quotes the two PMC summaries just written (same kernel sources by construction)
  python3 bench.py > $OUT/${R}_bench_default.json 2>$OUT/bench_default.err
  cut -c1-200 $OUT/${R}_bench_default.json
  python3 bench.py --model small_denoiser_config --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-sweep --no-side-workloads > $OUT/${R}_bench_small_denoiser_b16.json 2>/dev/null
  ;;
train)
  python3 bench.py --workload train --steps 10 --warmup 2 > $OUT/${R}_bench_train.json 2>/dev/null
  cut -c1-200 $OUT/${R}_bench_train.json
  stats train_b112 ${R}_train_bench_under_rocprof_b112.log 40 10 --workload train --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-sweep --no-side-workloads
  mv $OUT/${R}_train_b112_kernel_stats.csv $OUT/${R}_train_kernel_stats_b112.csv
  ;;
sd)
  python3 bench.py --workload sd_img2img --steps 2 > $OUT/${R}_bench_sd_img2img.json 2>/dev/null
  cut -c1-200 $OUT/${R}_bench_sd_img2img.json
  python3 bench.py --workload sd_train --steps 10 --warmup 2 > $OUT/${R}_bench_sd_train.json 2>/dev/null
  cut -c1-200 $OUT/${R}_bench_sd_train.json
  # (rocprofv3 7.2 crashes once more than ~40 000 graph launches are queued between two host syncs: ONE 36 000-node replay per sync)
  stats sd_img2img_b32 ${R}_sd_img2img_bench_under_rocprof_b32.log 2 1 --workload sd_img2img --steps 1 --warmup 1 --no-roofline --no-cpu-baseline --no-side-workloads
  mv $OUT/${R}_sd_img2img_b32_kernel_stats.csv $OUT/${R}_sd_img2img_kernel_stats_b32.csv
  stats sd_train_b32 ${R}_sd_train_bench_under_rocprof_b32.log 12 4 --workload sd_train --steps 8 --warmup 4 --no-roofline --no-cpu-baseline --no-side-workloads
  mv $OUT/${R}_sd_train_b32_kernel_stats.csv $OUT/${R}_sd_train_kernel_stats_b32.csv
  ;;
pmc_side)
  # round 6: HBM traffic of the side workloads -- one step each under --pmc FETCH_SIZE / WRITE_SIZE (separate passes;
  # the program after `--` is python3 itself), summed per kernel group (scripts/collect_traffic_step.py) -> profiles/${R}_hbm_traffic_<workload>.json
  for wl in train sd_train sd_img2img; do
    # train / sd_train: ONE optimisation step (= one pass of the plan's forward + backward ops).  sd_img2img: the roofline leg's unit is one
    # SD-UNet forward (the trajectory also runs the VAE through the same conv kernels), so the counters run over two bare UNet forwards
    if [ $wl = sd_img2img ]; then cmd="scripts/bench_sd_unet.py 32"; passes=2; export PD_PMC_FORWARDS_ONLY=1
    else cmd="bench.py --workload $wl --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-side-workloads --no-box"; passes=1; unset PD_PMC_FORWARDS_ONLY; fi
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmcs_${wl}_$c -- python3 $cmd > /dev/null 2>$OUT/pmc_${wl}_$c.err
    done
    python3 scripts/collect_traffic_step.py $R $wl "$(ls /tmp/pmcs_${wl}_FETCH_SIZE/*/*_counter_collection.csv | head -1)" "$(ls /tmp/pmcs_${wl}_WRITE_SIZE/*/*_counter_collection.csv | head -1)" "$cmd" $passes > $OUT/traffic_$wl.txt
    cp profiles/${R}_hbm_traffic_$wl.json $OUT/
    cat $OUT/traffic_$wl.txt
    rm -rf /tmp/pmcs_${wl}_FETCH_SIZE /tmp/pmcs_${wl}_WRITE_SIZE
  done
  unset PD_PMC_FORWARDS_ONLY
  ;;
esac
ls -la $OUT
