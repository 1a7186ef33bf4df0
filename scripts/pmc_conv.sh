#!/bin/bash
# PMC passes over one pd_conv configuration (scripts/bench_conv.py): HBM traffic, L2 hit rate, SQ stall split.
#   bash scripts/pmc_conv.sh <tag> -- <bench_conv args>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=$1; shift; shift
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_${tag}_$i -- python3 scripts/bench_conv.py --iters 3 "$@" > $out/run_$i.log 2>&1
  f=$(ls /tmp/pmc_${tag}_$i/*/*_counter_collection.csv | head -1)
  python3 - "$f" <<'PY' >> $out/summary.txt
import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "conv" in r["Kernel_Name"]:
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in per.items():
    print(k[:90], {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
done
cat $out/summary.txt
