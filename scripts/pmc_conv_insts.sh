#!/bin/bash
# Dynamic instruction mix of one pd_conv configuration (scripts/bench_conv.py) from SQ counters: instructions per wave by class,
# where the vector port's time goes.  Counters only (no trace domains beside --kernel-trace), one pass per set.
#   bash scripts/pmc_conv_insts.sh <tag> -- <bench_conv args>
cd "$(dirname "$0")/.."
ROOT=$(pwd)
export TMPDIR=/tmp
tag=$1; shift; shift
out=$ROOT/gpurun_out/pmci_$tag; mkdir -p $out; rm -f $out/summary.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmci_${tag}_$i -- python3 $ROOT/scripts/bench_conv.py --iters 3 "$@" > $out/run_$i.log 2>&1)
  f=$(ls /tmp/pmci_${tag}_$i/*/*_counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "set $i: no counter file (see $out/run_$i.log)" >> $out/summary.txt; continue; }
  python3 - "$f" <<'PY' >> $out/summary.txt
import collections, csv, sys
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "conv" in r["Kernel_Name"]:
        per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in per.items():
    print(k[:60], {n: round(sum(v) / len(v), 1) for n, v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
done
cat $out/summary.txt
