#!/bin/bash
# Round 6: LDS bank conflicts per kernel over one step of a workload (separate --pmc pass, --kernel-trace only; the program after `--` is python3).
#   bash scripts/pmc_lds_scan.sh <tag> <python args...>      e.g.  pmc_lds_scan.sh sd_train bench.py --workload sd_train --steps 1 --warmup 0 ...
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/pmc_lds_scan; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d /tmp/pmc_lds_$TAG -- python3 "$@" > /dev/null 2>$OUT/$TAG.err
python3 - "$(ls /tmp/pmc_lds_$TAG/*/*_counter_collection.csv | head -1)" <<'PY' | tee $OUT/$TAG.txt
import collections, csv, sys
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:110]
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": n[k] += 1
rows = sorted(d.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0))
print(f"{'kernel':110s} launches  conflict_cycles  lds_active_cycles  conflict/active")
for k, c in rows[:40]:
    a = c.get("SQ_LDS_IDX_ACTIVE", 0); b = c.get("SQ_LDS_BANK_CONFLICT", 0)
    if a > 0: print(f"{k:110s} {n[k]:6d} {b:16.0f} {a:18.0f} {b / a:8.3f}")
PY
rm -rf /tmp/pmc_lds_$TAG
