#!/bin/bash
# Same-box A/B of one library under a kernel-selecting environment switch (e.g. PD_CONV_PLAIN=0 vs default) on bench.py workloads.
#   bash scripts/ab_env.sh PD_CONV_PLAIN=0 sd_img2img 2 [more bench args]
set -e
SW=$1; WL=$2; ST=$3; shift 3
one() { python bench.py --workload $WL --steps $ST --warmup 1 --no-side-workloads --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"; }
for r in 1 2; do
  echo "$WL $*  $SW: $(env $SW bash -c "$(declare -f one); WL=$WL ST=$ST; one $*")   default: $(one "$@")"
done
