#!/bin/bash
# same-box A/B of kernel-selecting environment switches on the whole workload: bash scripts/ab_envs.sh "<workload args>" ENV_A ENV_B ... (an ENV is
# a comma-separated list of VAR=value words or "-" for the default); ROUNDS rounds (default 4) in alternating order, STEPS timed steps each
WL="$1"; shift
ROUNDS=${ROUNDS:-4}; STEPS=${STEPS:-6}
one() { if [ "$1" = "-" ]; then E=""; else E=$(echo "$1" | tr ',' ' '); fi
  env $E python bench.py $WL --steps $STEPS --warmup 2 --no-side-workloads --no-cpu-baseline --no-roofline --no-sweep 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'])"; }
for r in $(seq 1 $ROUNDS); do
  if [ $((r % 2)) = 1 ]; then order=("$@"); else order=(); for ((i=$#; i>0; i--)); do order+=("${!i}"); done; fi
  for e in "${order[@]}"; do echo "round $r  $e  $(one "$e")"; done
done | tee /tmp/ab_envs.out
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("/tmp/ab_envs.out"):
    f = l.split()
    d[f[2]].append(float(f[3]))
base = None
for k, v in d.items():
    m = statistics.median(v)
    base = base or m
    print(f"{k:60s} median {m:8.3f}  x{m / base:.4f}  {v}")
PY
