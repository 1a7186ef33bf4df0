#!/bin/bash
# power / clock samples (rocm-smi, every ~0.5 s) while the headline workload runs: is the chip at its power cap?   bash scripts/power_probe.sh [ENV=val ...]
env "$@" python bench.py --steps 10 --warmup 2 --no-side-workloads --no-cpu-baseline --no-roofline --no-sweep > /tmp/pp_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | python -c "
import json,sys,re
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    sclk=int(re.sub(r'\D','',c.get('sclk clock speed:','0')) or 0)
    if sclk > 500: print(sclk, 'MHz', c.get('Current Socket Graphics Package Power (W)'), 'W', c.get('Temperature (Sensor junction) (C)'), 'C')
except Exception as e: pass
"
  sleep 0.4
done
python -c "
import json
d=json.loads(open('/tmp/pp_bench.log').read().strip().splitlines()[-1]); print('images/s', d['value'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max" | head -2
