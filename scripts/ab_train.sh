set -e
one() { python bench.py --workload $1 --steps $2 --warmup 2 --no-side-workloads --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"; }
for r in 1 2 3; do
  echo "train old: $(PD_LIB=build_ab/r3e.so PD_ALLOW_ABI_MISMATCH=1 one train 5)   new: $(one train 5)"
done
for r in 1 2; do
  echo "sd_train old: $(PD_LIB=build_ab/r3e.so PD_ALLOW_ABI_MISMATCH=1 one sd_train 3)   new: $(one sd_train 3)"
done
