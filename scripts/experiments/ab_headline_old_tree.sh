for r in 1 2; do
  for w in old new; do
    if [ $w = old ]; then d=build_ab/old_tree; lib=$PWD/build_ab/rev_a95.so; else d=.; lib=$PWD/phendiff_amd/libphendiff_hip.so; fi
    v=$(cd $d && PD_LIB=$lib PD_ALLOW_ABI_MISMATCH=1 python3 bench.py --steps 3 --warmup 1 --no-side-workloads --no-cpu-baseline --no-roofline --no-sweep 2>/dev/null | python3 -c "import json,sys;print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "round $r $w $v"
  done
done
