#!/usr/bin/env python3
"""Same-box A/B of several builds of the library on the WHOLE workload (bench.py, hipGraph replay: what the driver times) -- per-layer
event-fenced profiles miss the chip-level coupling (a faster kernel raises the power drawn, the clock of every other kernel falls).
    python scripts/ab_bench.py build_ab/a.so build_ab/b.so ... [--rounds 2] [--workload img2img] [bench.py args...]"""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 2
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
libs = [a for a in args if a.endswith(".so")]
rest = [a for a in args if not a.endswith(".so")]
if "--steps" not in rest: rest += ["--steps", "3"]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, PD_ALLOW_ABI_MISMATCH="1", PD_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--warmup", "1", "--no-side-workloads", "--no-cpu-baseline", "--no-roofline",
                              "--no-sweep"] + rest, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append((d["value"], d["ms_per_step"]))
        except Exception:
            print(f"!! {l}: {out.stderr[-500:]}", flush=True)
    print(f"round {r}: " + "  ".join(f"{os.path.basename(l)[:-3]} {res[l][-1][0]:.3f}" for l in libs if res[l]), flush=True)
base = statistics.median(v for v, _ in res[libs[0]])
for l in libs:
    if res[l]:
        v = statistics.median(v for v, _ in res[l])
        print(f"{os.path.basename(l)[:-3]:12s} median {v:9.3f}  ({v / base:.4f} x {os.path.basename(libs[0])[:-3]})   all: {[round(x, 3) for x, _ in res[l]]}")
