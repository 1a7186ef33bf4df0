#!/usr/bin/env python3
"""Same-box A/B of SEVERAL builds of libphendiff_hip.so on the per-layer forward profile: runs scripts/profile_forward.py under each
library in turn (child processes), `--rounds` times, and prints per-kind totals and the per-layer 3x3 table of the medians relative to
the first library.      python scripts/ab_multi.py build_ab/a.so build_ab/b.so ... [--rounds 3] [profile_forward args...]"""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 3
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
libs = [a for a in args if a.endswith(".so")]
rest = [a for a in args if not a.endswith(".so")]
def run(lib):
    env = dict(os.environ, PD_ALLOW_ABI_MISMATCH="1", PD_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profile_forward.py")] + rest, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    rows = []
    for l in out.stdout.splitlines():
        m = re.match(r"\s*(\d+)\s+(\S+)\s+(.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: rows.append((int(m.group(1)), m.group(2), m.group(3).strip(), float(m.group(4))))
    if not rows: print(f"!! {lib}: no rows; stderr tail: {out.stderr[-600:]}", flush=True)
    return rows
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        rows = run(l)
        if rows: res[l].append(rows)
    print(f"round {r} done", flush=True)
libs = [l for l in libs if res[l]]
n = len(res[libs[0]][0])
def med(l, i): return statistics.median(r[i][3] for r in res[l])
names = [os.path.basename(l)[:-3] for l in libs]
print(f"{'#':>3} {'kind':9} {'shape':44} " + " ".join(f"{x:>9}" for x in names))
kinds = {}
for i in range(n):
    k, shp = res[libs[0]][0][i][1], res[libs[0]][0][i][2]
    v = [med(l, i) for l in libs]
    kk = kinds.setdefault(k, [0.0] * len(libs))
    for j, x in enumerate(v): kk[j] += x
    if k.startswith("conv"):
        print(f"{i:3d} {k:9s} {shp:44s} {v[0]:9.3f} " + " ".join(f"{x / v[0]:9.3f}" for x in v[1:]))
print()
for k, v in kinds.items():
    print(f"{k:12s} " + " ".join(f"{x:9.3f}" for x in v) + "   | " + " ".join(f"{x / v[0]:.3f}" for x in v[1:]))
tot = [sum(v[j] for v in kinds.values()) for j in range(len(libs))]
print(f"{'total':12s} " + " ".join(f"{x:9.3f}" for x in tot) + "   | " + " ".join(f"{x / tot[0]:.3f}" for x in tot[1:]) + f"   ({rounds} rounds, medians; columns: {names})")
