#!/usr/bin/env python3
"""Same-box A/B of two builds of libphendiff_hip.so on the per-layer forward profile (boxes differ by +-3..5 %, more than most kernel
changes): runs scripts/profile_forward.py alternately under each library (child processes), prints per-kind totals and the
per-layer table of the medians.
    python scripts/ab_forward.py build_ab/old.so [new.so (default: the in-tree library)] [--rounds 3] [profile_forward args...]"""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 3
if "--rounds" in args:
    i = args.index("--rounds"); rounds = int(args[i + 1]); del args[i:i + 2]
libs = [a for a in args if a.endswith(".so")]
rest = [a for a in args if not a.endswith(".so")]
old = libs[0]; new = libs[1] if len(libs) > 1 else None
def run(lib):
    env = dict(os.environ, PD_ALLOW_ABI_MISMATCH="1")
    if lib: env["PD_LIB"] = os.path.abspath(lib)
    else: env.pop("PD_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "profile_forward.py")] + rest, env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True).stdout
    rows = []
    for l in out.splitlines():
        m = re.match(r"\s*(\d+)\s+(\S+)\s+(.*?)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l)
        if m: rows.append((int(m.group(1)), m.group(2), m.group(3).strip(), float(m.group(4))))
    return rows
res = {"old": [], "new": []}
for _ in range(rounds):
    res["old"].append(run(old)); res["new"].append(run(new))
def med(which, i): return statistics.median(r[i][3] for r in res[which])
n = len(res["old"][0])
kinds = {}
for i in range(n):
    k = res["old"][0][i][1]
    o, w = med("old", i), med("new", i)
    kinds.setdefault(k, [0.0, 0.0]); kinds[k][0] += o; kinds[k][1] += w
    if k.startswith("conv") or k == "attn_d8":
        print(f"{i:3d} {k:9s} {res['old'][0][i][2]:44s} {o:.3f} -> {w:.3f}  {w / o:.3f}")
tot = [sum(v[0] for v in kinds.values()), sum(v[1] for v in kinds.values())]
for k, (o, w) in kinds.items():
    print(f"{k:12s} {o:8.3f} -> {w:8.3f} ms  {w / o:.3f}")
print(f"{'total':12s} {tot[0]:8.3f} -> {tot[1]:8.3f} ms  {tot[1] / tot[0]:.3f}   ({rounds} rounds, medians)")
