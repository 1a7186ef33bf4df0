#!/usr/bin/env python3
"""Where the scratch (spill) accesses of the kernels that have any sit relative to their MFMA stream: before the first MFMA, between
MFMAs, after the last one.  Cross-compiles the given sources to gfx950 assembly (no GPU needed).
    python scripts/scratch_placement.py [conv_igemm wgrad ...] > profiles/rN_scratch_placement.txt
A spill that is stored before the first MFMA and reloaded after the last is a value parked across the loop (a handful of scratch
instructions per workgroup); one BETWEEN MFMAs is a scratch round trip inside the hot loop (a vmcnt drain) -- the kind to fix."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "phendiff_amd", "csrc")
names = sys.argv[1:] or ["conv_igemm", "wgrad"]
print("# kernel | MFMAs | scratch stores / loads before the first MFMA | between MFMAs | after the last MFMA")
for n in names:
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if n in ("attn_d8", "sd_bwd_kernels") else []
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S"] + extra +
                       [os.path.join(SRC, n + ".hip"), "-o", t.name], check=True, stderr=subprocess.DEVNULL)
        s = open(t.name).read()
    funcs = re.findall(r"^(_Z[^\n:]*):[^\n]*\n(.*?)\.Lfunc_end", s, re.M | re.S)
    dem = subprocess.run(["c++filt"], input="\n".join(f for f, _ in funcs), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for (f, body), d in zip(funcs, dem):
        if "scratch_" not in body:
            continue
        total = body.count("v_mfma")
        nm = 0
        pre = [0, 0]; mid = [0, 0]; post = [0, 0]
        for l in body.split("\n"):
            if "v_mfma" in l:
                nm += 1
            m = re.search(r"scratch_(store|load)", l)
            if m:
                k = 0 if m.group(1) == "store" else 1
                (pre if nm == 0 else post if nm == total else mid)[k] += 1
        print(f"{d} | {total} | {pre[0]} / {pre[1]} | {mid[0]} / {mid[1]} | {post[0]} / {post[1]}")
