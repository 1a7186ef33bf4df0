#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy report of the library's kernel sources (hipcc -Rpass-analysis=kernel-resource-usage,
cross-compiled for gfx950: runs without a GPU).     python scripts/kernel_resource_usage.py > profiles/r3_kernel_resource_usage.txt"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "phendiff_amd", "csrc")
srcs = sorted(f for f in os.listdir(CS) if f.endswith(".hip"))
def run(f):
    extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if f in ("attn_d8.hip", "sd_bwd_kernels.hip") else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                        *extra, os.path.join(CS, f), "-o", os.devnull], stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    return r.stderr
with ThreadPoolExecutor(4) as ex:
    outs = list(ex.map(run, srcs))
rows = {}
for out in outs:
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        if cur is None: continue
        for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("sgpr", r"TotalSGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(r"remark:\s+" + pat, line)
            if m and key not in cur: cur[key] = int(m.group(1))
names = sorted(rows)
dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
rows = {d: rows[n] for n, d in zip(names, dem)}
print("# kernel resource usage of libphendiff_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; scripts/kernel_resource_usage.py)")
print("# name | VGPRs | AGPRs | SGPRs | scratch B/lane | occupancy waves/SIMD | LDS static B")
for name in sorted(rows):
    r = rows[name]
    print(f"{name} | {r.get('vgpr')} | {r.get('agpr')} | {r.get('sgpr')} | {r.get('scratch')} | {r.get('occ')} | {r.get('lds')}")
scr = [n for n, r in rows.items() if r.get("scratch")]
print(f"# {len(rows)} kernels; with scratch: {len(scr)}")
for n in scr: print(f"#   {rows[n]['scratch']:4d} B  {n}")
