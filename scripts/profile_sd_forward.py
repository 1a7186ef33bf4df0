#!/usr/bin/env python3
"""Per-launch device time of ONE SD-2.1 UNet forward (HIP events between launches), grouped by kind and shape.  GPU only.
    python scripts/profile_sd_forward.py [--batch 32] [--size 64]"""
import argparse, collections, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phendiff_amd as P
from phendiff_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--size", type=int, default=64)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--all", action="store_true")
a = ap.parse_args()
torch.manual_seed(0)
dev = torch.device("cuda:0")
with torch.device(dev):
    m = P.SDUNet2DConditionModel(compute_dtype=a.dtype, **P.SD21_UNET_CONFIG)
x = torch.randn(a.batch, 4, a.size, a.size, device=dev)
ehs = torch.randn(a.batch, 77, 1024, device=dev)
m(x, 500, ehs)
plan = next(p for k, p in m._plans.items() if k[0] == a.batch)
st = torch.cuda.current_stream().cuda_stream
lib = L.lib()
evs = []
for _ in range(len(plan.ops) + 1):
    e = C.c_void_p(); L.check(lib.pd_event_create(C.byref(e))); evs.append(e)
tot = [0.0] * len(plan.ops)
for _ in range(a.reps):
    for i, op in enumerate(plan.ops):
        lib.pd_event_record(evs[i], st); L.check(op.fn(C.byref(op.args), st), op.what)
    lib.pd_event_record(evs[-1], st)
    torch.cuda.synchronize()
    ms = C.c_float()
    for i in range(len(plan.ops)):
        lib.pd_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)); tot[i] += ms.value / a.reps


def shape(op):
    g = op.args
    if hasattr(g, "Hin"):
        return f"{g.Hin}x{g.Win} {g.C0}+{g.C1}->{g.Cout} k{g.ksize} s{g.stride} up{g.upsample} gn{int(bool(g.scale))}"
    if isinstance(g, L.LinearArgs):
        return f"M={g.M} K={g.K} N={g.N} res{int(bool(g.residual))} gn{int(bool(g.scale))} st{int(bool(g.stats_out))} glu{g.glu}"
    if isinstance(g, L.AttnD64Args):
        return f"heads={g.heads} Nq={g.Nq} Nkv={g.Nkv}"
    return ""


groups = collections.OrderedDict()
for op, t in zip(plan.ops, tot):
    k = (op.what, shape(op))
    d = groups.setdefault(k, [0, 0.0, 0.0, 0.0])
    d[0] += 1; d[1] += t; d[2] += op.flops; d[3] += op.bytes
total = sum(tot)
print(f"{'kind':10} {'shape':52} {'n':>3} {'ms':>8} {'%':>5} {'TF/s':>7} {'GB/s':>7}")
for (what, shp), (n, t, fl, by) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    if t / total < 0.004 and not a.all:
        continue
    print(f"{what:10} {shp:52} {n:3d} {t:8.3f} {100*t/total:5.1f} {fl / t / 1e9 if t > 0 else 0:7.1f} {by / t / 1e6 if t > 0 else 0:7.1f}")
print(f"total ms: {total:.2f}  ({len(plan.ops)} launches)")
