#!/usr/bin/env python3
"""Per-launch device time of ONE UNet forward (HIP events between launches).  Diagnostic, GPU only.
    python scripts/profile_forward.py [--batch 32] [--size 256] [--dtype bf16]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phendiff_amd as P
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]      # same-box A/B of two builds (scripts/ab_forward.sh)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--model", default="super_small"); ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
torch.manual_seed(0)
m = P.CustomCondUNet2DModel(compute_dtype=a.dtype, **dict(P.UNET_CONFIGS[a.model], sample_size=a.size)).to("cuda:0")
plan = m.plan_for(a.batch, a.size, a.size, torch.device("cuda:0"))
x = torch.randn(a.batch, 3, a.size, a.size, device="cuda"); out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
temb = plan.temb_rows(torch.full((a.batch,), 500.0, device="cuda"), torch.zeros(a.batch, dtype=torch.int64, device="cuda"), None, st)
lib = L.lib()
plan.run(x.data_ptr(), temb.data_ptr(), out.data_ptr(), st)
evs = []
for _ in range(len(plan.ops) + 1):
    e = C.c_void_p(); L.check(lib.pd_event_create(C.byref(e))); evs.append(e)
tot = [0.0] * len(plan.ops)
for _ in range(a.reps):
    for i, op in enumerate(plan.ops):
        lib.pd_event_record(evs[i], st); L.check(op.fn(C.byref(op.args), st), op.what)
    lib.pd_event_record(evs[-1], st)
    ms = C.c_float()
    for i in range(len(plan.ops)):
        lib.pd_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)); tot[i] += ms.value / a.reps
print(f"{'#':>3} {'kind':9} {'shape':44} {'ms':>8} {'TF/s':>7} {'GB/s':>7}")
for i, (op, t) in enumerate(zip(plan.ops, tot)):
    g = op.args
    if op.what.startswith("conv") and hasattr(g, "Hin"):
        shp = f"{g.Hin}x{g.Win} {g.C0}+{g.C1}->{g.Cout} s{g.stride} up{g.upsample} gn{int(bool(g.scale))} res{int(bool(g.residual))} m{g.out_mode}"
    elif op.what == "attn_d8":
        shp = f"N={g.N} heads={g.heads}"
    elif op.what == "gn_stats":
        shp = f"HW={g.HW} C={g.C0}+{g.C1} splits={g.splits}"
    else:
        shp = ""
    print(f"{i:3d} {op.what:9} {shp:44} {t:8.3f} {op.flops / t / 1e9 if t > 0 else 0:7.1f} {op.bytes / t / 1e6 if t > 0 else 0:7.1f}")
print("total ms:", sum(tot))
