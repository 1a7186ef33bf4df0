#!/usr/bin/env python3
"""Upper bound of what folding the 41 `pd_gn_finalize` launches of a UNet forward into their producers could save (VERDICT r2 weak 10:
"costed and declined with an estimate, not a measurement"): one forward's launches back to back between one event pair, with and
without the finalize launches (the convolutions then read stale scale / shift vectors: same work, wrong numbers -- timing only).
    python scripts/experiments/gn_finalize_bound.py [--batch 32] [--size 256]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import phendiff_amd as P
from phendiff_amd import _lib as L
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--size", type=int, default=256); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(0)
m = P.CustomCondUNet2DModel(compute_dtype="bf16", **dict(P.UNET_CONFIGS["super_small"], sample_size=a.size)).to("cuda:0")
plan = m.plan_for(a.batch, a.size, a.size, torch.device("cuda:0"))
x = torch.randn(a.batch, 3, a.size, a.size, device="cuda"); out = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
temb = plan.temb_rows(torch.full((a.batch,), 500.0, device="cuda"), torch.zeros(a.batch, dtype=torch.int64, device="cuda"), None, st)
plan.run(x.data_ptr(), temb.data_ptr(), out.data_ptr(), st)
lib = L.lib()
def timed(ops):
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.check(lib.pd_event_create(C.byref(e0))); L.check(lib.pd_event_create(C.byref(e1)))
    for op in ops: L.check(op.fn(C.byref(op.args), st), op.what)
    best = 1e9
    for _ in range(3):
        lib.pd_event_record(e0, st)
        for _ in range(a.reps):
            for op in ops: op.fn(C.byref(op.args), st)
        lib.pd_event_record(e1, st)
        ms = C.c_float(); torch.cuda.synchronize(); lib.pd_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / a.reps)
    return best
full = timed(plan.ops)
nofin = timed([op for op in plan.ops if op.what != "gn_finalize"])
n = sum(op.what == "gn_finalize" for op in plan.ops)
print(f"forward {full:.3f} ms with {n} gn_finalize launches, {nofin:.3f} ms without them: upper bound of a fold = {full - nofin:.3f} ms "
      f"({100 * (full - nofin) / full:.2f} % of the forward, {1e3 * (full - nofin) / n:.1f} us per launch)")
