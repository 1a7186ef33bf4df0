#!/usr/bin/env python3
"""pd_attn_d8 alone (B=32, 32 heads, N=4096 by default).  GPU only."""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]
ap = argparse.ArgumentParser()
for k, d in dict(batch=32, heads=32, n=4096, iters=10, kmax=0).items(): ap.add_argument(f"--{k}", type=int, default=d)
ap.add_argument("--dtype", default="bf16"); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
code, tdt = {"bf16": (1, torch.bfloat16), "fp16": (2, torch.float16), "f32": (0, torch.float32)}[a.dtype]
lib = L.lib()
q, k, v = (torch.randn(a.batch, a.heads, a.n, 8, device="cuda") * a.scale for _ in range(3))
q, k, v = q.to(tdt), k.to(tdt), v.to(tdt)
out = torch.empty(a.batch, a.n, a.heads * 8, device="cuda", dtype=tdt)
kmax2 = (k.float() ** 2).sum(-1).amax(-1).contiguous() if a.kmax else None     # pd_linear kmax2_out's result -> the DMA-staged kernel
args = L.AttnArgs(dtype=code, B=a.batch, heads=a.heads, N=a.n, q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), out=out.data_ptr(), kmax2=L.ptr(kmax2))
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): L.check(lib.pd_attn_d8(C.byref(args), st))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): lib.pd_attn_d8(C.byref(args), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
tiles = a.batch * a.heads * (a.n / 32) ** 2
print(f"attn kmax={a.kmax} B={a.batch} h={a.heads} N={a.n} scale={a.scale}: {dt*1e3:.3f} ms  {4.0*a.batch*a.heads*a.n*a.n*8/dt/1e12:.1f} TF/s  {dt*2.4e9/(tiles/1024):.0f} cyc/tile/SIMD@2.4GHz")
