#!/usr/bin/env python3
"""Single pd_conv configuration in a loop (for rocprofv3 counter passes).  GPU only.
    python scripts/bench_conv.py --hw 256 --cin 64 --cout 64 --gn 1 --res 1 [--c1 0] [--ks 3] [--iters 20]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]
from phendiff_amd.packing import pack_conv_weight
ap = argparse.ArgumentParser()
for k, d in dict(batch=32, hw=256, cin=64, c1=0, cout=64, ks=3, stride=1, up=0, gn=1, res=0, iters=20, mode=0).items():
    ap.add_argument(f"--{k}", type=int, default=d)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
code, tdt = (1, torch.bfloat16) if a.dtype == "bf16" else (0, torch.float32)
dev = "cuda:0"; lib = L.lib()
B, H = a.batch, a.hw
x0 = torch.randn(B, H, H, a.cin, device=dev).to(tdt)
x1 = torch.randn(B, H, H, a.c1, device=dev).to(tdt) if a.c1 else None
cin = a.cin + a.c1
w = pack_conv_weight(torch.randn(a.cout, cin, a.ks, a.ks) / (cin * a.ks * a.ks) ** 0.5, tdt).to(dev)
bias = torch.randn(a.cout, device=dev)
hc = 2 * H if a.up else H
pad = 1 if a.ks == 3 else 0
ho = (hc + 2 * pad - a.ks) // a.stride + 1
y = torch.empty(B, ho, ho, a.cout, device=dev, dtype=tdt)
res = torch.randn(B, ho, ho, a.cout, device=dev).to(tdt) if a.res else None
sc = torch.rand(B, cin, device=dev) + 0.5 if a.gn else None
sh = torch.randn(B, cin, device=dev) if a.gn else None
temb = torch.randn(B, 2752, device=dev)
args = L.ConvArgs(dtype=code, B=B, Hin=H, Win=H, Hout=ho, Wout=ho, C0=a.cin, C1=a.c1, Cout=a.cout, Cout_pad=a.cout, ksize=a.ks,
                  stride=a.stride, pad=pad, upsample=a.up, silu=a.gn, out_mode=0, heads=0, x0=x0.data_ptr(), x1=L.ptr(x1),
                  scale=L.ptr(sc), shift=L.ptr(sh), w_packed=w.data_ptr(), bias=bias.data_ptr(), temb=temb.data_ptr(),
                  temb_stride=2752, residual=L.ptr(res), y=y.data_ptr())
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.check(lib.pd_conv(C.byref(args), st))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters):
    lib.pd_conv(C.byref(args), st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
fl = 2.0 * B * ho * ho * a.cout * cin * a.ks * a.ks
print(f"conv {H}x{H} {a.cin}+{a.c1}->{a.cout} k{a.ks} s{a.stride} up{a.up} gn{a.gn} res{a.res}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s")
