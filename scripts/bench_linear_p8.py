#!/usr/bin/env python3
"""Same-process A/B of pd_linear's eight-phase 256 x 256 kernel (csrc/linear_p8.hip, PD_LIN_P8=1) against the shipped choice of
round 5 (PD_LIN_P8=0) on the nn.Linear shapes of the SD-2.1 UNet at batch B (forward layers and, with K / N swapped, their input
gradients), random operands, interleaved rounds (the switch is read at every dispatch).  GPU only.
    python scripts/bench_linear_p8.py [B] [--all]"""
import ctypes as C, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
from phendiff_amd.packing import pack_conv_weight

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
ALL = "--all" in sys.argv
dev, lib = "cuda:0", L.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, args, iters):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn(C.byref(args), st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


shapes = []   # (name, M, K, N, residual, glu)
for hw, ch in ((64, 320), (32, 640), (16, 1280)):
    M = B * hw * hw
    shapes += [(f"{hw}^2 ff1+glu", M, ch, 8 * ch, 0, 1), (f"{hw}^2 ff2", M, 4 * ch, ch, 1, 0), (f"{hw}^2 qkv", M, ch, 3 * ch, 0, 0),
               (f"{hw}^2 out", M, ch, ch, 1, 0)]
    if ALL:   # input gradients: dX = dY . W (K = the forward's N)
        shapes += [(f"{hw}^2 ff1 dgrad", M, 8 * ch, ch, 0, 0), (f"{hw}^2 ff2 dgrad", M, ch, 4 * ch, 0, 0), (f"{hw}^2 qkv dgrad", M, 3 * ch, ch, 0, 0)]
print(f"B = {B}; ms (median of 5 interleaved rounds x 20 launches) | TF/s", flush=True)
for name, M, K, N, res, glu in shapes:
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(M, K, generator=g) * 2 - 1).bfloat16().to(dev)
    w = pack_conv_weight(((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5)[:, :, None, None], torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    NO = N // 2 if glu else N
    y = torch.empty(M, NO, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, NO, device=dev).bfloat16() if res else None
    a = L.LinearArgs(dtype=1, M=M, K=K, N=N, N_pad=N, x=x.data_ptr(), x_stride=K, w_packed=w.data_ptr(), bias=bias.data_ptr(),
                     residual=L.ptr(r), y=y.data_ptr(), glu=glu)
    t = {"0": [], "1": []}
    for v in ("0", "1"):
        os.environ["PD_LIN_P8"] = v
        for _ in range(3): L.check(lib.pd_linear(C.byref(a), st))
    for _ in range(5):
        for v in ("0", "1"):
            os.environ["PD_LIN_P8"] = v
            t[v].append(timed(lib.pd_linear, a, 20))
    fl = 2.0 * M * K * N
    m0, m1 = statistics.median(t["0"]), statistics.median(t["1"])
    print(f"{name:16s} M={M:6d} K={K:5d} N={N:5d} res{res} glu{glu}: r5 {m0*1e3:.3f} ms {fl/m0/1e12:6.0f} | p8 {m1*1e3:.3f} ms {fl/m1/1e12:6.0f}"
          f" (best {fl/min(t['1'])/1e12:6.0f}) | x{m0/m1:.3f}", flush=True)
os.environ.pop("PD_LIN_P8", None)
