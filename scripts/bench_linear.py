#!/usr/bin/env python3
"""pd_linear vs pd_conv (1x1) on the Linear shapes of the SD-2.1 transformer blocks.  GPU only.
    python scripts/bench_linear.py [B]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
from phendiff_amd.packing import pack_conv_weight
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]      # same-box A/B of two builds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev, lib = "cuda:0", L.lib()
st = torch.cuda.current_stream().cuda_stream
def run(fn, args, iters=20):
    for _ in range(3): L.check(fn(C.byref(args), st))
    best = float("inf")
    for _ in range(4):                     # best of 4 x 20: single groups are hit by multi-millisecond hiccups on shared boxes
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): fn(C.byref(args), st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / iters)
    return best
for hw, ch in ((64, 320), (32, 640), (16, 1280)):
    for name, K, N in (("qkv", ch, 3 * ch), ("out", ch, ch), ("ff1", ch, 8 * ch), ("ff2", 4 * ch, ch)):
        M = B * hw * hw
        x = torch.randn(M, K, device=dev).bfloat16()
        w = pack_conv_weight(torch.randn(N, K, 1, 1) / K ** 0.5, torch.bfloat16).to(dev)
        bias = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        la = L.LinearArgs(dtype=1, M=M, K=K, N=N, N_pad=N, x=x.data_ptr(), x_stride=K, w_packed=w.data_ptr(), bias=bias.data_ptr(), residual=None, y=y.data_ptr())
        ca = L.ConvArgs(dtype=1, B=B, Hin=hw, Win=hw, Hout=hw, Wout=hw, C0=K, C1=0, Cout=N, Cout_pad=N, ksize=1, stride=1, pad=0, upsample=0, silu=0,
                        out_mode=0, heads=0, x0=x.data_ptr(), x1=None, scale=None, shift=None, w_packed=w.data_ptr(), bias=bias.data_ptr(), temb=None,
                        temb_stride=0, residual=None, y=y.data_ptr())
        fl = 2.0 * M * K * N
        tl = run(lib.pd_linear, la)
        tc = run(lib.pd_conv, ca) if os.environ.get("PD_BENCH_CONV1X1") else float("nan")
        extra = ""
        if name == "ff1":          # fused GEGLU epilogue (weights are random: the tile interleave does not matter for timing)
            la.glu = 1
            tg = run(lib.pd_linear, la)
            extra = f" | glu {tg*1e3:.3f} ms {fl/tg/1e12:6.0f} TF/s"
        print(f"{hw}x{hw} C={ch} {name:4s} M={M} K={K} N={N}: pd_linear {tl*1e3:.3f} ms {fl/tl/1e12:6.0f} TF/s | pd_conv 1x1 {tc*1e3:.3f} ms {fl/tc/1e12:6.0f} TF/s{extra}", flush=True)
