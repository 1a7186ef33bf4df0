#!/usr/bin/env python3
"""Why is pd_linear 30x slow at M=32768 K=640 N=1920?  (diagnostic, GPU only)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phendiff_amd import _lib as L
from phendiff_amd.packing import pack_conv_weight
dev, lib = "cuda:0", L.lib()
st = torch.cuda.current_stream().cuda_stream
def run(M, K, N, iters=10, scale=1.0):
    x = (torch.randn(M, K, device=dev) * scale).bfloat16()
    w = pack_conv_weight(torch.randn(N, K, 1, 1) / K ** 0.5, torch.bfloat16).to(dev)
    bias = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    la = L.LinearArgs(dtype=1, M=M, K=K, N=N, N_pad=N, x=x.data_ptr(), x_stride=K, w_packed=w.data_ptr(), bias=bias.data_ptr(), residual=None, y=y.data_ptr())
    for _ in range(3): L.check(lib.pd_linear(C.byref(la), st))
    torch.cuda.synchronize(); ts = []
    for _ in range(iters):
        t0 = time.perf_counter(); lib.pd_linear(C.byref(la), st); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"M={M} K={K} N={N} scale={scale}: min {min(ts):.3f} ms  max {max(ts):.3f} ms  {2.0*M*K*N/min(ts)/1e9:.0f} TF/s", flush=True)
for (M, K, N) in ((32768, 640, 1920), (32768, 640, 1792), (32768, 640, 2048), (32768, 640, 640), (32768, 576, 1920), (32768, 704, 1920),
                  (32768 - 128, 640, 1920), (16384, 640, 1920), (65536, 640, 1920), (32768, 640, 5120), (32768, 1280, 1920)):
    run(M, K, N)
run(32768, 640, 1920, scale=0.0)
