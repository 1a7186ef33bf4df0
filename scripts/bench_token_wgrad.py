#!/usr/bin/env python3
"""pd_token_wgrad on the Linear shapes of the SD-2.1 transformer blocks (dW = dY^T X over M = B * tokens).  GPU only.
    python scripts/bench_token_wgrad.py [B]          (PD_TW_DMA=0..3 forces a form; PD_LIB=<.so> another build)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev, lib = "cuda:0", L.lib()
lib.pd_token_wgrad_workspace.restype = C.c_size_t
st = torch.cuda.current_stream().cuda_stream
def run(args, iters=10):
    for _ in range(3): L.check(lib.pd_token_wgrad(C.byref(args), st))
    best = float("inf")
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters): lib.pd_token_wgrad(C.byref(args), st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / iters)
    return best
tot_t = tot_f = 0.0
for hw, ch in ((64, 320), (32, 640), (16, 1280)):
    for name, K, N in (("qkv", ch, 3 * ch), ("out", ch, ch), ("ff1", ch, 8 * ch), ("ff2", 4 * ch, ch)):
        M = B * hw * hw
        x = torch.randn(M, K, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        dw = torch.zeros(N, K, device=dev)
        a = L.TokenWgradArgs(dtype=1, M=M, K=K, N=N, x=x.data_ptr(), x_stride=K, dy=dy.data_ptr(), dy_stride=N, dw=dw.data_ptr(), accumulate=0)
        need = lib.pd_token_wgrad_workspace(C.byref(a))
        slab = torch.empty(need // 4, dtype=torch.float32, device=dev)
        a.slab, a.slab_bytes = slab.data_ptr(), need
        t = run(a)
        fl = 2.0 * M * K * N
        tot_t += t; tot_f += fl
        print(f"{hw}x{hw} C={ch} {name:4s} M={M} K={K} N={N}: {t*1e3:.3f} ms {fl/t/1e12:6.0f} TF/s  slab {need/2**20:.0f} MiB", flush=True)
print(f"sum {tot_t*1e3:.3f} ms  {tot_f/tot_t/1e12:.0f} TF/s")
