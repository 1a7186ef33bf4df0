#!/usr/bin/env python3
"""pd_attn_d64 / pd_attn_d64_bwd at the SD-2.1 self-attention shapes (B x heads x N x 64, q / k / v slices of one fused projection).
    python scripts/bench_attn_d64.py [B] [--launches N]     (--launches: only N forward launches of the 64^2 shape: the PMC target)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phendiff_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 32
only = int(sys.argv[sys.argv.index("--launches") + 1]) if "--launches" in sys.argv else 0
bwd = "--bwd" in sys.argv          # time (or, with --launches, only launch) pd_attn_d64_bwd instead
dev, lib = "cuda:0", L.lib()
st = torch.cuda.current_stream().cuda_stream
for hw, heads in ((64, 5), (32, 10), (16, 20)):
    N, Cc = hw * hw, heads * 64
    qkv = torch.randn(B, N, 3 * Cc, device=dev).bfloat16()
    out = torch.empty(B, N, Cc, device=dev, dtype=torch.bfloat16)
    es = 2
    a = L.AttnD64Args(dtype=1, B=B, heads=heads, Nq=N, Nkv=N, q=qkv.data_ptr(), q_stride=3 * Cc, k=qkv.data_ptr() + Cc * es, v=qkv.data_ptr() + 2 * Cc * es,
                      kv_stride=3 * Cc, out=out.data_ptr(), out_stride=Cc)
    if bwd:
        lse = torch.empty(B, heads, N, device=dev); delta = torch.empty(B, heads, N, device=dev)
        a.lse = lse.data_ptr()
        L.check(lib.pd_attn_d64(C.byref(a), st))
        do = torch.randn(B, N, Cc, device=dev).bfloat16(); dqkv = torch.empty_like(qkv)
        ab = L.AttnD64BwdArgs(dtype=1, B=B, heads=heads, Nq=N, Nkv=N, q=qkv.data_ptr(), q_stride=3 * Cc, k=qkv.data_ptr() + Cc * es, v=qkv.data_ptr() + 2 * Cc * es,
                              kv_stride=3 * Cc, o=out.data_ptr(), dout=do.data_ptr(), o_stride=Cc, lse=lse.data_ptr(), delta=delta.data_ptr(),
                              dq=dqkv.data_ptr(), dq_stride=3 * Cc, dk=dqkv.data_ptr() + Cc * es, dv=dqkv.data_ptr() + 2 * Cc * es, dkv_stride=3 * Cc)
        if only:
            for _ in range(only): L.check(lib.pd_attn_d64_bwd(C.byref(ab), st))
            torch.cuda.synchronize()
            break
        for _ in range(3): L.check(lib.pd_attn_d64_bwd(C.byref(ab), st))
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): lib.pd_attn_d64_bwd(C.byref(ab), st)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
        fl = 10.0 * B * heads * N * N * 64
        print(f"{hw}^2 heads {heads} N {N}: pd_attn_d64_bwd {best*1e3:.3f} ms {fl/best/1e12:6.0f} TF/s (5-product count; 7 products run: x1.4)", flush=True)
        continue
    if only:
        for _ in range(only): L.check(lib.pd_attn_d64(C.byref(a), st))
        torch.cuda.synchronize()
        break
    for _ in range(3): L.check(lib.pd_attn_d64(C.byref(a), st))
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): lib.pd_attn_d64(C.byref(a), st)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 20)
    fl = 4.0 * B * heads * N * N * 64
    print(f"{hw}^2 heads {heads} N {N}: pd_attn_d64 {best*1e3:.3f} ms {fl/best/1e12:6.0f} TF/s", flush=True)
