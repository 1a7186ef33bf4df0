#!/usr/bin/env python3
"""pd_attn_d8_bwd at configs[1]'s shape (B = 112, 32 heads, N = 1 024, bf16): the two-kernel path against the one-pass form
(pd_attn_bwd_args.slab), HIP-event timed, alternating.   python scripts/bench_attn_bwd.py [B] [heads] [N]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import phendiff_amd._lib as L  # noqa: E402

B, heads, N = (int(v) for v in (sys.argv[1:4] + [112, 32, 1024][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
lib = L.lib()
g = torch.Generator().manual_seed(0)
Cc = heads * 8
q, k, v = (torch.randn(B, heads, N, 8, generator=g).to(torch.bfloat16).to(dev) for _ in range(3))
do = torch.randn(B, N, Cc, generator=g).to(torch.bfloat16).to(dev)
out = torch.empty((B, N, Cc), dtype=torch.bfloat16, device=dev)
lse = torch.empty((B, heads, N), device=dev)
st = torch.cuda.current_stream().cuda_stream
a = L.AttnArgs(dtype=1, B=B, heads=heads, N=N, q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), out=out.data_ptr(), lse=lse.data_ptr())
L.check(lib.pd_attn_d8(C.byref(a), st), "pd_attn_d8")
delta = torch.empty((B, heads, N), device=dev)
dqkv = torch.empty((B, N, 3 * Cc), dtype=torch.bfloat16, device=dev)
b = L.AttnBwdArgs(dtype=1, B=B, heads=heads, N=N, q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), o=out.data_ptr(), dout=do.data_ptr(),
                  lse=lse.data_ptr(), delta=delta.data_ptr(), dqkv=dqkv.data_ptr())
need = int(lib.pd_attn_d8_bwd_workspace(C.byref(b)))
slab = torch.empty(max(need // 4, 1), device=dev)


def run(fused, reps=20):
    b.slab, b.slab_bytes = (slab.data_ptr(), need) if fused else (None, 0)
    for _ in range(3):
        L.check(lib.pd_attn_d8_bwd(C.byref(b), st), "pd_attn_d8_bwd")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.pd_attn_d8_bwd(C.byref(b), st), "pd_attn_d8_bwd")
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flops = 10.0 * B * heads * N * N * 8
for r in range(3):
    t2, t1 = run(False), run(True)
    print(f"round {r}: two kernels {t2:.4f} ms ({flops / t2 / 1e9:.0f} TF/s)   one pass {t1:.4f} ms ({flops / t1 / 1e9:.0f} TF/s)   workspace {need / 2**20:.0f} MiB", flush=True)
