#!/usr/bin/env python3
"""Localise errors of the DMA-staged attention (diagnostic, GPU only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from phendiff_amd import _lib as L
lib = L.lib()
B, heads, N = 8, 32, 1024
g = torch.Generator().manual_seed(27)
q, k, v = (torch.randn(B, heads, N, 8, generator=g).bfloat16().float() for _ in range(3))
Q, K, V = (t.bfloat16().cuda().contiguous() for t in (q, k, v))
kmax2 = (K.float() ** 2).sum(-1).amax(-1).contiguous()
st = torch.cuda.current_stream().cuda_stream
outs = []
for km in (kmax2, None):
    out = torch.full((B, N, heads * 8), float("nan"), dtype=torch.bfloat16, device="cuda")
    a = L.AttnArgs(dtype=1, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr(), kmax2=L.ptr(km))
    L.check(lib.pd_attn_d8(C.byref(a), st)); torch.cuda.synchronize()
    outs.append(out.float().cpu().reshape(B, N, heads, 8))
ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2)       # [B][N][heads][8]
for name, o in (("dma", outs[0]), ("reg", outs[1])):
    e = (o - ref).abs()
    print(name, "max err", float(e.max()), "rel", float((o - ref).norm() / ref.norm()))
e = (outs[0] - ref).abs()
print("err by d:", e.amax((0, 1, 2)))
print("err by query%32 (first 32):", e.amax((0, 2, 3)).reshape(-1, 32).amax(0))
print("err by query block of 32:", e.amax((0, 2, 3)).reshape(-1, 32).amax(1))
print("err by head:", e.amax((0, 1, 3)))
print("ratio out/ref sample:", (outs[0][0, :4, 0] / ref[0, :4, 0]))
