#!/usr/bin/env python3
"""Race screen for the DMA-staged attention: many launches on the same inputs must give bit-identical outputs (and match the
register-staged kernel's error level).  Diagnostic, GPU only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from phendiff_amd import _lib as L
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
bad = 0
for (B, heads, N) in ((8, 32, 1024), (4, 32, 2100), (32, 32, 4096), (2, 32, 8192 + 96)):
    g = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(B, heads, N, 8, generator=g).bfloat16().cuda().contiguous() for _ in range(3))
    kmax2 = (k.float() ** 2).sum(-1).amax(-1).contiguous()
    outs = []
    for it in range(60):
        out = torch.full((B, N, heads * 8), float("nan"), dtype=torch.bfloat16, device="cuda")
        a = L.AttnArgs(dtype=1, B=B, heads=heads, N=N, q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), out=out.data_ptr(), kmax2=kmax2.data_ptr())
        L.check(lib.pd_attn_d8(C.byref(a), st))
        outs.append(out)
    torch.cuda.synchronize()
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    fin = bool(torch.isfinite(outs[0].float()).all())
    print(f"B={B} N={N}: 60 launches identical={same} finite={fin}", flush=True)
    bad += (not same) or (not fin)
sys.exit(1 if bad else 0)
