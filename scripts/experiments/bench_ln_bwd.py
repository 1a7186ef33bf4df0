#!/usr/bin/env python3
"""Round 6 probe: pd_layernorm_bwd (kernel + its partial reduce) against the number of workgroups (= rows of the partial the reduce folds)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phendiff_amd import _lib as L
dev = "cuda:0"; lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
for rows, Cc in ((131072, 320), (32768, 640), (8192, 1280)):
    x = torch.randn(rows, Cc, device=dev).bfloat16(); dy = torch.randn(rows, Cc, device=dev).bfloat16(); res = torch.randn(rows, Cc, device=dev).bfloat16()
    dx = torch.empty_like(x); gm = torch.randn(Cc, device=dev)
    dg, db, ds = (torch.zeros(Cc, device=dev) for _ in range(3))
    for cap in ("2048", "1024", "768", "512", "384", "256"):
        os.environ["PD_LN_BWD_BLOCKS"] = cap
        nb = lib.pd_layernorm_bwd_blocks(rows)        # (the bound: the launch uses min(cap, bound) workgroups)
        part = torch.empty(nb * 3 * Cc, device=dev)
        a = L.LayerNormBwdArgs(dtype=1, rows=rows, C=Cc, eps=1e-5, x=x.data_ptr(), dy=dy.data_ptr(), gamma=gm.data_ptr(), res=res.data_ptr(), dx=dx.data_ptr(),
                               dgamma=dg.data_ptr(), dbeta=db.data_ptr(), partial=part.data_ptr(), dxsum=ds.data_ptr())
        for _ in range(3): L.check(lib.pd_layernorm_bwd(C.byref(a), st))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): lib.pd_layernorm_bwd(C.byref(a), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(f"rows {rows} C {Cc} cap {cap}: {dt*1e6:.1f} us  ({4 * rows * Cc * 2 / dt / 1e12:.2f} TB/s)", flush=True)
