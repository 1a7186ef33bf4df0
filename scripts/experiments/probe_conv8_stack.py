#!/usr/bin/env python3
"""Round 6 probe: how fast would the 8 x 8 level's 3x3 convolutions run if a workgroup's tile covered several images?  The SAME 2 048 pixels as
[32][8][8] (64-pixel tiles, 640 workgroups), as [16][16][8] (128-pixel tiles, 320) and as [1][256][8] (256-pixel tiles, 160): no seam handling, timing only."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phendiff_amd import _lib as L
from phendiff_amd.packing import pack_conv_weight
dev = "cuda:0"; lib = L.lib(); tdt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
for cin, c1 in ((1280, 0), (1280, 1280)):
    cout = 1280
    w = pack_conv_weight(torch.randn(cout, cin + c1, 3, 3) / ((cin + c1) * 9) ** 0.5, tdt).to(dev)
    bias = torch.randn(cout, device=dev)
    for (B, H, W) in ((32, 8, 8), (16, 16, 8), (8, 32, 8), (1, 256, 8), (2, 128, 8)):
        x0 = torch.randn(B, H, W, cin, device=dev).to(tdt)
        x1 = torch.randn(B, H, W, c1, device=dev).to(tdt) if c1 else None
        y = torch.empty(B, H, W, cout, device=dev, dtype=tdt)
        args = L.ConvArgs(dtype=1, B=B, Hin=H, Win=W, Hout=H, Wout=W, C0=cin, C1=c1, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1, upsample=0, silu=0,
                          out_mode=0, heads=0, x0=x0.data_ptr(), x1=L.ptr(x1), scale=None, shift=None, w_packed=w.data_ptr(), bias=bias.data_ptr(),
                          temb=None, temb_stride=0, residual=None, y=y.data_ptr())
        for _ in range(5):
            L.check(lib.pd_conv(C.byref(args), st))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            lib.pd_conv(C.byref(args), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        fl = 2.0 * B * H * W * cout * (cin + c1) * 9
        print(f"[{B}][{H}][{W}] {cin}+{c1}->{cout}: {dt*1e6:.1f} us  {fl/dt/1e12:.0f} TF/s", flush=True)
