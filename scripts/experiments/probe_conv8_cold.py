#!/usr/bin/env python3
"""Round 6 probe: the 8 x 8 level's 3x3 convolutions with COLD weights (12 distinct 29.5-MB weight sets cycled: 354 MB > the 256-MB Infinity Cache,
as in the trajectory where 1.7 GB of weights pass between two uses of a layer), default work order against the channel-tile-major XCD order."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]
from phendiff_amd.packing import pack_conv_weight
dev = "cuda:0"; lib = L.lib(); tdt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
NW = 12
SHAPES = ((32, 8, 8, 1280, 0), (32, 8, 8, 1280, 1280), (16, 16, 8, 1280, 0), (8, 32, 8, 1280, 0))
if os.environ.get("PROBE_LEVELS"): SHAPES = ((32, 16, 16, 1280, 0), (32, 32, 32, 640, 0), (32, 64, 64, 320, 0), (32, 16, 16, 1280, 1280))
NW = int(os.environ.get("PROBE_NW", NW))
for (B, H, W, cin, c1) in SHAPES:
    cout = 1280 if H <= 16 else cin
    ws = [pack_conv_weight(torch.randn(cout, cin + c1, 3, 3) / ((cin + c1) * 9) ** 0.5, tdt).to(dev) for _ in range(NW)]
    bias = torch.randn(cout, device=dev)
    x0 = torch.randn(B, H, W, cin, device=dev).to(tdt)
    x1 = torch.randn(B, H, W, c1, device=dev).to(tdt) if c1 else None
    y = torch.empty(B, H, W, cout, device=dev, dtype=tdt)
    argl = [L.ConvArgs(dtype=1, B=B, Hin=H, Win=W, Hout=H, Wout=W, C0=cin, C1=c1, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1, upsample=0, silu=0,
                       out_mode=0, heads=0, x0=x0.data_ptr(), x1=L.ptr(x1), scale=None, shift=None, w_packed=w.data_ptr(), bias=bias.data_ptr(),
                       temb=None, temb_stride=0, residual=None, y=y.data_ptr()) for w in ws]
    ref = None
    for env in ("0", "1", "0", "1"):
        os.environ["PD_CONV_XCD"] = env
        for a in argl:
            L.check(lib.pd_conv(C.byref(a), st))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            for a in argl:
                lib.pd_conv(C.byref(a), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (n * NW)
        if ref is None: ref = y.clone()
        same = bool((y == ref).all())
        fl = 2.0 * B * H * W * cout * (cin + c1) * 9
        print(f"[{B}][{H}][{W}] {cin}+{c1}->{cout} cold weights PD_CONV_XCD={env}: {dt*1e6:.1f} us  {fl/dt/1e12:.0f} TF/s  identical={same}", flush=True)
