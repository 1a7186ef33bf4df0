"""Time pd_conv_wgrad on the weight-gradient shapes of a super_small training step (B x 128 x 128)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import phendiff_amd._lib as L  # noqa: E402
if os.environ.get("PD_LIB"):
    L.LIB_PATH = os.environ["PD_LIB"]      # same-box A/B of two builds


def bench(B, cin, cout, H, W, ksize=3, affine=True, reps=20, splits=None):
    lib = L.lib()
    dev = "cuda:0"
    x = torch.randn(B, H, W, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, W, cout, device=dev).to(torch.bfloat16)
    sc = torch.rand(B, cin, device=dev) + 0.5
    sh = torch.randn(B, cin, device=dev) * 0.1
    dw = torch.zeros(cout, cin, ksize, ksize, device=dev)
    a = L.WgradArgs(dtype=1, B=B, Hin=H, Win=W, Hout=H, Wout=W, C0=cin, C1=0, Cout=cout, ksize=ksize, stride=1, pad=ksize // 2,
                    upsample=0, silu=int(affine), x0=x.data_ptr(), x1=None, scale=sc.data_ptr() if affine else None,
                    shift=sh.data_ptr() if affine else None, dy=dy.data_ptr(), dw=dw.data_ptr(), accumulate=0)
    want = lib.pd_conv_wgrad_workspace(C.byref(a))
    per = ksize * ksize * ((cout + 63) // 64 * 64) * ((cin + 63) // 64 * 64) * 4
    nbytes = want if splits is None else per * splits
    slab = torch.empty(nbytes // 4, device=dev)
    a.slab, a.slab_bytes = slab.data_ptr(), nbytes
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.pd_conv_wgrad(C.byref(a), st), "wgrad")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.pd_conv_wgrad(C.byref(a), st), "wgrad")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * B * H * W * cin * cout * ksize * ksize
    print(f"B{B} {cin:4d}->{cout:4d} {H:3d}x{W:3d} k{ksize} splits={nbytes // per:4d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for cfg in [(64, 64, 128), (128, 64, 128), (192, 64, 128), (128, 128, 64), (256, 128, 64), (384, 128, 64), (256, 256, 32), (512, 256, 32), (384, 256, 32)]:
        bench(B, cfg[0], cfg[1], cfg[2], cfg[2])
    for cfg in [(320, 320, 64), (640, 320, 64), (640, 640, 32), (1280, 640, 32), (1280, 1280, 16), (2560, 1280, 16), (960, 640, 32)]:   # SD-2.1 UNet widths
        bench(B, cfg[0], cfg[1], cfg[2], cfg[2])
    bench(B, 256, 256, 32, 32, affine=False)
    for s in (1, 4, 16, 64):
        bench(B, 256, 256, 32, 32, splits=s)
    for s in (8, 64, 512):
        bench(B, 64, 64, 128, 128, splits=s)
    bench(B, 256, 768, 32, 32, ksize=1)
    bench(B, 64, 64, 128, 128, ksize=1)
