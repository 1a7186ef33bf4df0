#!/usr/bin/env python3
"""Gate 1 of the Winograd question: the stand-alone F(2x2, 3x3) kernel (wino_conv.hip -> libwino.so, built by
`hipcc --offload-arch=gfx950 -O3 -shared -fPIC wino_conv.hip -o libwino.so`) against the shipped implicit-GEMM pd_conv on the same
shape -- parity against F.conv2d (fp32 on the bf16-rounded operands) and HIP-event timing.
    python scripts/experiments/winograd/bench_wino.py [Cin Cout HW B]"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import phendiff_amd._lib as L  # noqa: E402
from phendiff_amd.packing import pack_conv_weight  # noqa: E402

cin, cout, hw, B = (int(v) for v in (sys.argv[1:5] + [256, 256, 64, 32][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
wl = C.CDLL(os.path.join(HERE, "libwino.so"))


class WinoP(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("x", C.c_void_p), ("u", C.c_void_p),
                ("bias", C.c_void_p), ("y", C.c_void_p)]


wl.wino_conv.argtypes = [C.POINTER(WinoP), C.c_void_p]
wl.wino_conv.restype = C.c_int
g = torch.Generator().manual_seed(0)
x = torch.randn(B, cin, hw, hw, generator=g).bfloat16()
w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).bfloat16()
bias = torch.randn(cout, generator=g)
# U = G g G^T (fp32), rounded to bf16 ONCE, in MFMA A-fragment order [16][Cout/32][Cin/16][64 lanes][8]
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
U = torch.einsum("ia,ocab,jb->ijoc", G, w.float(), G).reshape(16, cout, cin)
Up = U.reshape(16, cout // 32, 32, cin // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous().bfloat16().to(dev)     # [xi][ct][ks][h][r][j]
X = x.permute(0, 2, 3, 1).contiguous().to(dev)
Y = torch.full((B, hw, hw, cout), float("nan"), dtype=torch.bfloat16, device=dev)
bd = bias.to(dev)
st = torch.cuda.current_stream().cuda_stream
p = WinoP(B=B, H=hw, W=hw, Cin=cin, Cout=cout, x=X.data_ptr(), u=Up.data_ptr(), bias=bd.data_ptr(), y=Y.data_ptr())
rc = wl.wino_conv(C.byref(p), st)
torch.cuda.synchronize()
assert rc == 0, rc
nb = min(B, 2)
ref = F.conv2d(x[:nb].float(), w.float(), bias, padding=1)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


err_w = rel(Y[:nb].float().cpu().permute(0, 3, 1, 2), ref)
# the shipped kernel on the same operands: PLAIN (no prologue) and behind a GroupNorm + SiLU prologue (what the UNet's layers run)
lib = L.lib()
wp = pack_conv_weight(w.float(), torch.bfloat16, cout).to(dev)
Y2 = torch.empty_like(Y)
a = L.ConvArgs(dtype=1, B=B, Hin=hw, Win=hw, Hout=hw, Wout=hw, C0=cin, C1=0, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1, x0=X.data_ptr(),
               w_packed=wp.data_ptr(), bias=bd.data_ptr(), y=Y2.data_ptr())
L.check(lib.pd_conv(C.byref(a), st), "pd_conv")
torch.cuda.synchronize()
err_d = rel(Y2[:nb].float().cpu().permute(0, 3, 1, 2), ref)
sc, sh = torch.ones(B, cin, device=dev), torch.zeros(B, cin, device=dev)
a_gn = L.ConvArgs(dtype=1, B=B, Hin=hw, Win=hw, Hout=hw, Wout=hw, C0=cin, C1=0, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1, silu=1,
                  x0=X.data_ptr(), scale=sc.data_ptr(), shift=sh.data_ptr(), w_packed=wp.data_ptr(), bias=bd.data_ptr(), y=Y2.data_ptr())


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flops = 2.0 * B * hw * hw * cin * cout * 9
for r_ in range(3):
    tw = timed(lambda: wl.wino_conv(C.byref(p), st))
    td = timed(lambda: lib.pd_conv(C.byref(a), st))
    tg = timed(lambda: lib.pd_conv(C.byref(a_gn), st))
    print(f"{cin}->{cout} @{hw}^2 B={B} round {r_}: winograd {tw:.4f} ms ({flops / tw / 1e9:.0f} TF/s direct-equivalent)   pd_conv plain {td:.4f} ms "
          f"({flops / td / 1e9:.0f})   pd_conv GroupNorm+SiLU {tg:.4f} ms ({flops / tg / 1e9:.0f})   winograd / GN form = {tw / tg:.3f}", flush=True)
print(f"parity vs F.conv2d (fp32 on the bf16 operands), relative L2: winograd {err_w:.2e}   direct {err_d:.2e}   ratio {err_w / err_d:.2f}")
