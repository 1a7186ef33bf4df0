#!/usr/bin/env python3
"""Round-3 bounded experiment: does the VALU-bound d = 8 attention co-run with the MFMA-heavy 3x3 convolution when
their CU residency and wave priority are CONTROLLED?  One layer pair of the headline network at half batch each:
    attention  B = 16, 32 heads, N = 4096 (DMA-staged kernel)          ~0.83 ms per launch
    conv 3x3   256 -> 256 @64x64, B = 16, GroupNorm + SiLU prologue     ~0.09 ms per launch  (xK launches to balance the attention's time)
on two HIP streams, against the same launches back to back on one stream.  Knobs (environment, read by the library):
    PD_ATTN_LDS_PAD=<bytes>    dynamic LDS per attention workgroup: 70000 -> ONE attention workgroup (8 waves = 2 per SIMD, 127 VGPRs) per CU,
                               leaving registers (512 - 256 = 256 per SIMD) and LDS (~90 KB) for one NCO = 2 conv workgroup (250 VGPRs, 49 KB)
    PD_LIB=<lib built with -DPD_CONV_PRIO_BASE=2>   conv waves at s_setprio 2 / 3 (attention stays at 0)
Prints: alone / serial / concurrent wall times; gain = serial / concurrent.   GPU only.
    python scripts/experiments/overlap_pair.py [--reps 20] [--convs 9]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from phendiff_amd import _lib as L
if os.environ.get("PD_LIB"): L.LIB_PATH = os.environ["PD_LIB"]
from phendiff_amd.packing import pack_conv_weight
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=20); ap.add_argument("--convs", type=int, default=9); ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
dev, lib, tdt = "cuda:0", L.lib(), torch.bfloat16
B, heads, N = a.batch, 32, 4096
q, k, v = (torch.randn(B, heads, N, 8, device=dev).to(tdt) for _ in range(3))
out = torch.empty(B, N, heads * 8, device=dev, dtype=tdt)
kmax2 = (k.float() ** 2).sum(-1).amax(-1).contiguous()
aa = L.AttnArgs(dtype=1, B=B, heads=heads, N=N, q=q.data_ptr(), k=k.data_ptr(), v=v.data_ptr(), out=out.data_ptr(), kmax2=kmax2.data_ptr())
H, Cc = 64, 256
x0 = torch.randn(B, H, H, Cc, device=dev).to(tdt)
w = pack_conv_weight(torch.randn(Cc, Cc, 3, 3) / (Cc * 9) ** 0.5, tdt).to(dev)
bias = torch.randn(Cc, device=dev); y = torch.empty(B, H, H, Cc, device=dev, dtype=tdt)
sc, sh = torch.rand(B, Cc, device=dev) + 0.5, torch.randn(B, Cc, device=dev)
ca = L.ConvArgs(dtype=1, B=B, Hin=H, Win=H, Hout=H, Wout=H, C0=Cc, C1=0, Cout=Cc, Cout_pad=Cc, ksize=3, stride=1, pad=1, upsample=0, silu=1,
                out_mode=0, heads=0, x0=x0.data_ptr(), x1=None, scale=sc.data_ptr(), shift=sh.data_ptr(), w_packed=w.data_ptr(),
                bias=bias.data_ptr(), temb=None, temb_stride=0, residual=None, y=y.data_ptr())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def attn(st, n):
    for _ in range(n): L.check(lib.pd_attn_d8(C.byref(aa), st.cuda_stream))
def conv(st, n):
    for _ in range(n): L.check(lib.pd_conv(C.byref(ca), st.cuda_stream))
def timed(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
R, K = a.reps, a.convs
t_attn = timed(lambda: attn(s1, R))
t_conv = timed(lambda: conv(s1, R * K))
def serial():
    for _ in range(R): attn(s1, 1); conv(s1, K)
def concurrent():
    attn(s1, R); conv(s2, R * K)
t_ser, t_con = timed(serial), timed(concurrent)
t_con2 = timed(concurrent)
print(f"env: PD_ATTN_LDS_PAD={os.environ.get('PD_ATTN_LDS_PAD')} PD_LIB={os.environ.get('PD_LIB')} PD_CONV_NCO={os.environ.get('PD_CONV_NCO')}")
print(f"attention alone {t_attn / R:.3f} ms/launch; conv alone {t_conv / (R * K):.4f} ms/launch (x{K} = {t_conv / R:.3f} ms per attention)")
print(f"serial (1 stream) {t_ser / R:.3f} ms per pair-group; concurrent (2 streams) {t_con / R:.3f} / {t_con2 / R:.3f} ms; "
      f"gain over serial {t_ser / min(t_con, t_con2):.3f}x; over the sum of the un-capped alone times see the other runs")
