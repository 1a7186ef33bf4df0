"""Backward building blocks on a real MI355X against torch.autograd of the same fp32 ops (the oracle's modules are these
torch ops): convolution input gradients through pd_conv (flipped/transposed weights, zero-stuffed stride-2, pooled
upsample), GroupNorm(+SiLU) backward, channel sums."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import DT, TOL, bf16_round, env, nhwc, rel, run_conv, stream  # noqa: F401  (env is a fixture)

pytestmark = pytest.mark.gpu


def from_nhwc(y):
    return y.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 96, 16, 16, 3), (1, 32, 64, 40, 24, 3), (2, 128, 64, 8, 8, 1)])
def test_conv_input_gradient_stride1(env, mode, shape):
    from phendiff_amd.packing import dgrad_weight
    B, cin, cout, H, W, k = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, cin, H, W, generator=g, requires_grad=True)
    w = bf16_round(torch.randn(cout, cin, k, k, generator=g) * 0.05, mode)
    dy = bf16_round(torch.randn(B, cout, H, W, generator=g), mode)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, padding=k // 2), x, dy)
    got = run_conv(env, mode, dy, dgrad_weight(w), torch.zeros(cin), ksize=k, pad=k // 2)
    assert rel(from_nhwc(got), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [(32, 32), (16, 48), (8, 8)])
def test_conv_input_gradient_stride2_zero_stuffed(env, mode, hw):
    from phendiff_amd.packing import dgrad_weight
    H, W = hw
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 64, H, W, generator=g, requires_grad=True)
    w = bf16_round(torch.randn(96, 64, 3, 3, generator=g) * 0.05, mode)
    dy = bf16_round(torch.randn(2, 96, H // 2, W // 2, generator=g), mode)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, stride=2, padding=1), x, dy)
    got = run_conv(env, mode, dy, dgrad_weight(w), torch.zeros(64), upsample=2)
    assert rel(from_nhwc(got), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_conv_input_gradient_fused_upsample(env, mode):
    from phendiff_amd.packing import dgrad_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 64, 8, 16, generator=g, requires_grad=True)
    w = bf16_round(torch.randn(64, 64, 3, 3, generator=g) * 0.05, mode)
    dy = bf16_round(torch.randn(2, 64, 16, 32, generator=g), mode)
    (ref,) = torch.autograd.grad(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1), x, dy)
    du = run_conv(env, mode, dy, dgrad_weight(w), torch.zeros(64))
    prev = torch.randn(2, 8, 16, 64, generator=g).to(tdt).to(dev)
    for accumulate in (0, 1):
        dx = prev.clone()
        a = L.Pool2x2Args(dtype=code, B=2, H=8, W=16, C=64, du=du.data_ptr(), dx=dx.data_ptr(), accumulate=accumulate)
        L.check(lib.pd_pool2x2_sum(C.byref(a), stream()), "pd_pool2x2_sum")
        torch.cuda.synchronize()
        want = ref + (from_nhwc(prev) if accumulate else 0)
        assert rel(from_nhwc(dx), want) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("combined", [0, 1])
@pytest.mark.parametrize("cfg", [(2, 64, 0, 16, 16, 1), (2, 128, 64, 8, 8, 1), (1, 256, 256, 8, 4, 0), (3, 32, 32, 4, 4, 1),
                                 (2, 1280, 1280, 4, 4, 1), (1, 1280, 640, 8, 4, 1), (2, 1280, 0, 4, 4, 0)])
def test_groupnorm_silu_backward(env, mode, cfg, combined):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, c0, c1, H, W, silu = cfg
    Cc, hw = c0 + c1, H * W
    g = torch.Generator().manual_seed(14)
    x = bf16_round(torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.3, mode).requires_grad_(True)
    gamma = (torch.randn(Cc, generator=g) * 0.5 + 1.0).requires_grad_(True)
    beta = (torch.randn(Cc, generator=g) * 0.3).requires_grad_(True)
    dz = bf16_round(torch.randn(B, Cc, H, W, generator=g), mode)
    y = F.group_norm(x, 32, gamma, beta, eps=1e-5)
    z = F.silu(y) if silu else y
    rx, rg, rb = torch.autograd.grad(z, (x, gamma, beta), dz)

    xd = x.detach()
    X0, X1 = nhwc(xd[:, :c0].to(dev), tdt), (nhwc(xd[:, c0:].to(dev), tdt) if c1 else None)
    D0, D1 = nhwc(dz[:, :c0].to(dev), tdt), (nhwc(dz[:, c0:].to(dev), tdt) if c1 else None)
    res = None
    if combined:      # dz as one [C0+C1]-channel tensor + a skip gradient added to dx (the resnet-block use)
        D0, D1 = nhwc(dz.to(dev), tdt), None
        res = bf16_round(torch.randn(B, Cc, H, W, generator=g), mode)
        rx = rx + res
    RES = nhwc(res.to(dev), tdt) if res is not None else None
    splits = 4
    partial = torch.empty((B, splits, Cc, 2), dtype=torch.float64, device=dev)
    scale, shift = torch.empty((B, Cc), device=dev), torch.empty((B, Cc), device=dev)
    gm, bt = gamma.detach().to(dev), beta.detach().to(dev)
    a = L.GnStatsArgs(dtype=code, B=B, HW=hw, C0=c0, C1=c1, groups=32, eps=1e-5, x0=X0.data_ptr(), x1=L.ptr(X1),
                      gamma=gm.data_ptr(), beta=bt.data_ptr(), partial=partial.data_ptr(), splits=splits,
                      scale=scale.data_ptr(), shift=shift.data_ptr())
    if Cc <= 1024:      # the standalone statistics kernel (not used by the backward) is limited to the pixel-space widths
        L.check(lib.pd_gn_stats(C.byref(a), stream()), "pd_gn_stats")
    # mean / rstd per (sample, group) as pd_gn_finalize's optional outputs deliver them (exercised in the UNet tests)
    xg = xd.reshape(B, 32, -1).double()
    mean = xg.mean(-1).float().to(dev)
    rstd = (1.0 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-5)).float().to(dev)
    coef = torch.empty((B, 32, 2), device=dev)
    prev0 = torch.randn(B, H, W, c0, generator=g).to(tdt).to(dev)
    dx0 = prev0.clone()
    dx1 = torch.full((B, H, W, c1), float("nan"), dtype=tdt, device=dev) if c1 else None
    dgamma, dbeta = torch.ones(Cc, device=dev), torch.full((Cc,), 2.0, device=dev)
    b = L.GnBwdArgs(dtype=code, B=B, HW=hw, C0=c0, C1=c1, groups=32, silu=silu, x0=X0.data_ptr(), x1=L.ptr(X1),
                    dz0=D0.data_ptr(), dz1=L.ptr(D1), mean=mean.data_ptr(), rstd=rstd.data_ptr(), gamma=gm.data_ptr(),
                    beta=bt.data_ptr(), partial=partial.data_ptr(), splits=splits, coef=coef.data_ptr(),
                    dx0=dx0.data_ptr(), dx1=L.ptr(dx1), accumulate0=1, accumulate1=0, dgamma=dgamma.data_ptr(),
                    dbeta=dbeta.data_ptr(), dz_combined=int(bool(combined and c1)), res=L.ptr(RES))
    L.check(lib.pd_gn_silu_bwd(C.byref(b), stream()), "pd_gn_silu_bwd")
    torch.cuda.synchronize()
    tol = 2e-5 if mode == "f32" else TOL[mode]
    assert rel(from_nhwc(dx0), rx[:, :c0] + from_nhwc(prev0)) < tol
    if c1:
        assert rel(from_nhwc(dx1), rx[:, c0:]) < tol
    assert rel(dgamma.cpu() - 1.0, rg) < 2e-5 and rel(dbeta.cpu() - 2.0, rb) < 2e-5


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_channel_sum(env, mode):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(15)
    x = bf16_round(torch.randn(3, 96, 10, 12, generator=g), mode)
    X = nhwc(x.to(dev), tdt)
    out = torch.ones((3, 128), device=dev)
    a = L.ChannelSumArgs(dtype=code, B=3, HW=120, C=96, x=X.data_ptr(), out=out.data_ptr(), out_stride=128, accumulate=1)
    L.check(lib.pd_channel_sum(C.byref(a), stream()), "pd_channel_sum")
    torch.cuda.synchronize()
    assert rel(out.cpu()[:, :96] - 1.0, x.sum((2, 3))) < 1e-5
    assert torch.equal(out.cpu()[:, 96:], torch.ones(3, 32))


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("Cc", [2560, 10240])
def test_channel_sum_wide(env, mode, Cc):
    """More than 2048 channels (the GEGLU projection bias of the SD transformer blocks): walked in chunks; split-pixel form."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(16)
    x = bf16_round(torch.randn(2, Cc, 6, 5, generator=g), mode)
    X = nhwc(x.to(dev), tdt)
    out = torch.zeros((2, Cc), device=dev)
    tot = torch.ones(Cc, device=dev)
    ws = torch.empty(2 * 3 * Cc, device=dev)
    a = L.ChannelSumArgs(dtype=code, B=2, HW=30, C=Cc, x=X.data_ptr(), out=out.data_ptr(), out_stride=Cc, accumulate=0,
                         total=tot.data_ptr(), total_valid=Cc, workspace=ws.data_ptr(), splits=3)
    L.check(lib.pd_channel_sum(C.byref(a), stream()), "pd_channel_sum")
    torch.cuda.synchronize()
    assert rel(out.cpu(), x.sum((2, 3))) < 1e-5
    assert rel(tot.cpu() - 1.0, x.sum((0, 2, 3))) < 1e-5


def run_wgrad(env, mode, x0, dy, *, x1=None, ksize=3, stride=1, pad=1, upsample=0, silu=0, scale=None, shift=None,
              cout_valid=0, cin_valid=0, prev=None, slab_splits=None):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, c0, hin, win = x0.shape
    c1 = x1.shape[1] if x1 is not None else 0
    cout, hout, wout = dy.shape[1], dy.shape[2], dy.shape[3]
    X0 = nhwc(x0.to(dev), tdt)
    X1 = nhwc(x1.to(dev), tdt) if x1 is not None else None
    DY = nhwc(dy.to(dev), tdt)
    sc = scale.to(dev).float().contiguous() if scale is not None else None
    sh = shift.to(dev).float().contiguous() if shift is not None else None
    cov, civ = cout_valid or cout, cin_valid or (c0 + c1)
    dw = (prev.clone().to(dev) if prev is not None else torch.full((cov, civ, ksize, ksize), float("nan"), device=dev))
    a = L.WgradArgs(dtype=code, B=B, Hin=hin, Win=win, Hout=hout, Wout=wout, C0=c0, C1=c1, Cout=cout, ksize=ksize, stride=stride,
                    pad=pad, upsample=upsample, silu=silu, x0=X0.data_ptr(), x1=L.ptr(X1), scale=L.ptr(sc), shift=L.ptr(sh),
                    dy=DY.data_ptr(), dw=dw.data_ptr(), Cout_valid=cout_valid, Cin_valid=cin_valid,
                    accumulate=int(prev is not None))
    want = lib.pd_conv_wgrad_workspace(C.byref(a))
    assert want > 0
    nbytes = want if slab_splits is None else slab_splits * ksize * ksize * ((cout + 63) // 64 * 64) * ((c0 + c1 + 63) // 64 * 64) * 4
    slab = torch.empty(nbytes // 4, device=dev)
    a.slab, a.slab_bytes = slab.data_ptr(), nbytes
    L.check(lib.pd_conv_wgrad(C.byref(a), stream()), "pd_conv_wgrad")
    torch.cuda.synchronize()
    # round 6 (ABI 8): every call of this helper also runs the two launches separately -- stage 1 (GEMM -> slab), stage 2 (fold) on a second
    # stream behind an event -- and wants the same bits
    dw2 = (prev.clone().to(dev) if prev is not None else torch.full((cov, civ, ksize, ksize), float("nan"), device=dev))
    slab.fill_(float("nan"))
    a.dw, a.stage = dw2.data_ptr(), 1
    L.check(lib.pd_conv_wgrad(C.byref(a), stream()), "pd_conv_wgrad")
    ev, side = torch.cuda.Event(), torch.cuda.Stream()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    a.stage = 2
    L.check(lib.pd_conv_wgrad(C.byref(a), side.cuda_stream), "pd_conv_wgrad")
    side.synchronize()
    assert torch.equal(dw2, dw)
    return dw.cpu()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 32, 96, 16, 16), (3, 128, 64, 8, 8), (1, 64, 128, 40, 72), (2, 192, 32, 20, 12)])
def test_conv_weight_gradient_3x3(env, mode, shape):
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(21)
    x = bf16_round(torch.randn(B, cin, H, W, generator=g), mode)
    dy = bf16_round(torch.randn(B, cout, H, W, generator=g), mode)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, padding=1), w, dy)
    got = run_wgrad(env, mode, x, dy)
    assert rel(got, ref) < (2e-5 if mode == "f32" else 1e-4)      # bf16 inputs are exact here: only the fp32 sum order differs
    one_split = run_wgrad(env, mode, x, dy, slab_splits=1)
    assert rel(one_split, ref) < (2e-5 if mode == "f32" else 1e-4)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_conv_weight_gradient_fused_input_transform_and_concat(env, mode):
    B, c0, c1, cout, H, W = 2, 64, 32, 64, 24, 16
    g = torch.Generator().manual_seed(22)
    x = bf16_round(torch.randn(B, c0 + c1, H, W, generator=g), mode)
    scale, shift = torch.rand(B, c0 + c1, generator=g) + 0.5, torch.randn(B, c0 + c1, generator=g) * 0.3
    dy = bf16_round(torch.randn(B, cout, H, W, generator=g), mode)
    z = bf16_round(F.silu(x * scale[:, :, None, None] + shift[:, :, None, None]), mode)
    w = torch.zeros(cout, c0 + c1, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(z, w, None, padding=1), w, dy)
    prev = torch.randn(cout, c0 + c1, 3, 3, generator=g)
    got = run_wgrad(env, mode, x[:, :c0], dy, x1=x[:, c0:], silu=1, scale=scale, shift=shift, prev=prev)
    assert rel(got - prev, ref) < (2e-5 if mode == "f32" else 4e-3)    # bf16: fast SiLU + rounding of Z to bf16


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("hw", [(32, 32), (16, 48), (8, 8)])
def test_conv_weight_gradient_stride2(env, mode, hw):
    H, W = hw
    g = torch.Generator().manual_seed(23)
    x = bf16_round(torch.randn(2, 64, H, W, generator=g), mode)
    dy = bf16_round(torch.randn(2, 96, H // 2, W // 2, generator=g), mode)
    w = torch.zeros(96, 64, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, stride=2, padding=1), w, dy)
    got = run_wgrad(env, mode, x, dy, stride=2)
    assert rel(got, ref) < (2e-5 if mode == "f32" else 1e-4)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_conv_weight_gradient_upsample_and_1x1_and_padded_channels(env, mode):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(24)
    # conv fused with the nearest x2 upsample (Upsample2D)
    x = bf16_round(torch.randn(2, 64, 8, 12, generator=g), mode)
    dy = bf16_round(torch.randn(2, 64, 16, 24, generator=g), mode)
    w = torch.zeros(64, 64, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1), w, dy)
    assert rel(run_wgrad(env, mode, x, dy, upsample=1), ref) < (2e-5 if mode == "f32" else 1e-4)
    # 1x1 (attention projections, shortcuts): [q|k|v] 3C output channels
    x = bf16_round(torch.randn(2, 128, 8, 8, generator=g), mode)
    dy = bf16_round(torch.randn(2, 384, 8, 8, generator=g), mode)
    w = torch.zeros(384, 128, 1, 1, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w), w, dy)
    assert rel(run_wgrad(env, mode, x, dy, ksize=1, pad=0), ref) < (2e-5 if mode == "f32" else 1e-4)
    # conv_out: 3 real output channels in a 32-channel dy
    x = bf16_round(torch.randn(2, 64, 16, 16, generator=g), mode)
    dy = torch.zeros(2, 32, 16, 16)
    dy[:, :3] = bf16_round(torch.randn(2, 3, 16, 16, generator=g), mode)
    w = torch.zeros(3, 64, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(x, w, None, padding=1), w, dy[:, :3])
    assert rel(run_wgrad(env, mode, x, dy, cout_valid=3), ref) < (2e-5 if mode == "f32" else 1e-4)
    # conv_in: im2col3 + 1x1 over 27 (of 32) gathered channels
    img = torch.randn(2, 3, 20, 24, generator=g)
    dy = bf16_round(torch.randn(2, 64, 20, 24, generator=g), mode)
    cols = torch.empty((2, 20, 24, 32), dtype=tdt, device=dev)
    ximg = img.to(dev).contiguous()
    a = L.Im2col3Args(dtype=code, B=2, H=20, W=24, C=3, x=ximg.data_ptr(), out=cols.data_ptr())
    L.check(lib.pd_im2col3(C.byref(a), stream()), "pd_im2col3")
    torch.cuda.synchronize()
    w = torch.zeros(64, 3, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(bf16_round(img, mode), w, None, padding=1), w, dy)
    got = run_wgrad(env, mode, cols.float().cpu().permute(0, 3, 1, 2), dy, ksize=1, pad=0, cin_valid=27)
    assert rel(got.reshape(64, 3, 3, 3), ref) < (2e-5 if mode == "f32" else 1e-4)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("shape,splits", [((2, 64, 64, 8, 12), None), ((1, 96, 128, 33, 40), None), ((3, 64, 32, 32, 32), 1), ((2, 32, 64, 40, 70), 20)])
def test_upsampler_weight_gradient_through_four_subpixel_phases(env, mode, shape, splits):
    """pd_conv_wgrad(phase = 1 + 2 a + b): the gradient of Upsample2D's 3x3 weights through the sub-pixel form -- phase (a, b) multiplies the
    LOW-resolution input with the pixels (2 y + a, 2 x + b) of d out (a 2x2 weight gradient, 4 / 9 of the FLOPs of the upsample = 1 form) and
    adds each tap gradient to the 3x3 taps that tap is the sum of; the four launches in order (few- and many-split reductions, on top of a
    previous gradient) equal torch.autograd through conv2d(interpolate(x, nearest x2), w, padding=1)."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(29)
    x = bf16_round(torch.randn(B, cin, h, w_, generator=g), mode)
    dy = bf16_round(torch.randn(B, cout, 2 * h, 2 * w_, generator=g), mode)
    w = torch.zeros(cout, cin, 3, 3, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1), w, dy)
    prev = torch.randn(cout, cin, 3, 3, generator=g)
    X, DY = nhwc(x.to(dev), tdt), nhwc(dy.to(dev), tdt)
    dw = prev.clone().to(dev)
    for ph in range(4):
        a = L.WgradArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=h, Wout=w_, C0=cin, C1=0, Cout=cout, ksize=2, stride=1, pad=0, upsample=0, silu=0,
                        x0=X.data_ptr(), x1=None, scale=None, shift=None, dy=DY.data_ptr(), dw=dw.data_ptr(), Cout_valid=0, Cin_valid=0,
                        accumulate=1, phase=1 + ph)
        want = lib.pd_conv_wgrad_workspace(C.byref(a))
        assert want > 0
        nbytes = want if splits is None else splits * 4 * ((cout + 63) // 64 * 64) * ((cin + 63) // 64 * 64) * 4
        slab = torch.empty(nbytes // 4, device=dev)
        a.slab, a.slab_bytes = slab.data_ptr(), nbytes
        L.check(lib.pd_conv_wgrad(C.byref(a), stream()), "pd_conv_wgrad")
    torch.cuda.synchronize()
    assert rel(dw.cpu() - prev, ref) < (2e-5 if mode == "f32" else 1e-4)
    # phase 1 without `accumulate` SETS the taps it touches; refused: a phase with the fused upsample / another kernel size
    a.phase, a.accumulate = 1, 0
    dw2 = torch.full_like(dw, float("nan"))
    a.dw = dw2.data_ptr()
    L.check(lib.pd_conv_wgrad(C.byref(a), stream()), "pd_conv_wgrad")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dw2).all())
    a.upsample = 1
    assert lib.pd_conv_wgrad(C.byref(a), stream()) != 0
    a.upsample, a.ksize = 0, 3
    assert lib.pd_conv_wgrad(C.byref(a), stream()) != 0


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 4, 64), (1, 8, 1024), (2, 8, 200), (1, 2, 16), (1, 3, 300),
                                 # round 5, the one-pass form: one key block exactly, three ragged key blocks, ragged queries and keys, two blocks x batch
                                 (2, 2, 512), (1, 2, 1300), (1, 3, 600), (2, 4, 2048)])
def test_attention_backward(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, N = cfg
    Cc = heads * 8
    g = torch.Generator().manual_seed(31)
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g) * 1.2, mode).requires_grad_(True) for _ in range(3))
    dout = bf16_round(torch.randn(B, N, Cc, generator=g), mode)
    ref_o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, Cc)
    rq, rk, rv = torch.autograd.grad(ref_o, (q, k, v), dout)

    Q, K, V = (t.detach().to(tdt).to(dev).contiguous() for t in (q, k, v))
    out = torch.empty((B, N, Cc), dtype=tdt, device=dev)
    lse = torch.full((B, heads, N), float("nan"), device=dev)
    a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr(),
                   lse=lse.data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    scores = torch.einsum("bhqd,bhkd->bhqk", q.detach().double(), k.detach().double()) / 8 ** 0.5
    want_lse = torch.logsumexp(scores, -1) * 1.4426950408889634
    torch.cuda.synchronize()
    assert float((lse.cpu().double() - want_lse).abs().max()) < (1e-4 if mode == "f32" else 8e-2)   # bf16: q*scale is rounded to bf16 before the MFMA (forward and backward alike)

    DO = dout.to(tdt).to(dev).contiguous()
    delta = torch.empty((B, heads, N), device=dev)
    dqkv = torch.full((B, N, 3 * Cc), float("nan"), dtype=tdt, device=dev)
    b = L.AttnBwdArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), o=out.data_ptr(),
                      dout=DO.data_ptr(), lse=lse.data_ptr(), delta=delta.data_ptr(), dqkv=dqkv.data_ptr())
    L.check(lib.pd_attn_d8_bwd(C.byref(b), stream()), "pd_attn_d8_bwd")
    torch.cuda.synchronize()
    got = dqkv.float().cpu().reshape(B, N, 3, heads, 8).permute(2, 0, 3, 1, 4)     # [which][B][heads][N][8]
    tol = 3e-5 if mode == "f32" else 2.5e-2       # bf16: P, dS and the outputs are rounded to 8 mantissa bits
    for name, gg, rr in (("dq", got[0], rq), ("dk", got[1], rk), ("dv", got[2], rv)):
        assert rel(gg, rr) < tol, name
    # the one-pass form (round 5: a workspace for the per-key-block partial dQ; 16-bit engines, N >= 512) -- same gradients, twice the same bits
    need = int(lib.pd_attn_d8_bwd_workspace(C.byref(b)))
    assert (need > 0) == (mode != "f32" and N >= 512)
    if need:
        slab = torch.full((need // 4,), float("nan"), device=dev)
        b.slab, b.slab_bytes = slab.data_ptr(), need
        runs = []
        for _ in range(2):
            dqkv.fill_(float("nan"))
            L.check(lib.pd_attn_d8_bwd(C.byref(b), stream()), "pd_attn_d8_bwd")
            torch.cuda.synchronize()
            runs.append(dqkv.clone())
        assert torch.equal(runs[0], runs[1])
        got1 = runs[0].float().cpu().reshape(B, N, 3, heads, 8).permute(2, 0, 3, 1, 4)
        for name, gg, rr in (("dq", got1[0], rq), ("dk", got1[1], rk), ("dv", got1[2], rv)):
            assert rel(gg, rr) < tol, ("one pass", name)
        b.slab_bytes = need - 1                      # a workspace that is too small is not used: the two-kernel path answers
        dqkv.fill_(float("nan"))
        L.check(lib.pd_attn_d8_bwd(C.byref(b), stream()), "pd_attn_d8_bwd")
        torch.cuda.synchronize()
        assert torch.equal(dqkv.float().cpu().reshape(B, N, 3, heads, 8).permute(2, 0, 3, 1, 4), got)


def test_native_comm_world_one_allreduce(env):
    """pd_comm_* (RCCL behind the C ABI, csrc/comm_rccl.hip) on the one GPU a test box has: a one-rank communicator, where the sum /
    mean of a bucket is the bucket itself, through both forms (ncclAllReduce; reduce-scatter + all-gather: since round 4 the one-rank
    communicator really takes the reduce-scatter + all-gather branch and its in-place pointer arithmetic) and on a side stream --
    id, init, collective, destroy.  Two ranks per device are refused by RCCL: the multi-rank exchange itself is covered through
    torch.distributed (tests/test_gpu_two_rank_overlap.py, gloo) and stays unmeasured on hardware."""
    from phendiff_amd.comm import NativeComm
    cid = NativeComm.unique_id()
    assert len(cid) == 128 and any(cid)
    comm = NativeComm(0, 1, cid)
    assert comm.query() == (0, 1)               # rank / size as the communicator reports them (pd_comm_query: ncclCommUserRank / ncclCommCount)
    g = torch.Generator().manual_seed(71)
    x = torch.randn(1 << 20, generator=g).cuda()
    want = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for algo in (0, 1):
        for mean in (False, True):
            comm.allreduce_(x, mean=mean, algo=algo, stream=side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, want)
    with pytest.raises(Exception):
        comm.allreduce_(x.half())
    comm.close()
