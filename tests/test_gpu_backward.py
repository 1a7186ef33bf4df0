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
@pytest.mark.parametrize("cfg", [(2, 64, 0, 16, 16, 1), (2, 128, 64, 8, 8, 1), (1, 256, 256, 8, 4, 0), (3, 32, 32, 4, 4, 1)])
def test_groupnorm_silu_backward(env, mode, cfg):
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
    splits = 4
    partial = torch.empty((B, splits, Cc, 2), dtype=torch.float64, device=dev)
    scale, shift = torch.empty((B, Cc), device=dev), torch.empty((B, Cc), device=dev)
    gm, bt = gamma.detach().to(dev), beta.detach().to(dev)
    a = L.GnStatsArgs(dtype=code, B=B, HW=hw, C0=c0, C1=c1, groups=32, eps=1e-5, x0=X0.data_ptr(), x1=L.ptr(X1),
                      gamma=gm.data_ptr(), beta=bt.data_ptr(), partial=partial.data_ptr(), splits=splits,
                      scale=scale.data_ptr(), shift=shift.data_ptr())
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
                    dbeta=dbeta.data_ptr())
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
