"""Per-kernel parity on a real MI355X: every C-ABI entry point against plain PyTorch fp32 on the same seeded
inputs (the torch ops are exactly what the oracle's modules dispatch to).  fp32 mode must agree to fp32
round-off; bf16 mode to bf16 round-off of inputs/outputs (tolerances stated per test)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16), "fp16": (2, torch.float16)}
# relative L2 error bounds: fp32 MFMA is an exact fp32 fma chain; bf16 = 8-bit, fp16 = 11-bit mantissa on inputs and outputs
TOL = {"f32": 5e-6, "bf16": 5e-3, "fp16": 6e-4}     # ~2x the measured maxima (profiles/r2_parity_errors.json)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    from conftest import record_error
    return record_error(float((a - b).norm() / (b.norm() + 1e-30)))


@pytest.fixture(scope="module")
def env():
    import phendiff_amd._lib as L
    from phendiff_amd.packing import pack_conv_weight
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return L, L.lib(), pack_conv_weight, torch.device("cuda:0")


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc(x, tdt):  # NCHW fp32 -> NHWC dtype (device)
    return x.permute(0, 2, 3, 1).contiguous().to(tdt)


def run_conv(env, mode, x0, w, b, *, x1=None, ksize=3, stride=1, pad=1, upsample=0, silu=0, scale=None, shift=None,
             temb=None, residual=None, out_mode=0, heads=0, cout_pad=None):
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    B, _, hin, win = x0.shape
    cout = w.shape[0]
    cp = cout_pad or cout
    hc, wc = (2 * hin, 2 * win) if upsample else (hin, win)
    extra = 1 if (ksize == 3 and pad == 0) else 0
    hout = (hc + 2 * pad + extra - ksize) // stride + 1
    wout = (wc + 2 * pad + extra - ksize) // stride + 1
    X0 = nhwc(x0.to(dev), tdt)
    X1 = nhwc(x1.to(dev), tdt) if x1 is not None else None
    wp = pack(w.float().cpu(), tdt, cp).to(dev)
    bias = torch.zeros(cp)
    bias[:cout] = b
    bias = bias.to(dev)
    if out_mode == 0:
        y = torch.full((B, hout, wout, cout), float("nan"), dtype=tdt, device=dev)
    elif out_mode == 1:
        y = torch.full((B, cout, hout, wout), float("nan"), dtype=torch.float32, device=dev)
    else:
        y = torch.full((3, B, heads, hout * wout, 8), float("nan"), dtype=tdt, device=dev)
    sc = scale.to(dev).float().contiguous() if scale is not None else None
    sh = shift.to(dev).float().contiguous() if shift is not None else None
    tb = temb.to(dev).float().contiguous() if temb is not None else None
    res = nhwc(residual.to(dev), tdt) if residual is not None else None
    a = L.ConvArgs(dtype=code, B=B, Hin=hin, Win=win, Hout=hout, Wout=wout, C0=x0.shape[1],
                   C1=(x1.shape[1] if x1 is not None else 0), Cout=cout, Cout_pad=cp, ksize=ksize, stride=stride, pad=pad,
                   upsample=upsample, silu=silu, out_mode=out_mode, heads=heads, x0=X0.data_ptr(), x1=L.ptr(X1),
                   scale=L.ptr(sc), shift=L.ptr(sh), w_packed=wp.data_ptr(), bias=bias.data_ptr(), temb=L.ptr(tb),
                   temb_stride=(tb.shape[1] if tb is not None else 0), residual=L.ptr(res), y=y.data_ptr())
    L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    return y


def bf16_round(t, mode):
    """Round to the engine's storage dtype (bf16 / fp16; identity in the exact-fp32 mode)."""
    return t.to(DT[mode][1]).float() if mode != "f32" else t


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 32, 96, 16, 16), (3, 64, 32, 8, 8), (1, 64, 128, 40, 72),
                                   (2, 128, 64, 5, 7)])
def test_conv3x3_plain(env, mode, shape):
    B, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, h, w_, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    y = run_conv(env, mode, x, w, b)
    ref = F.conv2d(bf16_round(x, mode), bf16_round(w, mode), b, padding=1)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_conv3x3_fused_prologue_epilogue(env, mode):
    """GroupNorm-affine + SiLU prologue (zero padding AFTER the transform), concat of two sources,
    + bias + temb + residual epilogue: ResnetBlock2D.conv1 / conv2 as fused."""
    g = torch.Generator().manual_seed(2)
    B, c0, c1, cout, h, w_ = 2, 64, 32, 64, 32, 32
    x0, x1 = torch.randn(B, c0, h, w_, generator=g), torch.randn(B, c1, h, w_, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    scale, shift = torch.rand(B, c0 + c1, generator=g) + 0.5, torch.randn(B, c0 + c1, generator=g)
    temb = torch.randn(B, 200, generator=g)
    res = torch.randn(B, cout, h, w_, generator=g)
    y = run_conv(env, mode, x0, w, b, x1=x1, silu=1, scale=scale, shift=shift, temb=temb[:, 17:], residual=res)
    xin = torch.cat([bf16_round(x0, mode), bf16_round(x1, mode)], 1)
    xin = bf16_round(F.silu(xin * scale[:, :, None, None] + shift[:, :, None, None]), mode)
    ref = F.conv2d(xin, bf16_round(w, mode), b, padding=1) + temb[:, 17:17 + cout, None, None] + bf16_round(res, mode)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["bf16", "fp16", "f32"])
@pytest.mark.parametrize("tail", [False, True])
def test_conv3x3_two_output_tiles_per_workgroup(env, mode, tail):
    """A shape at which pd_conv's rounds-of-workgroups estimate picks the NCO = 2 kernel in the 16-bit engines (B = 16, 64 x 64,
    128 -> 256 channels behind a GroupNorm prologue: 1024 workgroups of the one-tile form = 1.33 rounds on a 256-CU chip, 512
    of the two-tile form = one whole round): prologue over a channel concat, bias + temb, residual or the fused 1x1 shortcut
    tail, and the per-tile GroupNorm statistics of both output tiles."""
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(31)
    B, c0, c1, cout, h, w_ = 16, 96, 32, 256, 64, 64
    x0, x1 = torch.randn(B, c0, h, w_, generator=g), torch.randn(B, c1, h, w_, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    scale, shift = torch.rand(B, c0 + c1, generator=g) + 0.5, torch.randn(B, c0 + c1, generator=g)
    temb = torch.randn(B, 300, generator=g)
    xin = torch.cat([bf16_round(x0, mode), bf16_round(x1, mode)], 1)
    xin = bf16_round(F.silu(xin * scale[:, :, None, None] + shift[:, :, None, None]), mode)
    ref = F.conv2d(xin, bf16_round(w, mode), b, padding=1) + temb[:, 9:9 + cout, None, None]
    X0, X1 = nhwc(x0.to(dev), tdt), nhwc(x1.to(dev), tdt)
    sc, sh, tb = scale.to(dev), shift.to(dev), temb[:, 9:].contiguous().to(dev)
    y = torch.full((B, h, w_, cout), float("nan"), dtype=tdt, device=dev)
    T = lib.pd_conv_stat_tiles(h, w_, 3, 1)
    st = torch.full((B, T, cout, 2), float("nan"), device=dev)
    a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=h, Wout=w_, C0=c0, C1=c1, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1,
                   upsample=0, silu=1, out_mode=0, heads=0, x0=X0.data_ptr(), x1=X1.data_ptr(), scale=sc.data_ptr(), shift=sh.data_ptr(),
                   bias=None, temb=tb.data_ptr(), temb_stride=tb.shape[1], residual=None, y=y.data_ptr(), stats_out=st.data_ptr(), im2col3=0)
    if tail:
        t0 = 64
        xa = torch.randn(B, t0, h, w_, generator=g)
        ws = torch.randn(cout, t0, 1, 1, generator=g) / t0 ** 0.5
        bs = torch.randn(cout, generator=g)
        p2, ps = pack(w, tdt), pack(ws, tdt)
        ct = p2.shape[0]
        wp = torch.cat([p2.reshape(ct, -1, 64, 8), ps.reshape(ct, -1, 64, 8)], 1).contiguous().to(dev)
        XA = nhwc(xa.to(dev), tdt)
        bias = (b + bs).to(dev)
        a.tail_x0, a.tail_x1, a.tail_C0, a.tail_C1 = XA.data_ptr(), None, t0, 0
        ref = ref + F.conv2d(bf16_round(xa, mode), bf16_round(ws, mode), bs)
    else:
        res = torch.randn(B, cout, h, w_, generator=g)
        R = nhwc(res.to(dev), tdt)
        wp = pack(w, tdt).to(dev)
        bias = b.to(dev)
        a.residual = R.data_ptr()
        ref = ref + bf16_round(res, mode)
    a.w_packed, a.bias = wp.data_ptr(), bias.data_ptr()
    L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    got = y.float().permute(0, 3, 1, 2)
    assert rel(got, ref) < TOL[mode]
    # statistics of what was stored: per (sample, channel) sums over all tiles
    yc = got.cpu().double()
    assert rel(st[..., 0].sum(1).cpu(), yc.sum((2, 3))) < 1e-4 and rel(st[..., 1].sum(1).cpu(), (yc * yc).sum((2, 3))) < 1e-4


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("hw", [(32, 32), (16, 16), (64, 96), (8, 8)])
def test_conv3x3_stride2(env, mode, hw):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, *hw, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    y = run_conv(env, mode, x, w, b, stride=2, pad=1)
    ref = F.conv2d(bf16_round(x, mode), bf16_round(w, mode), b, stride=2, padding=1)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]
    # Downsample2D(padding=0): zero-pad (0,1,0,1) then pad-0 stride-2 conv
    y = run_conv(env, mode, x, w, b, stride=2, pad=0)
    ref = F.conv2d(F.pad(bf16_round(x, mode), (0, 1, 0, 1)), bf16_round(w, mode), b, stride=2, padding=0)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("hw", [(16, 16), (8, 8), (32, 48), (4, 4)])
def test_conv3x3_upsample(env, mode, hw):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 64, *hw, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g)
    y = run_conv(env, mode, x, w, b, upsample=1)
    ref = F.conv2d(F.interpolate(bf16_round(x, mode), scale_factor=2.0, mode="nearest"), bf16_round(w, mode), b, padding=1)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_conv1x1_and_output_modes(env, mode):
    g = torch.Generator().manual_seed(5)
    B, cin, h, w_ = 2, 64, 16, 16
    x = torch.randn(B, cin, h, w_, generator=g)
    # 1x1 shortcut with residual
    w = torch.randn(96, cin, 1, 1, generator=g) / 8.0
    b = torch.randn(96, generator=g)
    y = run_conv(env, mode, x, w, b, ksize=1, pad=0)
    ref = F.conv2d(bf16_round(x, mode), bf16_round(w, mode), b)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]
    # conv_out: Cout=3 padded to 32, NCHW fp32 output
    w3 = torch.randn(3, cin, 3, 3, generator=g) / 24.0
    b3 = torch.randn(3, generator=g)
    y = run_conv(env, mode, x, w3, b3, out_mode=1, cout_pad=32)
    ref = F.conv2d(bf16_round(x, mode), bf16_round(w3, mode), b3, padding=1)
    assert rel(y, ref) < TOL[mode]
    # QKV projection, head-major output [3][B][heads][N][8]
    heads = cin // 8
    wq = torch.randn(3 * cin, cin, 1, 1, generator=g) / 8.0
    bq = torch.randn(3 * cin, generator=g)
    y = run_conv(env, mode, x, wq, bq, ksize=1, pad=0, out_mode=2, heads=heads)
    ref = F.conv2d(bf16_round(x, mode), bf16_round(wq, mode), bq)  # [B, 3C, h, w]
    ref = ref.view(B, 3, heads, 8, h * w_).permute(1, 0, 2, 4, 3)
    assert rel(y.float(), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 64, 0, 32 * 32), (2, 256, 128, 16 * 16), (1, 128, 64, 8 * 8), (3, 256, 256, 64)])
def test_gn_stats(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, c0, c1, hw = cfg
    g = torch.Generator().manual_seed(6)
    Cc = c0 + c1
    x = torch.randn(B, Cc, hw, generator=g) * 2 + 0.7
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xs = bf16_round(x, mode)
    X0 = xs[:, :c0].permute(0, 2, 1).contiguous().to(tdt).to(dev)
    X1 = xs[:, c0:].permute(0, 2, 1).contiguous().to(tdt).to(dev) if c1 else None
    splits = 4
    partial = torch.empty((B, splits, Cc, 2), dtype=torch.float64, device=dev)
    scale = torch.empty((B, Cc), device=dev)
    shift = torch.empty((B, Cc), device=dev)
    gm, bt = gamma.to(dev), beta.to(dev)
    a = L.GnStatsArgs(dtype=code, B=B, HW=hw, C0=c0, C1=c1, groups=32, eps=1e-5, x0=X0.data_ptr(), x1=L.ptr(X1),
                      gamma=gm.data_ptr(), beta=bt.data_ptr(), partial=partial.data_ptr(), splits=splits,
                      scale=scale.data_ptr(), shift=shift.data_ptr())
    L.check(lib.pd_gn_stats(C.byref(a), stream()), "pd_gn_stats")
    torch.cuda.synchronize()
    ref = F.group_norm(xs, 32, gamma, beta, eps=1e-5)
    got = xs * scale.cpu()[:, :, None] + shift.cpu()[:, :, None]
    assert rel(got, ref) < 1e-5


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 4, 64), (1, 32, 1024), (2, 8, 200), (1, 2, 16),
                                 (8, 32, 1024), (4, 32, 2100)])      # the last two: 8-wave workgroups in bf16 (even / ragged N)
def test_attention(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, N = cfg
    g = torch.Generator().manual_seed(7)
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g) * 1.5, mode) for _ in range(3))
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    out = torch.full((B, N, heads * 8), float("nan"), dtype=tdt, device=dev)
    a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, heads * 8)
    assert rel(out.float(), ref) < {"f32": 5e-6, "bf16": 8e-3, "fp16": 1e-3}[mode]      # measured 1.6e-6 / 3.9e-3 / 4.9e-4


def test_attention_online_softmax_rescale(env):
    """Force the running-max rescale branch: one key far above the rest late in the sequence (guide rule 26)."""
    L, lib, _, dev = env
    g = torch.Generator().manual_seed(8)
    B, heads, N = 1, 2, 512
    q, k, v = (torch.randn(B, heads, N, 8, generator=g) for _ in range(3))
    k[:, :, 300] = q[:, :, 5] * 6.0   # spikes the score of query 5 (and others) at key 300
    k[:, :, 77] = q[:, :, 130] * 9.0
    Q, K, V = (t.to(dev).contiguous() for t in (q, k, v))
    out = torch.empty((B, N, heads * 8), device=dev)
    a = L.AttnArgs(dtype=0, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).transpose(1, 2).reshape(B, N, heads * 8)
    assert float((out.cpu().double() - ref).abs().max()) < 1e-4


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_attention_rescale_8_wave_workgroups(env, mode):
    """The same spikes through the 16-bit 8-wave variants (two halves of the workgroup stage alternate key tiles).  fp16 keeps
    p = 2^(s - m) below 2^14 (its rescale threshold): a score 2^16 above the running maximum would be inf in fp16."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(18)
    B, heads, N = 4, 32, 2048
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g), mode) for _ in range(3))
    k[:, :, 1900] = q[:, :, 5] * 6.0
    k[:, :, 300] = q[:, :, 1030] * 9.0
    k[:, :, 777] = q[:, :, 2047] * 4.0
    q, k = bf16_round(q, mode), bf16_round(k, mode)
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    out = torch.full((B, N, heads * 8), float("nan"), dtype=tdt, device=dev)
    a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, heads * 8)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) < (8e-3 if mode == "bf16" else 1e-3)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(8, 32, 1024), (4, 32, 2100), (2, 64, 4096)])
def test_attention_dma_staged_with_producer_key_bound(env, mode, cfg):
    """pd_attn_args.kmax2 (max |k|^2 per (sample, head), what pd_linear's kmax2_out emits) selects the DMA-staged kernel:
    K / V tiles by global_load_lds, V^T fragments by transposed LDS reads, no per-tile key norms.  Same function of the
    inputs as the register-staged kernel: compared with SDPA at its tolerance and with that kernel's output."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, N = cfg
    g = torch.Generator().manual_seed(27)
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g) * 1.5, mode) for _ in range(3))
    k[:, :, N - 100] = q[:, :, 5] * 6.0              # late spikes: the rescale path of the new kernel
    k[:, 1, 300] = q[:, 1, N - 1] * 9.0
    q, k = bf16_round(q, mode), bf16_round(k, mode)
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    kmax2 = (K.float() ** 2).sum(-1).amax(-1).contiguous()          # [B][heads]
    outs = []
    for km in (kmax2, None):
        out = torch.full((B, N, heads * 8), float("nan"), dtype=tdt, device=dev)
        a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr(), kmax2=L.ptr(km))
        L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, heads * 8)
    assert torch.isfinite(outs[0]).all()
    tol = 8e-3 if mode == "bf16" else 1e-3
    assert rel(outs[0], ref) < tol
    assert rel(outs[0], outs[1]) < tol            # the two kernels differ only in the reference maxima they pick


def test_attention_dma_staged_loose_bound_and_lse(env):
    """A bound far above the true max |k| (the slow path runs for every tile: exact maxima, rescales) and the log-sum-exp
    output of the training forward."""
    L, lib, _, dev = env
    code, tdt = DT["bf16"]
    B, heads, N = 4, 32, 1024
    g = torch.Generator().manual_seed(28)
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g), "bf16") for _ in range(3))
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    kmax2 = torch.full((B, heads), 1e6, device=dev)
    out = torch.empty((B, N, heads * 8), dtype=tdt, device=dev)
    lse = torch.empty((B, heads, N), device=dev)
    a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr(),
                   lse=lse.data_ptr(), kmax2=kmax2.data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, heads * 8)
    assert rel(out.float(), ref) < 8e-3
    s = (q @ k.transpose(-1, -2)) * (8 ** -0.5) * 1.4426950408889634
    ref_lse = torch.logsumexp(s * 0.6931471805599453, -1) * 1.4426950408889634      # log2-domain
    assert float((lse.cpu() - ref_lse).abs().max()) < 2e-2


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_attention_dma_staged_zero_key_bound_falls_back(env, mode):
    """A kmax2 slot that under-estimates max |k|^2 -- here ZERO, what a caller that skipped pd_linear / reused a pd_zero'ed slot
    would pass -- violates the documented precondition; the kernel's safety net (a first-sub-tile score above the implied bound
    -> exact running-maximum path) must still return finite, correct probabilities instead of an unbounded 2^(s - m) (fp16: inf).
    Keys grow towards the end of the sequence so that the late scores exceed the first tile's by far more than the fp16 rescale
    threshold."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, N = 4, 32, 2048
    g = torch.Generator().manual_seed(29)
    q, k, v = (bf16_round(torch.randn(B, heads, N, 8, generator=g), mode) for _ in range(3))
    k = bf16_round(k * torch.linspace(0.5, 12.0, N).view(1, 1, N, 1), mode)
    q = bf16_round(q * 3.0, mode)
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    out = torch.full((B, N, heads * 8), float("nan"), dtype=tdt, device=dev)
    a = L.AttnArgs(dtype=code, B=B, heads=heads, N=N, q=Q.data_ptr(), k=K.data_ptr(), v=V.data_ptr(), out=out.data_ptr(),
                   kmax2=torch.zeros((B, heads), device=dev).data_ptr())
    L.check(lib.pd_attn_d8(C.byref(a), stream()), "pd_attn_d8")
    torch.cuda.synchronize()
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, heads * 8)
    assert torch.isfinite(out.float()).all()
    assert rel(out.float(), ref) < (8e-3 if mode == "bf16" else 1e-3)


def test_conv_in(env):
    L, lib, _, dev = env
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 24, 40, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    for mode in ("f32", "bf16", "fp16"):
        code, tdt = DT[mode]
        y = torch.empty((2, 24, 40, 64), dtype=tdt, device=dev)
        X, W_, Bb = x.to(dev), w.to(dev), b.to(dev)
        a = L.ConvInArgs(dtype=code, B=2, H=24, W=40, Cin=3, Cout=64, x=X.data_ptr(), w=W_.data_ptr(), bias=Bb.data_ptr(), y=y.data_ptr())
        L.check(lib.pd_conv_in(C.byref(a), stream()), "pd_conv_in")
        torch.cuda.synchronize()
        assert rel(y.float().permute(0, 3, 1, 2), ref) < (1e-6 if mode == "f32" else 4e-3)


@pytest.mark.parametrize("pt", ["epsilon", "sample", "v_prediction"])
def test_ddim_step_matches_oracle(env, pt):
    from oracle import DDIMSchedulerRef, DDIMInverseSchedulerRef
    import phendiff_amd as P
    dev = env[3]
    cfg = dict(P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"], prediction_type=pt)
    g = torch.Generator().manual_seed(10)
    x, out = torch.randn(2, 3, 16, 16, generator=g), torch.randn(2, 3, 16, 16, generator=g)
    for S in (4, 50):
        ref, got = DDIMSchedulerRef(**cfg), P.DDIMScheduler(**cfg)
        ref.set_timesteps(S); got.set_timesteps(S)
        iref, igot = DDIMInverseSchedulerRef.from_config(ref.config), P.DDIMInverseScheduler.from_config(got.config)
        iref.set_timesteps(S); igot.set_timesteps(S)
        for r_, g_ in ((ref, got), (iref, igot)):
            assert torch.equal(r_.timesteps, g_.timesteps)      # bit-exact step indexing
            for t in r_.timesteps[1:-1:max(1, S // 4)].tolist() + [int(r_.timesteps[-1])]:
                a = r_.step(out, torch.tensor(t), x)
                b = g_.step(out.to(dev), torch.tensor(t), x.to(dev))
                assert torch.allclose(b.prev_sample.cpu(), a.prev_sample, rtol=1e-5, atol=1e-6), (pt, S, t)
                assert torch.allclose(b.pred_original_sample.cpu(), a.pred_original_sample, rtol=1e-5, atol=1e-6)
    ts = torch.tensor([5, 1500])
    noise = torch.randn(2, 3, 16, 16, generator=g)
    assert torch.allclose(got.add_noise(x.to(dev), noise.to(dev), ts).cpu(), ref.add_noise(x, noise, ts), rtol=1e-6, atol=1e-6)
    assert torch.allclose(got.get_velocity(x.to(dev), noise.to(dev), ts).cpu(), ref.get_velocity(x, noise, ts), rtol=1e-6, atol=1e-6)


def test_temb(env):
    from oracle import CondUNet2DRef
    import phendiff_amd as P
    dev = env[3]
    torch.manual_seed(0)
    cfg = {k: v for k, v in P.UNET_CONFIGS["super_small"].items() if k in CondUNet2DRef.__init__.__code__.co_varnames}
    r = CondUNet2DRef(**dict(cfg, sample_size=32)).eval()
    m = P.CustomCondUNet2DModel(compute_dtype="f32", **dict(P.UNET_CONFIGS["super_small"], sample_size=32))
    m.load_state_dict(r.state_dict())
    m.to(dev)
    plan = m.plan_for(3, 32, 32, dev)
    ts = torch.tensor([2999.0, 59.0, 1234.0], device=dev)
    labels = torch.tensor([0, 1, 1], device=dev)
    got = plan.temb_rows(ts, labels, None, stream())
    torch.cuda.synchronize()
    with torch.no_grad():
        embs = torch.stack([r.embed(1, int(t), torch.tensor([int(l)]))[0] for t, l in zip(ts.cpu(), labels.cpu())])
        ref = torch.cat([res.time_emb_proj(F.silu(embs)) for _, res in r.named_modules() if hasattr(res, "time_emb_proj")], 1)
    assert got.shape == ref.shape == (3, 2752)
    assert rel(got, ref) < 2e-5


@pytest.mark.parametrize("rows,with_emb", [(11, True), (11, False), (3, True)])
def test_temb_wide_projection_stack(env, rows, with_emb):
    """pd_temb at SD width (tdim 1280, > 4 M projection weights): with the `emb` hand-over buffer the second layer is split
    over tdim / 256 blocks per row and the projections run 8 rows per block; without it one block per row does everything.
    Both against a CPU fp64 restatement of Timesteps -> TimestepEmbedding -> silu -> stacked time_emb_proj."""
    import math
    L, lib, _, dev = env
    g = torch.Generator().manual_seed(21)
    c0, tdim, pdim = 320, 1280, 3400                   # 3400: not a multiple of 256
    w1, w2, wp = (torch.randn(i, o, generator=g) / math.sqrt(i) for i, o in ((c0, tdim), (tdim, tdim), (tdim, pdim)))
    b1, b2, bp = (torch.randn(n, generator=g) * 0.1 for n in (tdim, tdim, pdim))
    ts = torch.randint(0, 1000, (rows,), generator=g).float()
    D = [t.to(dev).contiguous() for t in (w1, b1, w2, b2, wp, bp, ts)]
    emb = torch.full((rows, tdim), float("nan"), device=dev) if with_emb else None
    proj = torch.full((rows, pdim), float("nan"), device=dev)
    z1 = torch.full((rows, tdim), float("nan"), device=dev)
    feat = torch.full((rows, c0), float("nan"), device=dev)
    a = L.TembArgs(rows=rows, c0=c0, tdim=tdim, proj_dim=pdim, flip_sin_to_cos=1, freq_shift=0.0, num_classes=0,
                   timesteps=D[6].data_ptr(), labels=None, class_emb=None, w1=D[0].data_ptr(), b1=D[1].data_ptr(),
                   w2=D[2].data_ptr(), b2=D[3].data_ptr(), class_table=None, wp=D[4].data_ptr(), bp=D[5].data_ptr(),
                   emb=L.ptr(emb), proj=proj.data_ptr(), feat=feat.data_ptr(), z1=z1.data_ptr())
    L.check(lib.pd_temb(C.byref(a), stream()), "pd_temb")
    torch.cuda.synchronize()
    half = c0 // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float64) / half)
    arg = ts.double()[:, None] * freqs[None]
    se = torch.cat([arg.cos(), arg.sin()], 1)          # flip_sin_to_cos
    z1_ref = se @ w1.double() + b1.double()
    emb_ref = F.silu(z1_ref) @ w2.double() + b2.double()
    proj_ref = F.silu(emb_ref) @ wp.double() + bp.double()
    assert rel(feat.cpu().double(), se) < 1e-4         # fp32 sin / cos of arguments up to 1000
    assert rel(z1.cpu().double(), z1_ref) < 1e-4
    if with_emb:
        assert rel(emb.cpu().double(), emb_ref) < 1e-4
    assert rel(proj.cpu().double(), proj_ref) < 1e-4


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 40, 72, 3, 1), (1, 32, 96, 16, 16, 3, 1), (2, 64, 64, 32, 32, 3, 2), (2, 64, 128, 8, 8, 1, 1)])
def test_conv_fused_gn_statistics(env, mode, shape):
    """pd_conv(stats_out) + pd_gn_finalize == GroupNorm statistics of the stored conv output (also for a channel
    concat of two producers, and with groups straddling the two sources: 64 + 32 channels -> 3 per group)."""
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    B, cin, cout, h, w_, ks, stride = shape
    g = torch.Generator().manual_seed(11)
    outs, stats, tiles = [], [], []
    for co in (cout, 32):
        x = torch.randn(B, cin, h, w_, generator=g) + 0.3
        w = torch.randn(co, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        b = torch.randn(co, generator=g)
        pad = 1 if ks == 3 else 0
        ho, wo = (h + 2 * pad - ks) // stride + 1, (w_ + 2 * pad - ks) // stride + 1
        T = lib.pd_conv_stat_tiles(ho, wo, ks, stride)
        X = nhwc(x.to(dev), tdt); wp = pack(w, tdt).to(dev); bb = b.to(dev)
        y = torch.empty((B, ho, wo, co), dtype=tdt, device=dev)
        st = torch.full((B, T, co, 2), float("nan"), device=dev)
        a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=ho, Wout=wo, C0=cin, C1=0, Cout=co, Cout_pad=co, ksize=ks,
                       stride=stride, pad=pad, upsample=0, silu=0, out_mode=0, heads=0, x0=X.data_ptr(), x1=None, scale=None,
                       shift=None, w_packed=wp.data_ptr(), bias=bb.data_ptr(), temb=None, temb_stride=0, residual=None,
                       y=y.data_ptr(), stats_out=st.data_ptr(), im2col3=0)
        L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
        outs.append(y); stats.append(st); tiles.append(T)
    torch.cuda.synchronize()
    ycat = torch.cat([o.float().permute(0, 3, 1, 2) for o in outs], 1).cpu()
    Cc = ycat.shape[1]
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    gm, bt = gamma.to(dev), beta.to(dev)
    scale, shift = torch.empty((B, Cc), device=dev), torch.empty((B, Cc), device=dev)
    a = L.GnFinalizeArgs(B=B, HW=ycat.shape[2] * ycat.shape[3], groups=32, eps=1e-5, C0=cout, T0=tiles[0], stats0=stats[0].data_ptr(),
                         C1=32, T1=tiles[1], stats1=stats[1].data_ptr(), gamma=gm.data_ptr(), beta=bt.data_ptr(),
                         scale=scale.data_ptr(), shift=shift.data_ptr())
    L.check(lib.pd_gn_finalize(C.byref(a), stream()), "pd_gn_finalize")
    torch.cuda.synchronize()
    ref = F.group_norm(ycat, 32, gamma, beta, eps=1e-5)
    got = ycat * scale.cpu()[:, :, None, None] + shift.cpu()[:, :, None, None]
    assert rel(got, ref) < 2e-5


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("hw", [(24, 40), (32, 32), (8, 8), (64, 64)])
def test_conv_in_im2col_mode(env, mode, hw):
    """conv_in (cond_unet_2d.py:127-129) as pd_conv(im2col3): NCHW fp32 sample -> NHWC, + output statistics."""
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(12)
    h, w_ = hw
    x = torch.randn(2, 3, h, w_, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) / 5
    b = torch.randn(64, generator=g)
    wv = torch.zeros(64, 32, 1, 1)
    wv[:, :27, 0, 0] = w.reshape(64, 27)
    wp = pack(wv, tdt).to(dev)
    X, bb = x.to(dev), b.to(dev)
    y = torch.empty((2, h, w_, 64), dtype=tdt, device=dev)
    a = L.ConvArgs(dtype=code, B=2, Hin=h, Win=w_, Hout=h, Wout=w_, C0=32, C1=0, Cout=64, Cout_pad=64, ksize=1, stride=1, pad=0,
                   upsample=0, silu=0, out_mode=0, heads=0, x0=X.data_ptr(), x1=None, scale=None, shift=None,
                   w_packed=wp.data_ptr(), bias=bb.data_ptr(), temb=None, temb_stride=0, residual=None, y=y.data_ptr(),
                   stats_out=None, im2col3=3)
    L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    ref = F.conv2d(bf16_round(x, mode), bf16_round(w, mode), b, padding=1)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 96, 32, 64, 32, 32), (1, 128, 64, 0, 64, 16, 16), (2, 32, 32, 32, 96, 8, 8)])
@pytest.mark.parametrize("plain", [False, True])
def test_conv3x3_with_fused_1x1_tail(env, mode, shape, plain):
    """ResnetBlock2D tail: conv2(silu(gn(h))) + conv_shortcut(cat[x0, x1]) in ONE pd_conv (tail chunks).  plain: no GroupNorm / SiLU
    prologue (the latent-diffusion UNet applies its GroupNorms with pd_gn_apply) -- in the 16-bit engines the prologue-free
    instantiations multiply with 16x16x32 MFMAs on tiles of width >= 16."""
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    B, cm, t0, t1, cout, h, w_ = shape
    g = torch.Generator().manual_seed(13)
    hmid = torch.randn(B, cm, h, w_, generator=g)
    xa = torch.randn(B, t0, h, w_, generator=g)
    xb = torch.randn(B, t1, h, w_, generator=g) if t1 else None
    w2 = torch.randn(cout, cm, 3, 3, generator=g) / (cm * 9) ** 0.5
    ws = torch.randn(cout, t0 + t1, 1, 1, generator=g) / (t0 + t1) ** 0.5
    b2, bs = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    scale, shift = torch.rand(B, cm, generator=g) + 0.5, torch.randn(B, cm, generator=g)
    p2, ps = pack(w2, tdt), pack(ws, tdt)
    ct = p2.shape[0]
    wp = torch.cat([p2.reshape(ct, -1, 64, 8), ps.reshape(ct, -1, 64, 8)], 1).contiguous().to(dev)
    H_, XA = nhwc(hmid.to(dev), tdt), nhwc(xa.to(dev), tdt)
    XB = nhwc(xb.to(dev), tdt) if xb is not None else None
    bias = (b2 + bs).to(dev)
    sc, sh = scale.to(dev), shift.to(dev)
    y = torch.full((B, h, w_, cout), float("nan"), dtype=tdt, device=dev)
    a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=h, Wout=w_, C0=cm, C1=0, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1,
                   upsample=0, silu=0 if plain else 1, out_mode=0, heads=0, x0=H_.data_ptr(), x1=None,
                   scale=None if plain else sc.data_ptr(), shift=None if plain else sh.data_ptr(),
                   w_packed=wp.data_ptr(), bias=bias.data_ptr(), temb=None, temb_stride=0, residual=None, y=y.data_ptr(),
                   stats_out=None, tail_x0=XA.data_ptr(), tail_x1=L.ptr(XB), tail_C0=t0, tail_C1=t1, im2col3=0)
    L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    hin = bf16_round(hmid, mode) if plain else bf16_round(F.silu(bf16_round(hmid, mode) * scale[:, :, None, None] + shift[:, :, None, None]), mode)
    xcat = torch.cat([bf16_round(xa, mode)] + ([bf16_round(xb, mode)] if xb is not None else []), 1)
    ref = F.conv2d(hin, bf16_round(w2, mode), b2, padding=1) + F.conv2d(xcat, bf16_round(ws, mode), bs)
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 128, 96, 40, 72), (2, 32, 64, 9, 33), (2, 64, 64, 16, 16), (1, 32, 64, 20, 17)])      # the last two: 16 x 16 tiles
def test_upsample_conv_as_four_subpixel_phases(env, mode, shape):
    """Round 4: Upsample2D (F.interpolate(nearest x2) -> conv 3x3 pad 1; diffusers resnet.py, cond_unet_2d.py:200-228) as four 2x2
    convolutions over the LOW-resolution tensor (pd_conv phase 1..4 with packing.upsample_phase_weights): every output pixel equals
    the 3x3 convolution over the upsampled tensor (4 / 9 of its multiply-adds), the image border included (odd sizes, partial tiles),
    and the four launches together leave the output's GroupNorm statistic tiles."""
    from phendiff_amd.packing import upsample_phase_weights
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    B, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, cin, h, w_, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(F.interpolate(bf16_round(x, mode), scale_factor=2.0, mode="nearest"), w, b, padding=1)
    X = nhwc(x.to(dev), tdt)
    y = torch.full((B, 2 * h, 2 * w_, cout), float("nan"), dtype=tdt, device=dev)
    T = lib.pd_conv_stat_tiles(h, w_, 2, 1)
    st = torch.full((B, 4 * T, cout, 2), float("nan"), device=dev)
    bias = b.to(dev)
    keep = []
    for ph, k in enumerate(upsample_phase_weights(w)):
        wp = pack(k, tdt).to(dev)
        keep.append(wp)
        a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=h, Wout=w_, C0=cin, C1=0, Cout=cout, Cout_pad=cout, ksize=2, stride=1, pad=0,
                       upsample=0, silu=0, out_mode=0, heads=0, x0=X.data_ptr(), x1=None, scale=None, shift=None, w_packed=wp.data_ptr(),
                       bias=bias.data_ptr(), temb=None, temb_stride=0, residual=None, y=y.data_ptr(), stats_out=st.data_ptr(), im2col3=0,
                       phase=1 + ph)
        L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    got = y.float().permute(0, 3, 1, 2)
    assert bool(torch.isfinite(got).all())                      # every pixel of the upsampled tensor was written by exactly one phase
    # (the phase weights are sums of up to four 16-bit-rounded taps: one rounding more than the 3x3 form)
    assert rel(got, ref) < {"f32": 2e-6, "bf16": 5e-3, "fp16": 6e-4}[mode]      # measured 5.2e-7 / 2.1e-3 / 2.6e-4 (profiles/r4_parity_errors.json)
    yc = got.cpu().double()
    assert bool(torch.isfinite(st).all())
    assert rel(st[..., 0].sum(1).cpu(), yc.sum((2, 3))) < 1e-4 and rel(st[..., 1].sum(1).cpu(), (yc * yc).sum((2, 3))) < 1e-4
    # refused: a phase with a GroupNorm prologue / another kernel size / Hout != Hin
    a.ksize = 3
    assert lib.pd_conv(C.byref(a), stream()) != 0
    a.ksize, a.Hout = 2, h + 1
    assert lib.pd_conv(C.byref(a), stream()) != 0


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 32), (1, 96, 128, 40, 72), (2, 64, 32, 9, 33), (2, 64, 64, 16, 16), (1, 64, 32, 20, 17)])
def test_upsample_conv_input_gradient_as_four_subpixel_phases(env, mode, shape):
    """The input gradient of Upsample2D's convolution through the sub-pixel form (pd_conv phase = 1 + 2 a + b with phase_in = 1): phase
    (a, b) reads the pixels (2 y + a, 2 x + b) of d out and adds a 2x2 convolution of them -- transposed phase kernel, taps flipped,
    rows starting at y - a -- into the gradient of the LOW-resolution tensor; the four launches together equal torch.autograd through
    conv2d(interpolate(x, nearest x2), w, padding=1), on top of what the buffer already held (`residual`)."""
    from phendiff_amd.packing import dgrad_weight, upsample_phase_weights
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    B, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(43)
    x = torch.randn(B, cin, h, w_, generator=g).requires_grad_(True)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    dy = bf16_round(torch.randn(B, cout, 2 * h, 2 * w_, generator=g), mode)
    prev = bf16_round(torch.randn(B, cin, h, w_, generator=g), mode)          # what the gradient buffer already holds
    F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, None, padding=1).backward(dy)
    DY = nhwc(dy.to(dev), tdt)
    dx = nhwc(prev.to(dev), tdt).contiguous()
    zero = torch.zeros(cin, device=dev)
    keep = []
    for ph, k in enumerate(upsample_phase_weights(w)):
        wp = pack(dgrad_weight(k), tdt).to(dev)
        keep.append(wp)
        a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=w_, Hout=h, Wout=w_, C0=cout, C1=0, Cout=cin, Cout_pad=cin, ksize=2, stride=1, pad=0,
                       upsample=0, silu=0, out_mode=0, heads=0, x0=DY.data_ptr(), x1=None, scale=None, shift=None, w_packed=wp.data_ptr(),
                       bias=zero.data_ptr(), temb=None, temb_stride=0, residual=dx.data_ptr(), y=dx.data_ptr(), stats_out=None, im2col3=0,
                       phase=1 + ph, phase_in=1)
        L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
    torch.cuda.synchronize()
    got = dx.float().permute(0, 3, 1, 2).cpu()
    # (16-bit engines: the accumulation through `residual` rounds the running sum after every phase)
    assert rel(got, prev + x.grad) < {"f32": 2e-6, "bf16": 8e-3, "fp16": 1e-3}[mode]
    # refused: statistics with an input-side phase, phase_in without a phase
    st = torch.empty(B, 4 * lib.pd_conv_stat_tiles(h, w_, 2, 1), cin, 2, device=dev)
    a.stats_out = st.data_ptr()
    assert lib.pd_conv(C.byref(a), stream()) != 0
    a.stats_out, a.phase = None, 0
    assert lib.pd_conv(C.byref(a), stream()) != 0


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("B,tail", [(5, False), (2, True), (4, False), (3, True)])
def test_conv3x3_two_8x8_images_per_tile(env, mode, B, tail, monkeypatch):
    """Round 6 (opt-in, PD_CONV_STACK=1): at the 8 x 8 level a workgroup's tile is TWO samples (conv_kernel STACK: images stacked with their own zero halo rows).  Everything
    that is per sample -- time-embedding row, residual, output statistics, the odd batch's last tile -- against torch, and BIT FOR BIT against the
    64-pixel tile form (PD_CONV_STACK=0: the same K order per pixel, the same pixel order per statistic)."""
    L, lib, pack, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(61)
    c0, c1, cout, h = 64, 32, 96, 8
    x0, x1 = torch.randn(B, c0, h, h, generator=g), torch.randn(B, c1, h, h, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    temb = torch.randn(B, 200, generator=g)            # row stride 200, the layer's slice starts at column 0
    res = torch.randn(B, cout, h, h, generator=g)
    xa = torch.randn(B, 64, h, h, generator=g)
    ws = torch.randn(cout, 64, 1, 1, generator=g) / 8
    wp = pack(w, tdt)
    if tail:
        ct = wp.shape[0]
        wp = torch.cat([wp.reshape(ct, -1, 64, 8), pack(ws, tdt).reshape(ct, -1, 64, 8)], 1).contiguous()
    wp = wp.to(dev)
    X0, X1, XA, R = nhwc(x0.to(dev), tdt), nhwc(x1.to(dev), tdt), nhwc(xa.to(dev), tdt), nhwc(res.to(dev), tdt)
    bias, tb = b.to(dev), temb.to(dev)
    T = lib.pd_conv_stat_tiles(h, h, 3, 1)
    outs = []
    for stack in ("1", "0"):
        monkeypatch.setenv("PD_CONV_STACK", stack)
        y = torch.full((B, h, h, cout), float("nan"), dtype=tdt, device=dev)
        st = torch.full((B, T, cout, 2), float("nan"), device=dev)
        a = L.ConvArgs(dtype=code, B=B, Hin=h, Win=h, Hout=h, Wout=h, C0=c0, C1=c1, Cout=cout, Cout_pad=cout, ksize=3, stride=1, pad=1, upsample=0, silu=0,
                       out_mode=0, heads=0, x0=X0.data_ptr(), x1=X1.data_ptr(), scale=None, shift=None, w_packed=wp.data_ptr(), bias=bias.data_ptr(),
                       temb=tb.data_ptr(), temb_stride=200, residual=R.data_ptr(), y=y.data_ptr(), stats_out=st.data_ptr(),
                       tail_x0=XA.data_ptr() if tail else None, tail_x1=None, tail_C0=64 if tail else 0, tail_C1=0, im2col3=0)
        L.check(lib.pd_conv(C.byref(a), stream()), "pd_conv")
        torch.cuda.synchronize()
        outs.append((y, st))
    (y, st), (y0, st0) = outs
    assert torch.equal(y, y0) and torch.equal(st, st0)
    xin = torch.cat([bf16_round(x0, mode), bf16_round(x1, mode)], 1)
    ref = F.conv2d(xin, bf16_round(w, mode), b, padding=1) + temb[:, :cout, None, None]
    if tail:
        ref = ref + F.conv2d(bf16_round(xa, mode), bf16_round(ws, mode))
    ref = bf16_round(ref, mode) + bf16_round(res, mode) if mode != "f32" else ref + res
    assert rel(y.float().permute(0, 3, 1, 2), ref) < TOL[mode] * 1.5
    yf = y.float().cpu()
    assert rel(st[:, 0, :, 0], yf.sum((1, 2))) < 1e-5 and rel(st[:, 0, :, 1], (yf * yf).sum((1, 2))) < 1e-5


def test_upsample_phase_weights_on_device_match_the_host_contraction(env):
    """Round 6: pd_upsample_phase_weights (what the fine-tuning step's re-pack calls after every optimizer step) against the host-side
    contraction R_a w R_b^T of packing.upsample_phase_weights_stacked on the CPU, and against the convolution identity it stands for."""
    from phendiff_amd.packing import upsample_phase_weights_stacked
    L, lib, pack, dev = env
    g = torch.Generator().manual_seed(62)
    w = torch.randn(96, 40, 3, 3, generator=g)
    host = upsample_phase_weights_stacked(w)                       # CPU: einsum
    got = upsample_phase_weights_stacked(w.to(dev))                # device: the HIP entry point
    assert got.shape == host.shape == (4, 96, 40, 2, 2)
    assert rel(got, host) < 1e-7
    into = torch.full_like(got, float("nan"))
    assert upsample_phase_weights_stacked(w.to(dev), out=into) is into and torch.equal(into, got)
    x = torch.randn(2, 40, 5, 7, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    k = got.cpu()
    for a in range(2):
        for b in range(2):
            xp = F.pad(x, (1 - b, b, 1 - a, a))
            ph = F.conv2d(xp, k[2 * a + b])
            assert rel(ph, ref[:, :, a::2, b::2]) < 1e-5
