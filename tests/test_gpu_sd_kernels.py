"""SD-tier kernels on MI355X against plain PyTorch fp32 of the same ops (what the oracle's Transformer2D blocks dispatch to)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import DT, bf16_round, env, rel, stream  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(2, 2, 256, 256), (1, 5, 1024, 1024), (2, 3, 200, 77), (1, 1, 16, 16), (2, 2, 130, 4), (1, 20, 64, 64),
                                 (16, 16, 1024, 1024), (16, 16, 1100, 1000)])     # the last two: 64 queries per wave in bf16
def test_attention_d64(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, Nq, Nkv = cfg
    Cc = heads * 64
    g = torch.Generator().manual_seed(41)
    self_attn = Nq == Nkv
    if self_attn:       # q, k, v are slices of one fused projection output [B][N][3C]
        qkv = bf16_round(torch.randn(B, Nq, 3 * Cc, generator=g), mode)
        q, k, v = qkv[..., :Cc], qkv[..., Cc:2 * Cc], qkv[..., 2 * Cc:]
        QKV = qkv.to(tdt).to(dev).contiguous()
        esz = QKV.element_size()
        qptr, kptr, vptr, qs, kvs = QKV.data_ptr(), QKV.data_ptr() + Cc * esz, QKV.data_ptr() + 2 * Cc * esz, 3 * Cc, 3 * Cc
    else:               # cross attention: q [B][Nq][C], kv = fused [B][Nkv][2C]
        q = bf16_round(torch.randn(B, Nq, Cc, generator=g), mode)
        kv = bf16_round(torch.randn(B, Nkv, 2 * Cc, generator=g), mode)
        k, v = kv[..., :Cc], kv[..., Cc:]
        Q, KV = q.to(tdt).to(dev).contiguous(), kv.to(tdt).to(dev).contiguous()
        qptr, kptr, vptr, qs, kvs = Q.data_ptr(), KV.data_ptr(), KV.data_ptr() + Cc * KV.element_size(), Cc, 2 * Cc
    out = torch.full((B, Nq, Cc), float("nan"), dtype=tdt, device=dev)
    a = L.AttnD64Args(dtype=code, B=B, heads=heads, Nq=Nq, Nkv=Nkv, q=qptr, q_stride=qs, k=kptr, v=vptr, kv_stride=kvs,
                      out=out.data_ptr(), out_stride=Cc)
    L.check(lib.pd_attn_d64(C.byref(a), stream()), "pd_attn_d64")
    torch.cuda.synchronize()
    sp = lambda t, n: t.reshape(B, n, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, Nq), sp(k, Nkv), sp(v, Nkv)).transpose(1, 2).reshape(B, Nq, Cc)
    assert rel(out.float(), ref) < {"f32": 5e-6, "bf16": 8e-3, "fp16": 1e-3}[mode]       # measured 7.6e-7 / 2.9e-3 / 3.7e-4


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(37, 64), (1000, 320), (300, 640), (513, 1280), (4, 2048)])
def test_layernorm(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    rows, Cc = cfg
    g = torch.Generator().manual_seed(42)
    x = bf16_round(torch.randn(rows, Cc, generator=g) * 2 + 0.5, mode)
    gamma, beta = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    X, y = x.to(tdt).to(dev), torch.empty((rows, Cc), dtype=tdt, device=dev)
    gm, bt = gamma.to(dev), beta.to(dev)
    a = L.LayerNormArgs(dtype=code, rows=rows, C=Cc, eps=1e-5, x=X.data_ptr(), gamma=gm.data_ptr(), beta=bt.data_ptr(), y=y.data_ptr())
    L.check(lib.pd_layernorm(C.byref(a), stream()), "pd_layernorm")
    torch.cuda.synchronize()
    assert rel(y.float(), F.layer_norm(x, (Cc,), gamma, beta, 1e-5)) < {"f32": 2e-6, "bf16": 4e-3, "fp16": 5e-4}[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_geglu(env, mode):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(43)
    x = bf16_round(torch.randn(301, 2 * 256, generator=g) * 2, mode)
    X, y = x.to(tdt).to(dev), torch.empty((301, 256), dtype=tdt, device=dev)
    a = L.GegluArgs(dtype=code, rows=301, inner=256, x=X.data_ptr(), y=y.data_ptr())
    L.check(lib.pd_geglu(C.byref(a), stream()), "pd_geglu")
    torch.cuda.synchronize()
    h, gate = x.chunk(2, -1)
    assert rel(y.float(), h * F.gelu(gate)) < {"f32": 2e-6, "bf16": 4e-3, "fp16": 5e-4}[mode]


# ---- backward kernels of the Transformer2D blocks: against autograd over plain PyTorch fp32 ---------------------------------
@pytest.mark.parametrize("mode,shape", [("f32", (2, 3, 320)), ("bf16", (2, 3, 320)), ("bf16", (16, 16, 1024)), ("fp16", (16, 16, 1024))])   # the last two: 64 queries per wave
def test_attention_d64_deferred_rescale(env, mode, shape):
    """Scores that jump far above the running reference maximum late in the key sequence (and a first tile far BELOW the
    rest) force the rescale branch of the deferred-rescale online softmax; checked against fp64 softmax, lse included."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, N = shape
    Cc = heads * 64
    g = torch.Generator().manual_seed(43)
    q = torch.randn(B, N, Cc, generator=g)
    k = torch.randn(B, N, Cc, generator=g)
    v = torch.randn(B, N, Cc, generator=g)
    sp = lambda t: t.reshape(B, N, heads, 64).transpose(1, 2)
    qh, kh = sp(q), sp(k)                                   # views: writes land in q / k
    kh[:, :, 200] = qh[:, :, 5] * 3.0                       # query 5: score ~ 3 |q|^2 / 8 ~ 24 (x log2 e = 35) at key 200
    kh[:, :, 290] = qh[:, :, 170] * 6.0
    kh[:, :, 100] = qh[:, :, 319] * 2.5
    kh[:, :, :32] -= 4.0 * qh[:, :, 40:41] / qh[:, :, 40:41].norm(dim=-1, keepdim=True)   # query 40: first sub-tile ~ -32 below the rest
    if N >= 1024:
        # round 6 (the two-fragment form takes the row maximum lazily, from the lane sums of the probabilities): a score ~ 160 nats (230 in the
        # log2 domain) above everything before it -- exp2 overflows to +inf on the lazy path, which must fall back to the exact one --, and one
        # that only just lifts a lane sum over the 2^15 trigger
        kh[:, :, 700] = qh[:, :, 900] * 20.0
        kh[:, :, 901] = qh[:, :, 333] * 1.6
    q, k, v = (bf16_round(t, mode) for t in (q, k, v))
    Q, K, V = (t.to(tdt).to(dev).contiguous() for t in (q, k, v))
    out = torch.full((B, N, Cc), float("nan"), dtype=tdt, device=dev)
    lse = torch.full((B, heads, N), float("nan"), dtype=torch.float32, device=dev)
    a = L.AttnD64Args(dtype=code, B=B, heads=heads, Nq=N, Nkv=N, q=Q.data_ptr(), q_stride=Cc, k=K.data_ptr(), v=V.data_ptr(),
                      kv_stride=Cc, out=out.data_ptr(), out_stride=Cc, lse=lse.data_ptr())
    L.check(lib.pd_attn_d64(C.byref(a), stream()), "pd_attn_d64")
    torch.cuda.synchronize()
    s = torch.einsum("bhid,bhjd->bhij", sp(q).double(), sp(k).double()) / 8
    assert float((s.max(-1).values - s[..., :32].max(-1).values).max()) > 20         # the spikes are real
    ref = (torch.softmax(s, -1) @ sp(v).double()).transpose(1, 2).reshape(B, N, Cc)
    assert torch.isfinite(out).all()
    assert rel(out.float(), ref) < (1e-5 if mode == "f32" else 1.5e-2)
    assert rel(lse.cpu(), torch.logsumexp(s, -1) * 1.4426950408889634) < (1e-5 if mode == "f32" else 2e-3)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 2, 256, 256), (1, 5, 1024, 1024), (2, 3, 200, 77), (1, 1, 16, 16), (2, 2, 130, 4), (1, 2, 70, 200)])
def test_attention_d64_backward(env, mode, cfg):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, heads, Nq, Nkv = cfg
    Cc = heads * 64
    g = torch.Generator().manual_seed(51)
    q = bf16_round(torch.randn(B, Nq, Cc, generator=g), mode).requires_grad_()
    kv = bf16_round(torch.randn(B, Nkv, 2 * Cc, generator=g), mode).requires_grad_()
    do = bf16_round(torch.randn(B, Nq, Cc, generator=g), mode)
    sp = lambda t, n: t.reshape(B, n, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, Nq), sp(kv[..., :Cc], Nkv), sp(kv[..., Cc:], Nkv)).transpose(1, 2).reshape(B, Nq, Cc)
    ref.backward(do)
    Q, KV, DO = q.detach().to(tdt).to(dev), kv.detach().to(tdt).to(dev), do.to(tdt).to(dev)
    esz = Q.element_size()
    out = torch.empty((B, Nq, Cc), dtype=tdt, device=dev)
    lse = torch.empty((B, heads, Nq), dtype=torch.float32, device=dev)
    a = L.AttnD64Args(dtype=code, B=B, heads=heads, Nq=Nq, Nkv=Nkv, q=Q.data_ptr(), q_stride=Cc, k=KV.data_ptr(),
                      v=KV.data_ptr() + Cc * esz, kv_stride=2 * Cc, out=out.data_ptr(), out_stride=Cc, lse=lse.data_ptr())
    L.check(lib.pd_attn_d64(C.byref(a), stream()), "pd_attn_d64")
    # lse: log2-domain log-sum-exp of the scaled scores
    s = torch.einsum("bhid,bhjd->bhij", sp(q.detach(), Nq), sp(kv.detach()[..., :Cc], Nkv)) / 8
    assert rel(lse.cpu(), torch.logsumexp(s, -1) * 1.4426950408889634) < (1e-5 if mode == "f32" else 2e-3)
    dq = torch.full((B, Nq, Cc), float("nan"), dtype=tdt, device=dev)
    dkv = torch.full((B, Nkv, 2 * Cc), float("nan"), dtype=tdt, device=dev)
    delta = torch.empty((B, heads, Nq), dtype=torch.float32, device=dev)
    b = L.AttnD64BwdArgs(dtype=code, B=B, heads=heads, Nq=Nq, Nkv=Nkv, q=Q.data_ptr(), q_stride=Cc, k=KV.data_ptr(),
                         v=KV.data_ptr() + Cc * esz, kv_stride=2 * Cc, o=out.data_ptr(), dout=DO.data_ptr(), o_stride=Cc,
                         lse=lse.data_ptr(), delta=delta.data_ptr(), dq=dq.data_ptr(), dq_stride=Cc, dk=dkv.data_ptr(),
                         dv=dkv.data_ptr() + Cc * esz, dkv_stride=2 * Cc)
    L.check(lib.pd_attn_d64_bwd(C.byref(b), stream()), "pd_attn_d64_bwd")
    torch.cuda.synchronize()
    tol = 3e-5 if mode == "f32" else 2.5e-2
    assert rel(dq.float(), q.grad) < tol, rel(dq.float(), q.grad)
    assert rel(dkv.float()[..., :Cc], kv.grad[..., :Cc]) < tol
    assert rel(dkv.float()[..., Cc:], kv.grad[..., Cc:]) < tol


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [(37, 64), (5000, 320), (300, 640), (513, 1280), (4, 2048)])     # 1 / 1 / 2 / 3 / 4 pieces per lane
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_backward(env, mode, cfg, with_res):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    rows, Cc = cfg
    g = torch.Generator().manual_seed(52)
    x = bf16_round(torch.randn(rows, Cc, generator=g) * 2 + 0.5, mode).requires_grad_()
    gamma = torch.randn(Cc, generator=g).requires_grad_()
    beta = torch.randn(Cc, generator=g).requires_grad_()
    dy = bf16_round(torch.randn(rows, Cc, generator=g), mode)
    res = bf16_round(torch.randn(rows, Cc, generator=g), mode) if with_res else None
    F.layer_norm(x, (Cc,), gamma, beta, 1e-5).backward(dy)
    X, DY = x.detach().to(tdt).to(dev), dy.to(tdt).to(dev)
    R = res.to(tdt).to(dev) if with_res else None
    dx = torch.empty((rows, Cc), dtype=tdt, device=dev)
    dgm = torch.full((Cc,), 1.0, dtype=torch.float32, device=dev)        # accumulates (+=)
    dbt = torch.full((Cc,), -2.0, dtype=torch.float32, device=dev)
    nb = lib.pd_layernorm_bwd_blocks(rows)
    part = torch.empty(nb * 2 * Cc, dtype=torch.float32, device=dev)
    gm = gamma.detach().to(dev)
    a = L.LayerNormBwdArgs(dtype=code, rows=rows, C=Cc, eps=1e-5, x=X.data_ptr(), dy=DY.data_ptr(), gamma=gm.data_ptr(),
                           res=L.ptr(R), dx=dx.data_ptr(), dgamma=dgm.data_ptr(), dbeta=dbt.data_ptr(), partial=part.data_ptr())
    L.check(lib.pd_layernorm_bwd(C.byref(a), stream()), "pd_layernorm_bwd")
    torch.cuda.synchronize()
    want = x.grad + (res if with_res else 0)
    assert rel(dx.float(), want) < (3e-6 if mode == "f32" else 5e-3)
    assert rel(dgm - 1.0, gamma.grad) < 2e-5 and rel(dbt + 2.0, beta.grad) < 2e-5
    # round 6: + the column sums of the stored dx (the bias gradient of the Linear layer dx is the output gradient of), C <= 1536
    if Cc <= 1536:
        part3 = torch.empty(nb * 3 * Cc, dtype=torch.float32, device=dev)
        dxs = torch.full((Cc,), 0.5, dtype=torch.float32, device=dev)       # accumulates (+=)
        dgm3, dbt3, dx3 = torch.zeros_like(dgm), torch.zeros_like(dbt), torch.empty_like(dx)
        a.dgamma, a.dbeta, a.partial, a.dxsum, a.dx = dgm3.data_ptr(), dbt3.data_ptr(), part3.data_ptr(), dxs.data_ptr(), dx3.data_ptr()
        L.check(lib.pd_layernorm_bwd(C.byref(a), stream()), "pd_layernorm_bwd")
        torch.cuda.synchronize()
        # (another instantiation of the kernel: the compiler may contract the fp32 expressions differently -- last-bit differences in the f32 engine)
        assert rel(dx3.float(), dx.float()) < 1e-6 and rel(dgm3, dgm - 1.0) < 1e-5 and rel(dbt3, dbt + 2.0) < 1e-5
        assert rel(dxs - 0.5, dx3.double().sum(0).float()) < 1e-5
        a.dxsum = None
    else:
        a.dxsum = dx.data_ptr()
        assert lib.pd_layernorm_bwd(C.byref(a), stream()) != 0              # refused, not silently skipped
        a.dxsum = None
    # input-gradient-only form
    a.dgamma, a.dbeta, a.partial = None, None, None
    dx2 = torch.empty_like(dx)
    a.dx = dx2.data_ptr()
    L.check(lib.pd_layernorm_bwd(C.byref(a), stream()), "pd_layernorm_bwd")
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_geglu_backward(env, mode):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    rows, inner = 333, 1280
    g = torch.Generator().manual_seed(53)
    x = bf16_round(torch.randn(rows, 2 * inner, generator=g) * 1.5, mode).requires_grad_()
    dy = bf16_round(torch.randn(rows, inner, generator=g), mode)
    h, gate = x.chunk(2, dim=-1)
    (h * F.gelu(gate)).backward(dy)
    X, DY = x.detach().to(tdt).to(dev), dy.to(tdt).to(dev)
    dx = torch.empty((rows, 2 * inner), dtype=tdt, device=dev)
    a = L.GegluBwdArgs(dtype=code, rows=rows, inner=inner, x=X.data_ptr(), dy=DY.data_ptr(), dx=dx.data_ptr())
    L.check(lib.pd_geglu_bwd(C.byref(a), stream()), "pd_geglu_bwd")
    torch.cuda.synchronize()
    assert rel(dx.float(), x.grad) < (2e-6 if mode == "f32" else 4e-3)
    # round 6: the same launch leaves the per-split column sums of the stored dx (workspace layout of pd_channel_sum with x = NULL)
    for B, splits in ((3, 5), (1, 64), (37, 1)):          # rows = 333 = 3 x 111 = 37 x 9; 111 rows in 5 splits of 23: a ragged last split
        ws = torch.full((B * splits * 2 * inner,), float("nan"), dtype=torch.float32, device=dev)
        dx2 = torch.empty_like(dx)
        a2 = L.GegluBwdArgs(dtype=code, rows=rows, inner=inner, x=X.data_ptr(), dy=DY.data_ptr(), dx=dx2.data_ptr(), sums=ws.data_ptr(), sum_splits=splits, B=B)
        L.check(lib.pd_geglu_bwd(C.byref(a2), stream()), "pd_geglu_bwd")
        torch.cuda.synchronize()
        assert rel(dx2.float(), dx.float()) < 1e-6 and (mode == "f32" or torch.equal(dx2, dx))     # (f32: another kernel, other fp32 contraction)
        per = ws.reshape(B, splits, 2 * inner).sum(1)
        assert rel(per, dx2.double().reshape(B, rows // B, 2 * inner).sum(1).float()) < 1e-5
    a2.inner, a2.rows = 1288, 10
    assert lib.pd_geglu_bwd(C.byref(a2), stream()) != 0      # inner % 256 != 0 with sums: refused


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(1024, 320, 960, 0, 0), (300, 64, 64, 1, 0), (2048, 1280, 10240, 0, 0), (77 * 3, 96, 256, 0, 0), (515, 5120, 1280, 1, 0),
                                 (4096, 96, 320, 1, 32), (130, 32, 8, 0, 0), (65536 + 70, 128, 512, 1, 0),
                                 # round 3: shapes the 16-bit engines run on the DMA-staged 256-token kernel (linear_dma_kernel) -- NC = 2
                                 # with a ragged 128-channel tile and ragged tokens, one K chunk, a strided input, NC = 4 at K = 1280
                                 (32768 + 5, 320, 320, 1, 0), (70000, 64, 1280, 0, 0), (33000, 192, 384, 1, 64), (65536, 1280, 1280, 1, 0),
                                 # round 4: N = 320 with >= 256 token tiles -> the 320-channel tile (NC = 5, two-pass epilogue): ragged tokens + residual,
                                 # one-chunk-deep K with a strided input, no residual
                                 (65536 + 37, 1280, 320, 1, 0), (70000, 320, 320, 1, 64), (65536, 64, 320, 0, 0)])
def test_linear_gemm(env, mode, cfg):
    """pd_linear against F.linear: full / ragged token tiles, K with a trailing half chunk (96, 32), N not a multiple of the
    128-channel tile, residual, strided input rows (a slice of a fused projection's output).  The (2048, 1280, 10240) and the
    last shape are launched with 256-channel tiles (NC = 4) in the 16-bit engines (N % 256 == 0 and >= 512 workgroups)."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    M, K, N, with_res, xpad = cfg
    g = torch.Generator().manual_seed(61)
    xs = K + xpad
    xfull = bf16_round(torch.randn(M, xs, generator=g), mode)
    w = bf16_round(torch.randn(N, K, generator=g) / K ** 0.5, mode)
    bias = torch.randn(N, generator=g)
    res = bf16_round(torch.randn(M, N, generator=g), mode) if with_res else None
    npad = ((N + 31) // 32) * 32
    wp = pack_conv_weight(w[:, :, None, None], tdt, npad).to(dev)
    bp = torch.zeros(npad)
    bp[:N] = bias
    X, Bv = xfull.to(tdt).to(dev), bp.to(dev)
    R = res.to(tdt).to(dev) if with_res else None
    y = torch.full((M, N), float("nan"), dtype=tdt, device=dev)
    a = L.LinearArgs(dtype=code, M=M, K=K, N=N, N_pad=npad, x=X.data_ptr(), x_stride=xs, w_packed=wp.data_ptr(), bias=Bv.data_ptr(),
                     residual=L.ptr(R), y=y.data_ptr())
    L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
    torch.cuda.synchronize()
    ref = F.linear(xfull[:, :K], w, bias) + (res if with_res else 0)
    assert rel(y.float(), ref) < {"f32": 2e-6, "bf16": 4e-3, "fp16": 5e-4}[mode]


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(1024, 320, 1280), (300, 64, 256), (515, 1280, 5120), (130, 96, 32),
                                 (70000, 320, 1280), (40000, 64, 96)])       # the last two: linear_dma_kernel (NC = 4 / NC = 2) in the 16-bit engines
def test_linear_gemm_fused_geglu(env, mode, cfg):
    """pd_linear(glu = 1) = diffusers GEGLU: proj -> chunk(2) -> value * F.gelu(gate), with the value / gate weight rows
    interleaved per 32-row tile (two pd_pack_weight calls with a two-tile stride, as the re-pack after an optimizer step
    does) and the bias in module order.  M ragged, inner widths that are / are not multiples of the 64-channel output tile."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    M, K, inner = cfg
    g = torch.Generator().manual_seed(64)
    x = bf16_round(torch.randn(M, K, generator=g), mode)
    w = bf16_round(torch.randn(2 * inner, K, generator=g) / K ** 0.5, mode)
    bias = torch.randn(2 * inner, generator=g)
    X, W, Bv = x.to(tdt).to(dev), w.to(dev), bias.to(dev)
    tile = (K // 32) * 2 * 512
    wp = torch.full((2 * inner // 32, tile), float("nan"), dtype=tdt, device=dev)
    for half in (0, 1):
        src = W[half * inner:(half + 1) * inner].contiguous()
        a = L.PackWeightArgs(dtype=code, cout=inner, cin=K, cout_pad=inner, cin_pad=K, ksize=1, src_in=K, dgrad=0,
                             src=src.data_ptr(), dst=wp.data_ptr() + half * tile * wp.element_size(), dst_ct_stride=2 * tile)
        L.check(lib.pd_pack_weight(C.byref(a), stream()), "pd_pack_weight")
    y = torch.full((M, inner), float("nan"), dtype=tdt, device=dev)
    a = L.LinearArgs(dtype=code, M=M, K=K, N=2 * inner, N_pad=2 * inner, x=X.data_ptr(), x_stride=K, w_packed=wp.data_ptr(),
                     bias=Bv.data_ptr(), residual=None, y=y.data_ptr(), glu=1)
    L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
    torch.cuda.synchronize()
    assert not torch.isnan(wp.float()).any()
    proj = F.linear(x.double(), w.double(), bias.double())
    ref = proj[:, :inner] * F.gelu(proj[:, inner:])
    assert rel(y.float(), ref) < {"f32": 2e-6, "bf16": 4e-3, "fp16": 5e-4}[mode]
    a.N = a.N_pad = 2 * inner + 32                                             # halves that are not whole tiles: refused
    assert lib.pd_linear(C.byref(a), stream()) == -2


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("cfg", [(4096, 320, 960, 0), (300, 64, 64, 0), (2048, 1280, 2560, 0), (77 * 3, 96, 256, 0), (8200, 640, 200, 24), (64, 8, 8, 0),
                                 (2048, 256, 128, 8), (4096, 320, 320, 0), (65 * 32, 640, 320, 16), (2048 + 32, 96, 352, 0)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_token_wgrad(env, mode, cfg, accumulate):
    """pd_token_wgrad against dY^T X: ragged token chunks, channel counts that are not multiples of the 128 tile, strided rows,
    a deliberately small slab (fewer splits), accumulation into an existing gradient.  The bf16 cases with M % 32 == 0, M >= 2048 and
    32-multiple channel counts run the DMA-staged forms (320 x 128: (2048, 1280, 2560), (4096, 320, 320); 128 x 320: (4096, 320, 960),
    (2080, 640, 320); 128 x 128: (2048, 256, 128); planes past N / K clamped: (2080, 96, 352)), the others token_wgrad_kernel."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    M, K, N, pad = cfg
    g = torch.Generator().manual_seed(62)
    x = bf16_round(torch.randn(M, K + pad, generator=g), mode)
    dy = bf16_round(torch.randn(M, N + pad, generator=g), mode)
    X, DY = x.to(tdt).to(dev), dy.to(tdt).to(dev)
    prev = torch.randn(N, K, generator=g)
    dw = prev.clone().to(dev) if accumulate else torch.full((N, K), float("nan"), device=dev)
    a = L.TokenWgradArgs(dtype=code, M=M, K=K, N=N, x=X.data_ptr(), x_stride=K + pad, dy=DY.data_ptr(), dy_stride=N + pad,
                         dw=dw.data_ptr(), accumulate=accumulate)
    need = lib.pd_token_wgrad_workspace(C.byref(a))
    if M == 8200:
        need = max(need // 3, ((N + 127) // 128) * 128 * ((K + 127) // 128) * 128 * 4)       # fewer splits than planned
    slab = torch.empty(need // 4, dtype=torch.float32, device=dev)
    a.slab, a.slab_bytes = slab.data_ptr(), need
    L.check(lib.pd_token_wgrad(C.byref(a), stream()), "pd_token_wgrad")
    torch.cuda.synchronize()
    ref = dy[:, :N].double().t() @ x[:, :K].double()
    got = dw.cpu().double() - (prev.double() if accumulate else 0)
    assert rel(got.float(), ref.float()) < (2e-5 if mode == "f32" else 2e-3)
    # bitwise reproducible
    dw2 = prev.clone().to(dev) if accumulate else torch.full((N, K), float("nan"), device=dev)
    a.dw = dw2.data_ptr()
    L.check(lib.pd_token_wgrad(C.byref(a), stream()), "pd_token_wgrad")
    assert torch.equal(dw2, dw)
    # round 6 (ABI 8): the two launches separately -- stage 1 (GEMM -> slab), then stage 2 (fold) on ANOTHER stream behind an event: the same bits
    dw3 = prev.clone().to(dev) if accumulate else torch.full((N, K), float("nan"), device=dev)
    slab.fill_(float("nan"))
    a.dw, a.stage = dw3.data_ptr(), 1
    L.check(lib.pd_token_wgrad(C.byref(a), stream()), "pd_token_wgrad")
    ev, side = torch.cuda.Event(), torch.cuda.Stream()
    ev.record(torch.cuda.current_stream())
    side.wait_event(ev)
    a.stage = 2
    L.check(lib.pd_token_wgrad(C.byref(a), side.cuda_stream), "pd_token_wgrad")
    side.synchronize()
    assert torch.equal(dw3, dw)
    a.stage = 3
    assert lib.pd_token_wgrad(C.byref(a), stream()) != 0


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
def test_linear_gemm_groupnorm_prologue_and_head_major_output(env, mode):
    """pd_linear as the fused q/k/v projection of the pixel-UNet attention: x*scale[n] + shift[n] applied while staging, output
    written head-major [3][B][heads][tokens][8] (what pd_attn_d8 reads)."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, Ntok, Cc, heads = 3, 256, 128, 16
    g = torch.Generator().manual_seed(63)
    x = bf16_round(torch.randn(B, Ntok, Cc, generator=g), mode)
    scale, shift = torch.rand(B, Cc, generator=g) + 0.5, torch.randn(B, Cc, generator=g)
    w = bf16_round(torch.randn(3 * Cc, Cc, generator=g) / Cc ** 0.5, mode)
    bias = torch.randn(3 * Cc, generator=g)
    z = bf16_round(x * scale[:, None, :] + shift[:, None, :], mode)          # the staged operand is rounded to the compute dtype
    ref = torch.nn.functional.linear(z, w, bias)                             # [B][N][3C]
    ref = ref.reshape(B, Ntok, 3, heads, 8).permute(2, 0, 3, 1, 4).contiguous()
    wp = pack_conv_weight(w[:, :, None, None], tdt).to(dev)
    X, sc, sh, bv = x.to(tdt).to(dev), scale.to(dev), shift.to(dev), bias.to(dev)
    y = torch.full((3, B, heads, Ntok, 8), float("nan"), dtype=tdt, device=dev)
    a = L.LinearArgs(dtype=code, M=B * Ntok, K=Cc, N=3 * Cc, N_pad=3 * Cc, x=X.data_ptr(), x_stride=Cc, w_packed=wp.data_ptr(),
                     bias=bv.data_ptr(), residual=None, y=y.data_ptr(), scale=sc.data_ptr(), shift=sh.data_ptr(),
                     rows_per_sample=Ntok, qkv_heads=heads)
    L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
    torch.cuda.synchronize()
    assert rel(y.float(), ref) < (3e-6 if mode == "f32" else 4e-3)
    kmax2 = torch.zeros(B, heads, device=dev)
    a.kmax2_out = kmax2.data_ptr()
    if mode == "f32":
        assert lib.pd_linear(C.byref(a), stream()) == -2                      # the key bound is a 16-bit-engine feature
    else:
        # max |k|^2 per (sample, head) of the key rows AS STORED (pd_attn_d8's kmax2), exact up to fp32 summation order
        L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
        torch.cuda.synchronize()
        want = (y[1].float() ** 2).sum(-1).amax(-1)
        assert torch.allclose(kmax2, want, rtol=1e-5, atol=0)
    a.kmax2_out = None
    a.rows_per_sample = 200                                                   # not a multiple of 128: refused, not mis-addressed
    assert lib.pd_linear(C.byref(a), stream()) == -2


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(3, 256, 128, 16), (2, 1024, 256, 32), (2, 512, 64, 8)])
def test_linear_gemm_groupnorm_folded_into_per_sample_weights(env, mode, shape):
    """Round 4: the pixel-UNet attention's fused q/k/v projection (cond_unet_2d.py:176-178; diffusers Attention.group_norm -> to_q / to_k /
    to_v) through the DMA-staged GEMM: with a `fold_ws` workspace pd_linear folds the GroupNorm affine into per-sample weights
    W diag(scale_n) and biases b + W shift_n (one small launch) and multiplies the tokens as they are.  Same head-major output and
    key bound as the register-staged route; the arithmetic differs in where the 16-bit rounding sits (weights instead of the
    normalised activations), so both are compared with the un-rounded fp32 result."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, Ntok, Cc, heads = shape
    g = torch.Generator().manual_seed(64)
    x = bf16_round(torch.randn(B, Ntok, Cc, generator=g) * 1.5 + 0.3, mode)
    scale, shift = torch.rand(B, Cc, generator=g) + 0.5, torch.randn(B, Cc, generator=g)
    w = bf16_round(torch.randn(3 * Cc, Cc, generator=g) / Cc ** 0.5, mode)
    bias = torch.randn(3 * Cc, generator=g)
    ref = torch.nn.functional.linear(x * scale[:, None, :] + shift[:, None, :], w, bias)
    ref = ref.reshape(B, Ntok, 3, heads, 8).permute(2, 0, 3, 1, 4).contiguous()
    wp = pack_conv_weight(w[:, :, None, None], tdt).to(dev)
    X, sc, sh, bv = x.to(tdt).to(dev), scale.to(dev), shift.to(dev), bias.to(dev)
    a = L.LinearArgs(dtype=code, M=B * Ntok, K=Cc, N=3 * Cc, N_pad=3 * Cc, x=X.data_ptr(), x_stride=Cc, w_packed=wp.data_ptr(),
                     bias=bv.data_ptr(), residual=None, y=None, scale=sc.data_ptr(), shift=sh.data_ptr(),
                     rows_per_sample=Ntok, qkv_heads=heads)
    need = int(lib.pd_linear_fold_workspace(C.byref(a)))
    assert need == B * (3 * Cc * Cc * 2 + 3 * Cc * 4)
    outs = {}
    for route in ("staged", "folded"):
        y = torch.full((3, B, heads, Ntok, 8), float("nan"), dtype=tdt, device=dev)
        kmax2 = torch.zeros(B, heads, device=dev)
        ws = torch.empty(need + 64, dtype=torch.uint8, device=dev)
        a.y, a.kmax2_out = y.data_ptr(), kmax2.data_ptr()
        a.fold_ws, a.fold_ws_bytes = (ws.data_ptr(), need) if route == "folded" else (None, 0)
        L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
        torch.cuda.synchronize()
        assert rel(y.float(), ref) < (5e-3 if mode == "bf16" else 6e-4), route      # measured 2.2e-3 / 2.8e-4
        want = (y[1].float() ** 2).sum(-1).amax(-1)                           # the bound is over the key rows AS STORED
        assert torch.allclose(kmax2, want, rtol=1e-5, atol=0), route
        outs[route] = y.float()
    assert rel(outs["folded"], outs["staged"]) < (6e-3 if mode == "bf16" else 8e-4)      # measured 2.8e-3 / 3.5e-4
    # a workspace that is too small is not used (the staged route answers), a tile that would straddle samples has no folded route
    a.fold_ws_bytes = need - 1
    L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
    a.rows_per_sample = 128
    a.M = B * Ntok
    assert int(lib.pd_linear_fold_workspace(C.byref(a))) == 0


# ---- round 6: the 256 x 256 eight-phase GEMM (csrc/linear_p8.hip) -------------------------------------------------------------------
P8_PLAIN = [(256, 64, 256, 0, 0),          # one K tile: prologue + drain only
            (300, 128, 256, 1, 0),         # two K tiles, ragged tokens, residual
            (1000, 320, 320, 1, 0),        # odd number of K tiles; N_pad = 320: second channel tile mostly clamped weight tiles
            (515, 5120, 1280, 1, 0),       # 80 K tiles
            (2048 + 77, 640, 1920, 0, 64), # strided input rows, ragged last channel tile
            (4096, 1280, 2560, 1, 0),      # 16 x 10 tiles: the panel walk over a ragged channel panel (8 + 2)
            (1280, 192, 200, 0, 0),        # N a multiple of 8 only (N_pad = 224)
            (9 * 256, 256, 9 * 256, 1, 0)] # 9 x 9 tiles: ragged token panel (4 + 4 + 1) and channel panel (8 + 1), tile count % 8 != 0


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg", P8_PLAIN)
def test_linear_gemm_eight_phase(env, mode, cfg, monkeypatch):
    """pd_linear on the eight-phase 256 x 256 kernel (forced wherever it is eligible: PD_LIN_P8=1 is read at every dispatch) against
    F.linear in fp64; the same launch with the kernel switched off must agree with it to the rounding of the output type."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    M, K, N, with_res, xpad = cfg
    g = torch.Generator().manual_seed(71)
    xs = K + xpad
    xfull = bf16_round(torch.randn(M, xs, generator=g), mode)
    w = bf16_round(torch.randn(N, K, generator=g) / K ** 0.5, mode)
    bias = torch.randn(N, generator=g)
    res = bf16_round(torch.randn(M, N, generator=g), mode) if with_res else None
    npad = ((N + 31) // 32) * 32
    wp = pack_conv_weight(w[:, :, None, None], tdt, npad).to(dev)
    bp = torch.zeros(npad)
    bp[:N] = bias
    X, Bv = xfull.to(tdt).to(dev), bp.to(dev)
    R = res.to(tdt).to(dev) if with_res else None
    ref = F.linear(xfull[:, :K].double(), w.double(), bias.double()) + (res.double() if with_res else 0)
    outs = {}
    for p8 in ("1", "0"):
        monkeypatch.setenv("PD_LIN_P8", p8)
        y = torch.full((M + 3, N), float("nan"), dtype=tdt, device=dev)      # three guard rows: nothing may be written past M
        a = L.LinearArgs(dtype=code, M=M, K=K, N=N, N_pad=npad, x=X.data_ptr(), x_stride=xs, w_packed=wp.data_ptr(), bias=Bv.data_ptr(),
                         residual=L.ptr(R), y=y.data_ptr())
        L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
        torch.cuda.synchronize()
        assert torch.isnan(y[M:].float()).all()
        outs[p8] = y[:M].float().cpu()
        assert rel(outs[p8], ref.float()) < {"bf16": 4e-3, "fp16": 5e-4}[mode], p8
    assert rel(outs["1"], outs["0"]) < {"bf16": 3e-3, "fp16": 4e-4}[mode]


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("cfg", [(515, 1280, 5120), (1024, 320, 1280), (300, 64, 256), (2100, 640, 320), (4096, 128, 64)])
def test_linear_gemm_eight_phase_fused_geglu(env, mode, cfg, monkeypatch):
    """The fused GEGLU epilogue of the eight-phase kernel: even / odd packed weight tiles = (value, gate) pairs; inner widths that are
    / are not multiples of the 128-channel output tile."""
    L, lib, _, dev = env
    code, tdt = DT[mode]
    M, K, inner = cfg
    g = torch.Generator().manual_seed(72)
    x = bf16_round(torch.randn(M, K, generator=g), mode)
    w = bf16_round(torch.randn(2 * inner, K, generator=g) / K ** 0.5, mode)
    bias = torch.randn(2 * inner, generator=g)
    X, W, Bv = x.to(tdt).to(dev), w.to(dev), bias.to(dev)
    tile = (K // 32) * 2 * 512
    wp = torch.full((2 * inner // 32, tile), float("nan"), dtype=tdt, device=dev)
    for half in (0, 1):
        src = W[half * inner:(half + 1) * inner].contiguous()
        a = L.PackWeightArgs(dtype=code, cout=inner, cin=K, cout_pad=inner, cin_pad=K, ksize=1, src_in=K, dgrad=0,
                             src=src.data_ptr(), dst=wp.data_ptr() + half * tile * wp.element_size(), dst_ct_stride=2 * tile)
        L.check(lib.pd_pack_weight(C.byref(a), stream()), "pd_pack_weight")
    proj = F.linear(x.double(), w.double(), bias.double())
    ref = (proj[:, :inner] * F.gelu(proj[:, inner:])).float()
    monkeypatch.setenv("PD_LIN_P8", "1")
    y = torch.full((M + 3, inner), float("nan"), dtype=tdt, device=dev)
    a = L.LinearArgs(dtype=code, M=M, K=K, N=2 * inner, N_pad=2 * inner, x=X.data_ptr(), x_stride=K, w_packed=wp.data_ptr(),
                     bias=Bv.data_ptr(), residual=None, y=y.data_ptr(), glu=1)
    L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
    torch.cuda.synchronize()
    assert torch.isnan(y[M:].float()).all()
    assert rel(y[:M].float().cpu(), ref) < {"bf16": 4e-3, "fp16": 5e-4}[mode]


def test_linear_gemm_eight_phase_is_deterministic_under_load(env, monkeypatch):
    """Race screen: the kernel has no atomics and a fixed summation order, so every launch of the same operands must return the same
    bits.  A DMA piece that lands after its first read (or over a tile still being read) shows up as a changed output: 30 launches of
    two shapes (whole-chip grids, 20 and 80 K tiles) against the first one."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    monkeypatch.setenv("PD_LIN_P8", "1")
    g = torch.Generator().manual_seed(73)
    for M, K, N in ((8192, 1280, 10240), (8192, 5120, 1280)):
        x = torch.randn(M, K, generator=g).bfloat16().to(dev)
        w = pack_conv_weight((torch.randn(N, K, generator=g) / K ** 0.5)[:, :, None, None], torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        first = None
        for it in range(30):
            y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            a = L.LinearArgs(dtype=1, M=M, K=K, N=N, N_pad=N, x=x.data_ptr(), x_stride=K, w_packed=w.data_ptr(), bias=bias.data_ptr(),
                             residual=None, y=y.data_ptr())
            L.check(lib.pd_linear(C.byref(a), stream()), "pd_linear")
            torch.cuda.synchronize()
            if first is None:
                first = y
            else:
                assert torch.equal(y, first), (M, K, N, it)
