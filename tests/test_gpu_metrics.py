"""FID / IS / KID on MI355X (phendiff_amd/metrics.py + csrc/metric_kernels.hip) against the CPU oracle (oracle/inception_ref.py): the
kernels one by one against plain PyTorch, the 2048-d / logit features of the whole network, and the three scalars the reference logs
(utils_training.py:948-1001, utils_Img2Img.py:462-563).  Random-init network (no pretrained weights are obtainable here): structure parity."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import DT, bf16_round, env, rel, stream  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["f32", "bf16", "fp16"])
@pytest.mark.parametrize("cfg", [  # B, H, W, Cin, Cout, KH, KW, stride, ph, pw
    (2, 19, 23, 32, 32, 3, 3, 2, 0, 0), (2, 17, 17, 128, 160, 1, 7, 1, 0, 3), (2, 17, 17, 160, 192, 7, 1, 1, 3, 0), (3, 11, 9, 64, 96, 5, 5, 1, 2, 2),
    (2, 8, 8, 384, 384, 1, 3, 1, 0, 1), (2, 8, 8, 448, 384, 3, 3, 1, 1, 1), (1, 35, 35, 288, 96, 1, 1, 1, 0, 0), (5, 9, 9, 96, 96, 3, 3, 2, 0, 0)])
def test_conv_rect(env, mode, cfg):
    """pd_conv_rect against F.conv2d + bias + ReLU: every kernel shape InceptionV3 uses (3x3 stride 2 pad 0, 1x7, 7x1, 5x5, 1x3, 3x3, 1x1),
    an odd number of 32-channel output tiles, ragged pixel tiles, the output written into a channel slice of a wider tensor."""
    from phendiff_amd.packing import pack_conv_weight
    L, lib, _, dev = env
    code, tdt = DT[mode]
    B, H, W, Ci, Co, KH, KW, s, ph, pw = cfg
    g = torch.Generator().manual_seed(81)
    x = bf16_round(torch.randn(B, Ci, H, W, generator=g), mode)
    w = bf16_round(torch.randn(Co, Ci, KH, KW, generator=g) / (Ci * KH * KW) ** 0.5, mode)
    bias = torch.randn(Co, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), bias.double(), stride=s, padding=(ph, pw))).permute(0, 2, 3, 1)
    Ho, Wo = ref.shape[1:3]
    X = x.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev)
    wp, bp = pack_conv_weight(w, tdt, Co).to(dev), bias.to(dev)
    ycs, yco = Co + 64, 32
    y = torch.full((B, Ho, Wo, ycs), float("nan"), dtype=tdt, device=dev)
    a = L.ConvRectArgs(dtype=code, B=B, Hin=H, Win=W, Cin=Ci, Hout=Ho, Wout=Wo, Cout_pad=Co, KH=KH, KW=KW, stride=s, pad_h=ph, pad_w=pw, relu=1,
                       x=X.data_ptr(), x_cs=Ci, w_packed=wp.data_ptr(), bias=bp.data_ptr(), y=y.data_ptr(), y_cs=ycs, y_co=yco)
    L.check(lib.pd_conv_rect(C.byref(a), stream()), "pd_conv_rect")
    torch.cuda.synchronize()
    assert torch.isnan(y[..., :yco].float()).all() and torch.isnan(y[..., yco + Co:].float()).all()      # only its slice is written
    assert rel(y[..., yco:yco + Co].float(), ref.float()) < {"f32": 2e-6, "bf16": 4e-3, "fp16": 5e-4}[mode]
    a.Hout += 1
    assert lib.pd_conv_rect(C.byref(a), stream()) == -2                                                    # inconsistent output size: refused


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_pool2d_and_resize_and_fc(env, mode):
    L, lib, _, dev = env
    code, tdt = DT[mode]
    g = torch.Generator().manual_seed(82)
    x = bf16_round(torch.randn(2, 64, 17, 15, generator=g), mode)
    X = x.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev)
    for pmode, k, s, p, fn in ((0, 3, 2, 0, lambda t: F.max_pool2d(t, 3, 2)), (1, 3, 1, 1, lambda t: F.avg_pool2d(t, 3, 1, 1, count_include_pad=False)),
                               (0, 3, 1, 1, lambda t: F.max_pool2d(t, 3, 1, 1))):
        ref = fn(x).permute(0, 2, 3, 1)
        Ho, Wo = ref.shape[1:3]
        y = torch.full((2, Ho, Wo, 96), float("nan"), dtype=tdt, device=dev)
        a = L.Pool2dArgs(dtype=code, B=2, Hin=17, Win=15, C=64, Hout=Ho, Wout=Wo, k=k, stride=s, pad=p, mode=pmode, x=X.data_ptr(), x_cs=64,
                         y=y.data_ptr(), y_cs=96, y_co=32)
        L.check(lib.pd_pool2d(C.byref(a), stream()), "pd_pool2d")
        torch.cuda.synchronize()
        assert torch.isnan(y[..., :32].float()).all()
        assert rel(y[..., 32:].float(), ref) < (1e-6 if mode == "f32" else 4e-3)
    out = torch.empty(2, 64, device=dev)
    a = L.Pool2dArgs(dtype=code, B=2, Hin=17, Win=15, C=64, Hout=1, Wout=1, k=17, stride=1, pad=0, mode=2, x=X.data_ptr(), x_cs=64,
                     y=out.data_ptr(), y_cs=64, y_co=0)
    L.check(lib.pd_pool2d(C.byref(a), stream()), "pd_pool2d")
    assert rel(out, x.mean(dim=(2, 3))) < 2e-6
    # resize: uint8 NHWC -> 299 x 299, (v - 128) / 128, 32 channels (3 used)
    from oracle import tf1_bilinear_resize_ref
    u8 = torch.randint(0, 256, (3, 37, 52, 3), dtype=torch.uint8, generator=g)
    y = torch.full((3, 299, 299, 32), float("nan"), dtype=tdt, device=dev)
    U = u8.to(dev)
    a = L.ResizeTf1Args(dtype=code, N=3, H=37, W=52, OH=299, OW=299, scale_y=float(np.float32(37 / 299)), scale_x=float(np.float32(52 / 299)),
                        sub=128.0, div=128.0, x=U.data_ptr(), y=y.data_ptr())
    L.check(lib.pd_resize_tf1(C.byref(a), stream()), "pd_resize_tf1")
    torch.cuda.synchronize()
    ref = ((tf1_bilinear_resize_ref(u8.permute(0, 3, 1, 2).float(), (299, 299)) - 128) / 128).permute(0, 2, 3, 1)
    assert float(y[..., 3:].float().abs().max()) == 0.0
    if mode == "f32":
        assert float((y[..., :3].float().cpu() - ref).abs().max()) < 1e-6      # same fp32 operations in the same order
    else:
        assert rel(y[..., :3].float(), ref) < 4e-3
    f = torch.randn(5, 2048, generator=g)
    w, b = torch.randn(1008, 2048, generator=g) / 45, torch.randn(1008, generator=g)
    yo = torch.empty(5, 1008, device=dev)
    Fd, Wt, Bd = f.to(dev), w.t().contiguous().to(dev), b.to(dev)
    a = L.FcF32Args(rows=5, in_dim=2048, out_dim=1008, x=Fd.data_ptr(), wt=Wt.data_ptr(), bias=Bd.data_ptr(), y=yo.data_ptr())
    L.check(lib.pd_fc_f32(C.byref(a), stream()), "pd_fc_f32")
    assert rel(yo, F.linear(f.double(), w.double(), b.double()).float()) < 2e-6


def _synthetic_sets(n, size, seed):
    """Two image sets that differ (so FID / KID are far from zero): smooth random fields with different colour statistics."""
    g = torch.Generator().manual_seed(seed)
    def make(shift):
        low = torch.rand(n, 3, size // 4, size // 4, generator=g)
        img = F.interpolate(low, size=(size, size), mode="bilinear", align_corners=False) + 0.15 * torch.rand(n, 3, size, size, generator=g) + shift
        return (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    return make(torch.tensor([0.0, 0.0, 0.0]).view(1, 3, 1, 1)), make(torch.tensor([0.15, -0.1, 0.05]).view(1, 3, 1, 1))


_ORACLE = {}


def _oracle_side():
    """The CPU oracle's share, computed ONCE for the three engine modes (5.7 GMAC per image at 299 x 299 on the host cores): the seeded
    random-init network, 48 + 48 synthetic 48 x 48 images, their features and the three scalars."""
    if not _ORACLE:
        from oracle import InceptionV3FeaturesRef, calculate_metrics_ref, randomize_inception_
        ref = randomize_inception_(InceptionV3FeaturesRef(), seed=3)
        gen, real = _synthetic_sets(48, 48, 7)
        _ORACLE.update(ref=ref, gen=gen, real=real, feats=ref(torch.from_numpy(gen[:16]).permute(0, 3, 1, 2)),
                       metrics=calculate_metrics_ref(ref, gen, real, isc=True, fid=True, kid=True, kid_subset_size=24))
    return _ORACLE


# measured on MI355X (profiles/r6_parity_errors.json): features f32 8.1e-7 / bf16 3.4e-3 / fp16 4.6e-4 -> tolerances ~3x
@pytest.mark.parametrize("mode,tol,stol", [("f32", 5e-6, 1e-3), ("bf16", 1e-2, 8e-2), ("fp16", 1.5e-3, 1e-2)])
def test_inception_features_and_the_three_scalars_vs_oracle(mode, tol, stol):
    """48 + 48 synthetic images through the HIP InceptionV3 (random-init: structure parity) and through the oracle: pool3 features,
    un-biased logits, and FID / IS / KID (kid_subset_size 24) as the reference's calculate_metrics call returns them."""
    import phendiff_amd.metrics as M
    o = _oracle_side()
    ref, gen, real, want, m_ref = o["ref"], o["gen"], o["real"], o["feats"], o["metrics"]
    net = M.InceptionV3Features(mode)
    net.load_state_dict(ref.state_dict())
    net = net.to("cuda:0")
    got = net(torch.from_numpy(gen[:16]).cuda())
    for k in ("2048", "logits_unbiased", "logits"):
        assert rel(got[k], want[k]) < tol, k
    got_nchw = net(torch.from_numpy(gen[:16]).permute(0, 3, 1, 2).contiguous().cuda())           # NCHW input form
    assert torch.equal(got_nchw["2048"], got["2048"])
    m_got = M.calculate_metrics(net, gen, real, isc=True, fid=True, kid=True, kid_subset_size=24, batch_size=48)
    assert set(m_got) == set(m_ref) == {"inception_score_mean", "inception_score_std", "frechet_inception_distance",
                                        "kernel_inception_distance_mean", "kernel_inception_distance_std"}
    assert m_ref["frechet_inception_distance"] > 1.0 and m_ref["kernel_inception_distance_mean"] > 0     # the two sets DO differ
    for k in ("inception_score_mean", "frechet_inception_distance", "kernel_inception_distance_mean"):
        assert abs(m_got[k] - m_ref[k]) < stol * abs(m_ref[k]), (k, m_got[k], m_ref[k])
    if mode == "bf16":      # float images in [0, 1] are quantised like the reference's PNG files
        again = M.calculate_metrics(net, gen.astype(np.float32) / 255.0, real, isc=False, fid=True, batch_size=48)
        assert again["frechet_inception_distance"] == m_got["frechet_inception_distance"]


def test_eval_generation_hook_computes_class_metrics():
    """`generate_samples(..., on_class_done=class_metrics_hook(...))`: what `_compute_log_metrics` does per class (utils_training.py:948-1001)."""
    import phendiff_amd as P
    import phendiff_amd.metrics as M
    from oracle import InceptionV3FeaturesRef, randomize_inception_
    net = M.InceptionV3Features("bf16")
    net.load_state_dict(randomize_inception_(InceptionV3FeaturesRef(), seed=4).state_dict())
    net = net.to("cuda:0")
    torch.manual_seed(0)
    unet = P.CustomCondUNet2DModel(compute_dtype="bf16", **dict(P.UNET_CONFIGS["super_small"], sample_size=32)).to("cuda:0")
    pipe = P.ConditionalDDIMPipeline(unet, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    real0, real1 = _synthetic_sets(12, 32, 9)
    results = {}
    hook = M.class_metrics_hook(net, {0: real0, 1: real1}, results, isc=True, fid=True, kid=True, kid_subset_size=6, batch_size=16)
    from phendiff_amd.eval_generation import generate_samples
    generate_samples(pipe, nb_classes=2, nb_generated_images=12, eval_batch_size=8, num_inference_steps=2, class_names=["a", "b"], on_class_done=hook)
    assert set(results) == {f"{m}/{c}" for c in "ab" for m in ("inception_score_mean", "inception_score_std", "frechet_inception_distance",
                                                                "kernel_inception_distance_mean", "kernel_inception_distance_std")}
    assert all(np.isfinite(v) for v in results.values()) and results["frechet_inception_distance/a"] > 0
