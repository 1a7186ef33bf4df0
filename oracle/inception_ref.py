"""TEST INFRASTRUCTURE (CPU oracle) -- FID / IS / KID as ``torch_fidelity.calculate_metrics`` computes them for the reference
(``src/utils_training.py:948-1001`` per class at evaluation time, ``src/utils_Img2Img.py:462-563`` after a class-transfer experiment;
both with the library's defaults: feature extractor ``inception-v3-compat``, FID / KID on the 2048-d pool3 features, IS on the
1008 un-biased logits, 10 IS splits, 100 KID subsets of ``kid_subset_size``, polynomial kernel degree 3 / gamma 1/d / coef0 1, RNG
seed 2020).

The arithmetic lives in ``torch-fidelity==0.3.0`` (``environment.yaml:382``), which is absent from /root/reference and not
installable here; this file restates its published algorithm:
  * ``FeatureExtractorInceptionV3`` = the TF-Slim InceptionV3 of the original FID code: uint8 images -> float -> TF1-style bilinear
    resize to 299 x 299 (no half-pixel centres) -> (x - 128) / 128 -> the torchvision block structure with two FID-specific changes
    (average pools with ``count_include_pad=False``; the LAST block pools with max, reproducing the original graph) -> global
    average pool (2048) -> fc 2048 -> 1008.
  * ``fid_statistics_to_metric`` / ``isc_features_to_metric`` / ``kid_features_to_metric`` in fp64.
No pretrained weights are obtainable here (no network): the network is RANDOM-INIT -- structure parity only, and the header says so
wherever a number is quoted.  Known answer that pins the structure: with a 1000-way fc the parameter count is torchvision's
``inception_v3`` without its auxiliary head, 23 834 568 (tests/test_oracle_metrics.py).  Parity unpinned (see oracle/__init__.py).
Module names follow torch-fidelity's state_dict (``Conv2d_1a_3x3.conv.weight`` ...), so its ``pt_inception-2015-12-05`` weights load."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def tf1_bilinear_resize_ref(x: torch.Tensor, size=(299, 299)) -> torch.Tensor:
    """``interpolate_bilinear_2d_like_tensorflow1x(x, size, align_corners=False)``: source coordinate = destination index *
    (in / out) -- no half-pixel offset --, floor / +1 (clamped) neighbours, separable lerp."""
    N, C, H, W = x.shape
    oh, ow = size
    sy, sx = H / oh, W / ow
    gy = torch.arange(oh, dtype=x.dtype) * sy
    gx = torch.arange(ow, dtype=x.dtype) * sx
    y0, x0 = gy.long(), gx.long()
    y1, x1 = (y0 + 1).clamp_max(H - 1), (x0 + 1).clamp_max(W - 1)
    dy, dx = (gy - y0.to(x.dtype)).view(1, 1, oh, 1), (gx - x0.to(x.dtype)).view(1, 1, 1, ow)
    r0, r1 = x[:, :, y0, :], x[:, :, y1, :]
    i00, i01, i10, i11 = r0[:, :, :, x0], r0[:, :, :, x1], r1[:, :, :, x0], r1[:, :, :, x1]
    top = i00 + (i01 - i00) * dx
    bot = i10 + (i11 - i10) * dx
    return top + (bot - top) * dy


class BasicConv2dRef(nn.Module):
    def __init__(self, cin, cout, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, bias=False, **kw)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)))


class InceptionARef(nn.Module):
    def __init__(self, cin, pool_features):
        super().__init__()
        C = BasicConv2dRef
        self.branch1x1 = C(cin, 64, kernel_size=1)
        self.branch5x5_1 = C(cin, 48, kernel_size=1)
        self.branch5x5_2 = C(48, 64, kernel_size=5, padding=2)
        self.branch3x3dbl_1 = C(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = C(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = C(96, 96, kernel_size=3, padding=1)
        self.branch_pool = C(cin, pool_features, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b5 = self.branch5x5_2(self.branch5x5_1(x))
        b3 = self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x)))
        bp = self.branch_pool(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))
        return torch.cat([b1, b5, b3, bp], 1)


class InceptionBRef(nn.Module):
    def __init__(self, cin):
        super().__init__()
        C = BasicConv2dRef
        self.branch3x3 = C(cin, 384, kernel_size=3, stride=2)
        self.branch3x3dbl_1 = C(cin, 64, kernel_size=1)
        self.branch3x3dbl_2 = C(64, 96, kernel_size=3, padding=1)
        self.branch3x3dbl_3 = C(96, 96, kernel_size=3, stride=2)

    def forward(self, x):
        return torch.cat([self.branch3x3(x), self.branch3x3dbl_3(self.branch3x3dbl_2(self.branch3x3dbl_1(x))), F.max_pool2d(x, 3, 2)], 1)


class InceptionCRef(nn.Module):
    def __init__(self, cin, c7):
        super().__init__()
        C = BasicConv2dRef
        self.branch1x1 = C(cin, 192, kernel_size=1)
        self.branch7x7_1 = C(cin, c7, kernel_size=1)
        self.branch7x7_2 = C(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7_3 = C(c7, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = C(cin, c7, kernel_size=1)
        self.branch7x7dbl_2 = C(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = C(c7, c7, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = C(c7, c7, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = C(c7, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch_pool = C(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        b7 = self.branch7x7_3(self.branch7x7_2(self.branch7x7_1(x)))
        bd = self.branch7x7dbl_5(self.branch7x7dbl_4(self.branch7x7dbl_3(self.branch7x7dbl_2(self.branch7x7dbl_1(x)))))
        bp = self.branch_pool(F.avg_pool2d(x, 3, 1, 1, count_include_pad=False))
        return torch.cat([b1, b7, bd, bp], 1)


class InceptionDRef(nn.Module):
    def __init__(self, cin):
        super().__init__()
        C = BasicConv2dRef
        self.branch3x3_1 = C(cin, 192, kernel_size=1)
        self.branch3x3_2 = C(192, 320, kernel_size=3, stride=2)
        self.branch7x7x3_1 = C(cin, 192, kernel_size=1)
        self.branch7x7x3_2 = C(192, 192, kernel_size=(1, 7), padding=(0, 3))
        self.branch7x7x3_3 = C(192, 192, kernel_size=(7, 1), padding=(3, 0))
        self.branch7x7x3_4 = C(192, 192, kernel_size=3, stride=2)

    def forward(self, x):
        b3 = self.branch3x3_2(self.branch3x3_1(x))
        b7 = self.branch7x7x3_4(self.branch7x7x3_3(self.branch7x7x3_2(self.branch7x7x3_1(x))))
        return torch.cat([b3, b7, F.max_pool2d(x, 3, 2)], 1)


class InceptionERef(nn.Module):
    """``pool``: "avg" = FIDInceptionE_1 (count_include_pad=False), "max" = FIDInceptionE_2 (the original graph's last block)."""

    def __init__(self, cin, pool):
        super().__init__()
        C = BasicConv2dRef
        self.pool = pool
        self.branch1x1 = C(cin, 320, kernel_size=1)
        self.branch3x3_1 = C(cin, 384, kernel_size=1)
        self.branch3x3_2a = C(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3_2b = C(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = C(cin, 448, kernel_size=1)
        self.branch3x3dbl_2 = C(448, 384, kernel_size=3, padding=1)
        self.branch3x3dbl_3a = C(384, 384, kernel_size=(1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = C(384, 384, kernel_size=(3, 1), padding=(1, 0))
        self.branch_pool = C(cin, 192, kernel_size=1)

    def forward(self, x):
        b1 = self.branch1x1(x)
        t = self.branch3x3_1(x)
        b3 = torch.cat([self.branch3x3_2a(t), self.branch3x3_2b(t)], 1)
        t = self.branch3x3dbl_2(self.branch3x3dbl_1(x))
        bd = torch.cat([self.branch3x3dbl_3a(t), self.branch3x3dbl_3b(t)], 1)
        p = F.avg_pool2d(x, 3, 1, 1, count_include_pad=False) if self.pool == "avg" else F.max_pool2d(x, 3, 1, 1)
        return torch.cat([b1, b3, bd, self.branch_pool(p)], 1)


class InceptionV3FeaturesRef(nn.Module):
    """torch-fidelity ``FeatureExtractorInceptionV3`` ("inception-v3-compat"): ``forward(uint8 NCHW)`` -> dict with the features the
    three metrics use: "2048" (pool3), "logits_unbiased" (x @ fc.weight^T), "logits" (+ fc.bias)."""

    def __init__(self, num_logits: int = 1008):
        super().__init__()
        C = BasicConv2dRef
        self.Conv2d_1a_3x3 = C(3, 32, kernel_size=3, stride=2)
        self.Conv2d_2a_3x3 = C(32, 32, kernel_size=3)
        self.Conv2d_2b_3x3 = C(32, 64, kernel_size=3, padding=1)
        self.Conv2d_3b_1x1 = C(64, 80, kernel_size=1)
        self.Conv2d_4a_3x3 = C(80, 192, kernel_size=3)
        self.Mixed_5b, self.Mixed_5c, self.Mixed_5d = InceptionARef(192, 32), InceptionARef(256, 64), InceptionARef(288, 64)
        self.Mixed_6a = InceptionBRef(288)
        self.Mixed_6b, self.Mixed_6c = InceptionCRef(768, 128), InceptionCRef(768, 160)
        self.Mixed_6d, self.Mixed_6e = InceptionCRef(768, 160), InceptionCRef(768, 192)
        self.Mixed_7a = InceptionDRef(768)
        self.Mixed_7b, self.Mixed_7c = InceptionERef(1280, "avg"), InceptionERef(2048, "max")
        self.fc = nn.Linear(2048, num_logits)

    def trunk(self, x):
        x = self.Conv2d_2b_3x3(self.Conv2d_2a_3x3(self.Conv2d_1a_3x3(x)))
        x = F.max_pool2d(x, 3, 2)
        x = self.Conv2d_4a_3x3(self.Conv2d_3b_1x1(x))
        x = F.max_pool2d(x, 3, 2)
        for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e", "Mixed_7a",
                     "Mixed_7b", "Mixed_7c"):
            x = getattr(self, name)(x)
        return x

    @torch.no_grad()
    def forward(self, images_u8: torch.Tensor):
        assert images_u8.dtype == torch.uint8 and images_u8.ndim == 4 and images_u8.shape[1] == 3, "Expecting uint8 images (N, 3, H, W)"
        x = tf1_bilinear_resize_ref(images_u8.float(), (299, 299))
        x = (x - 128) / 128
        x = self.trunk(x)
        pool = F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)
        unbiased = pool.mm(self.fc.weight.T)
        return {"2048": pool, "logits_unbiased": unbiased, "logits": unbiased + self.fc.bias.unsqueeze(0)}


def randomize_inception_(m: nn.Module, seed: int = 0):
    """Random init that keeps activations O(1) through 94 conv + BatchNorm (eval) + ReLU layers, so that a wrong layer shows up in the
    features instead of vanishing: He-normal convolutions, BatchNorm gamma in [0.8, 1.2], beta in [-0.2, 0.4], running mean ~ N(0, 0.1),
    running variance in [0.5, 1.5].  No pretrained weights are obtainable here."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, nn.Conv2d):
                fan_in = mod.weight.shape[1] * mod.weight.shape[2] * mod.weight.shape[3]
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
            elif isinstance(mod, nn.BatchNorm2d):
                mod.weight.copy_(0.8 + 0.4 * torch.rand(mod.weight.shape, generator=g))
                mod.bias.copy_(-0.2 + 0.6 * torch.rand(mod.bias.shape, generator=g))
                mod.running_mean.copy_(0.1 * torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(0.5 + torch.rand(mod.running_var.shape, generator=g))
            elif isinstance(mod, nn.Linear):
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * (1.0 / mod.weight.shape[1]) ** 0.5)
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    return m.eval()


# ---- the three metrics (torch_fidelity metric_fid.py / metric_isc.py / metric_kid.py), fp64 ----------------------------------------
def fid_statistics_ref(features):
    f = np.asarray(features, dtype=np.float64)
    return np.mean(f, axis=0), np.cov(f, rowvar=False)


def fid_from_statistics_ref(mu1, sigma1, mu2, sigma2, eps=1e-6):
    import scipy.linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = scipy.linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = scipy.linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        assert np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3), "Imaginary component"
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def isc_ref(logits_unbiased, splits=10, shuffle=True, rng_seed=2020):
    f = torch.as_tensor(np.asarray(logits_unbiased))
    N = f.shape[0]
    if shuffle:
        f = f[np.random.RandomState(rng_seed).permutation(N), :]
    f = f.double()
    p, log_p = f.softmax(dim=1), f.log_softmax(dim=1)
    scores = []
    for i in range(splits):
        pc, lpc = p[(i * N // splits):((i + 1) * N // splits)], log_p[(i * N // splits):((i + 1) * N // splits)]
        q = pc.mean(dim=0, keepdim=True)
        scores.append((pc * (lpc - q.log())).sum(dim=1).mean().exp().item())
    return {"inception_score_mean": float(np.mean(scores)), "inception_score_std": float(np.std(scores))}


def _mmd2_ref(K_XX, K_XY, K_YY):
    m = K_XX.shape[0]
    Kt_XX_sum = (K_XX.sum(axis=1) - np.diagonal(K_XX)).sum()
    Kt_YY_sum = (K_YY.sum(axis=1) - np.diagonal(K_YY)).sum()
    return (Kt_XX_sum + Kt_YY_sum) / (m * (m - 1)) - 2 * K_XY.sum() / (m * m)


def kid_ref(features_1, features_2, kid_subsets=100, kid_subset_size=1000, degree=3, gamma=None, coef0=1, rng_seed=2020):
    f1, f2 = np.asarray(features_1, dtype=np.float64), np.asarray(features_2, dtype=np.float64)
    assert kid_subset_size <= len(f1) and kid_subset_size <= len(f2), "kid_subset_size must not exceed the number of samples"
    rng = np.random.RandomState(rng_seed)
    k = lambda X, Y: (X @ Y.T * (gamma if gamma is not None else 1.0 / X.shape[1]) + coef0) ** degree      # noqa: E731
    mmds = np.zeros(kid_subsets)
    for i in range(kid_subsets):
        a = f1[rng.choice(len(f1), kid_subset_size, replace=False)]
        b = f2[rng.choice(len(f2), kid_subset_size, replace=False)]
        mmds[i] = _mmd2_ref(k(a, a), k(a, b), k(b, b))
    return {"kernel_inception_distance_mean": float(np.mean(mmds)), "kernel_inception_distance_std": float(np.std(mmds))}


def calculate_metrics_ref(net: InceptionV3FeaturesRef, images1_u8, images2_u8, isc=True, fid=True, kid=False, kid_subset_size=1000,
                          batch_size=64):
    """``torch_fidelity.calculate_metrics(input1, input2, isc=, fid=, kid=, kid_subset_size=)`` on uint8 NHWC arrays: IS of input1,
    FID / KID between input1 and input2."""
    def feats(u8):
        outs = [net(torch.from_numpy(np.ascontiguousarray(u8[i:i + batch_size])).permute(0, 3, 1, 2)) for i in range(0, len(u8), batch_size)]
        return {k: torch.cat([o[k] for o in outs]).double().numpy() for k in outs[0]}
    f1 = feats(images1_u8)
    out = {}
    if isc:
        out.update(isc_ref(f1["logits_unbiased"]))
    if fid or kid:
        f2 = feats(images2_u8)
        if fid:
            out["frechet_inception_distance"] = fid_from_statistics_ref(*fid_statistics_ref(f1["2048"]), *fid_statistics_ref(f2["2048"]))
        if kid:
            out.update(kid_ref(f1["2048"], f2["2048"], kid_subset_size=kid_subset_size))
    return out
