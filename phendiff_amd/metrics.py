"""FID / IS / KID of generated images, as the reference computes them with ``torch_fidelity.calculate_metrics`` at evaluation time
(``src/utils_training.py:948-1001``: per class, generated folder vs the class's real images, ``isc`` / ``fid`` / ``kid`` /
``kid_subset_size`` from the arguments) and after a class-transfer experiment (``src/utils_Img2Img.py:462-563``).

torch-fidelity's defaults are what the reference runs with: the ``inception-v3-compat`` feature extractor (the TF-Slim InceptionV3 of the
original FID code: uint8 image -> TF1-style bilinear resize to 299 x 299 -> (x - 128) / 128 -> 94 conv + BatchNorm + ReLU layers, FID's two
pooling quirks -> 2048-d pool3 features -> fc to 1008 logits), FID and KID on the 2048-d features, IS on the un-biased logits, 10 IS
splits, 100 KID subsets, polynomial kernel (degree 3, gamma 1/d, coef0 1), RNG seed 2020.

Here the network runs on the HIP engine (``csrc/metric_kernels.hip``: ``pd_resize_tf1``, ``pd_conv_rect`` with the BatchNorm folded into
weights and bias at pack time and the branch concatenations written in place, ``pd_pool2d``, ``pd_fc_f32``); the statistics are fp64 on the
host, like the library's (numpy / scipy).  Deviation, stated: the reference writes PNG files and torch-fidelity reads them back; this module
takes the uint8 arrays themselves (``round(255 x)`` of the pipeline's float output -- the bytes the PNGs would hold), so there is no file
round trip.  The module tree carries torch-fidelity's state_dict names: its published ``pt_inception-2015-12-05`` weights load with
``load_state_dict``.  They are not obtainable here (no network), so tests and examples run the structure on seeded random weights."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .packing import pack_conv_weight

_DT = {"f32": (L.PD_F32 if hasattr(L, "PD_F32") else 0, torch.float32), "bf16": (1, torch.bfloat16), "fp16": (2, torch.float16)}


def _c32(c: int) -> int:
    return (c + 31) // 32 * 32


class BasicConv2d(nn.Module):
    """conv (no bias) + BatchNorm(eps 1e-3, inference statistics) + ReLU: torch-fidelity / torchvision ``BasicConv2d``."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__()
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
        ph, pw = (padding, padding) if isinstance(padding, int) else padding
        self.geom = (cin, cout, kh, kw, stride, ph, pw)
        self.conv = nn.Conv2d(cin, cout, (kh, kw), stride, (ph, pw), bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=0.001)


# block tables: every entry of a branch is (attribute name, cin, cout, kernel, stride, padding) or a pooling marker; a branch's LAST
# convolutions (one, or the two of InceptionE's split) are written into the block's concatenated output in table order
def _A(cin, pf):
    return [[("branch1x1", cin, 64, 1, 1, 0)],
            [("branch5x5_1", cin, 48, 1, 1, 0), ("branch5x5_2", 48, 64, 5, 1, 2)],
            [("branch3x3dbl_1", cin, 64, 1, 1, 0), ("branch3x3dbl_2", 64, 96, 3, 1, 1), ("branch3x3dbl_3", 96, 96, 3, 1, 1)],
            ["avg311", ("branch_pool", cin, pf, 1, 1, 0)]]


def _B(cin):
    return [[("branch3x3", cin, 384, 3, 2, 0)],
            [("branch3x3dbl_1", cin, 64, 1, 1, 0), ("branch3x3dbl_2", 64, 96, 3, 1, 1), ("branch3x3dbl_3", 96, 96, 3, 2, 0)],
            ["max320"]]


def _C(cin, c7):
    return [[("branch1x1", cin, 192, 1, 1, 0)],
            [("branch7x7_1", cin, c7, 1, 1, 0), ("branch7x7_2", c7, c7, (1, 7), 1, (0, 3)), ("branch7x7_3", c7, 192, (7, 1), 1, (3, 0))],
            [("branch7x7dbl_1", cin, c7, 1, 1, 0), ("branch7x7dbl_2", c7, c7, (7, 1), 1, (3, 0)), ("branch7x7dbl_3", c7, c7, (1, 7), 1, (0, 3)),
             ("branch7x7dbl_4", c7, c7, (7, 1), 1, (3, 0)), ("branch7x7dbl_5", c7, 192, (1, 7), 1, (0, 3))],
            ["avg311", ("branch_pool", cin, 192, 1, 1, 0)]]


def _D(cin):
    return [[("branch3x3_1", cin, 192, 1, 1, 0), ("branch3x3_2", 192, 320, 3, 2, 0)],
            [("branch7x7x3_1", cin, 192, 1, 1, 0), ("branch7x7x3_2", 192, 192, (1, 7), 1, (0, 3)), ("branch7x7x3_3", 192, 192, (7, 1), 1, (3, 0)),
             ("branch7x7x3_4", 192, 192, 3, 2, 0)],
            ["max320"]]


def _E(cin, pool):
    return [[("branch1x1", cin, 320, 1, 1, 0)],
            [("branch3x3_1", cin, 384, 1, 1, 0), [("branch3x3_2a", 384, 384, (1, 3), 1, (0, 1)), ("branch3x3_2b", 384, 384, (3, 1), 1, (1, 0))]],
            [("branch3x3dbl_1", cin, 448, 1, 1, 0), ("branch3x3dbl_2", 448, 384, 3, 1, 1),
             [("branch3x3dbl_3a", 384, 384, (1, 3), 1, (0, 1)), ("branch3x3dbl_3b", 384, 384, (3, 1), 1, (1, 0))]],
            [pool, ("branch_pool", cin, 192, 1, 1, 0)]]


STEM = [("Conv2d_1a_3x3", 3, 32, 3, 2, 0), ("Conv2d_2a_3x3", 32, 32, 3, 1, 0), ("Conv2d_2b_3x3", 32, 64, 3, 1, 1), "max320",
        ("Conv2d_3b_1x1", 64, 80, 1, 1, 0), ("Conv2d_4a_3x3", 80, 192, 3, 1, 0), "max320"]
BLOCKS = [("Mixed_5b", _A(192, 32)), ("Mixed_5c", _A(256, 64)), ("Mixed_5d", _A(288, 64)), ("Mixed_6a", _B(288)),
          ("Mixed_6b", _C(768, 128)), ("Mixed_6c", _C(768, 160)), ("Mixed_6d", _C(768, 160)), ("Mixed_6e", _C(768, 192)),
          ("Mixed_7a", _D(768)), ("Mixed_7b", _E(1280, "avg311")), ("Mixed_7c", _E(2048, "max311"))]
_POOLS = {"max320": (0, 3, 2, 0), "avg311": (1, 3, 1, 1), "max311": (0, 3, 1, 1)}      # marker -> (mode, k, stride, pad)


def _flat(entries):
    for e in entries:
        if isinstance(e, list):
            yield from _flat(e)
        elif isinstance(e, tuple):
            yield e


class InceptionV3Features(nn.Module):
    """torch-fidelity ``FeatureExtractorInceptionV3`` ("inception-v3-compat") on the HIP engine.  ``forward(images)``: uint8 images,
    NHWC (N, H, W, 3) or NCHW (N, 3, H, W), on the GPU -> {"2048": pool3 features, "logits_unbiased", "logits"} (fp32, on the device)."""

    def __init__(self, compute_dtype: str = "bf16", num_logits: int = 1008):
        super().__init__()
        if compute_dtype not in _DT:
            raise ValueError(f"compute_dtype {compute_dtype!r} not in {sorted(_DT)}")
        self.compute_dtype = compute_dtype
        for e in STEM:
            if isinstance(e, tuple):
                setattr(self, e[0], BasicConv2d(*e[1:]))
        for name, branches in BLOCKS:
            blk = nn.Module()
            for e in _flat(branches):
                setattr(blk, e[0], BasicConv2d(*e[1:]))
            setattr(self, name, blk)
        self.fc = nn.Linear(2048, num_logits)
        self._packed = None
        self._plans: Dict[tuple, "_Plan"] = {}
        self.eval()

    def invalidate(self):
        """Drop the packed weights and plans (after loading / changing parameters)."""
        self._packed, self._plans = None, {}

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return self.fc.weight.device

    # ---- weights: BatchNorm folded, channels padded to multiples of 32, MFMA fragment order -------------------------------------------
    def _pack(self):
        tdt = _DT[self.compute_dtype][1]
        out = {}
        for prefix, mod in self.named_modules():
            if not isinstance(mod, BasicConv2d):
                continue
            cin, cout, kh, kw, stride, ph, pw = mod.geom
            bn = mod.bn
            g = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
            w = mod.conv.weight.detach().double() * g.view(-1, 1, 1, 1)
            b = bn.bias.detach().double() - bn.running_mean.detach().double() * g
            cip, cop = _c32(cin), _c32(cout)
            wp = torch.zeros((cout, cip, kh, kw), dtype=torch.float32, device=w.device)
            wp[:, :cin] = w.float()
            bp = torch.zeros(cop, dtype=torch.float32, device=w.device)
            bp[:cout] = b.float()
            out[prefix] = (pack_conv_weight(wp, tdt, cop).contiguous(), bp.contiguous(), (cip, cop, kh, kw, stride, ph, pw))
        out["fc"] = (self.fc.weight.detach().float().t().contiguous(), self.fc.bias.detach().float().contiguous())
        return out

    @torch.no_grad()
    def forward(self, images: torch.Tensor):
        if not images.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback): move the images to 'cuda'")
        if images.dtype != torch.uint8 or images.ndim != 4:
            raise ValueError("Expecting uint8 images of shape (N, H, W, 3) or (N, 3, H, W)")
        if images.shape[-1] != 3:
            if images.shape[1] != 3:
                raise ValueError(f"Expecting 3-channel images, got {tuple(images.shape)}")
            images = images.permute(0, 2, 3, 1)
        images = images.contiguous()
        if self._packed is None:
            self._packed = self._pack()
        N, H, W, _ = images.shape
        key = (N, H, W)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = _Plan(self, N, H, W)
        return plan.run(images)


class _Plan:
    """Static launch plan for N images of H x W: every activation buffer (NHWC, channels padded to 32) and pre-filled argument struct."""

    def __init__(self, net: InceptionV3Features, N: int, H: int, W: int):
        self.lib = L.lib()
        self.code, self.tdt = _DT[net.compute_dtype]
        self.dev, self.N = net.device, N
        self.ops, self.bufs = [], []
        P = net._packed
        x = self._buf(299, 299, 32)
        self.resize = L.ResizeTf1Args(dtype=self.code, N=N, H=H, W=W, OH=299, OW=299, scale_y=float(np.float32(H / 299)),
                                      scale_x=float(np.float32(W / 299)), sub=128.0, div=128.0, x=None, y=x.data_ptr())
        for e in STEM:
            x = self._conv(P, e[0], x) if isinstance(e, tuple) else self._pool(x, e)
        for name, branches in BLOCKS:
            x = self._block(P, name, branches, x)
        self.pool = torch.empty((N, 2048), dtype=torch.float32, device=self.dev)
        h, w, c = x.shape[1:]
        self.ops.append((self.lib.pd_pool2d, L.Pool2dArgs(dtype=self.code, B=N, Hin=h, Win=w, C=c, Hout=1, Wout=1, k=h, stride=1, pad=0, mode=2,
                                                          x=x.data_ptr(), x_cs=c, y=self.pool.data_ptr(), y_cs=c, y_co=0), "pd_pool2d"))
        wt, bias = P["fc"]
        self.logits_u = torch.empty((N, wt.shape[1]), dtype=torch.float32, device=self.dev)
        self.logits = torch.empty_like(self.logits_u)
        for y, b in ((self.logits_u, None), (self.logits, bias)):
            self.ops.append((self.lib.pd_fc_f32, L.FcF32Args(rows=N, in_dim=wt.shape[0], out_dim=wt.shape[1], x=self.pool.data_ptr(), wt=wt.data_ptr(),
                                                             bias=L.ptr(b), y=y.data_ptr()), "pd_fc_f32"))

    def _buf(self, h, w, c):
        t = torch.empty((self.N, h, w, c), dtype=self.tdt, device=self.dev)
        self.bufs.append(t)
        return t

    def _conv(self, P, name, x, out=None, co=0):
        wp, bp, (cip, cop, kh, kw, stride, ph, pw) = P[name]
        _, hin, win, cs = x.shape
        assert cs == cip, (name, cs, cip)
        hout, wout = (hin + 2 * ph - kh) // stride + 1, (win + 2 * pw - kw) // stride + 1
        if out is None:
            out = self._buf(hout, wout, cop)
        a = L.ConvRectArgs(dtype=self.code, B=self.N, Hin=hin, Win=win, Cin=cip, Hout=hout, Wout=wout, Cout_pad=cop, KH=kh, KW=kw, stride=stride,
                           pad_h=ph, pad_w=pw, relu=1, x=x.data_ptr(), x_cs=cs, w_packed=wp.data_ptr(), bias=bp.data_ptr(), y=out.data_ptr(),
                           y_cs=out.shape[3], y_co=co)
        self.ops.append((self.lib.pd_conv_rect, a, f"pd_conv_rect {name}"))
        return out

    def _pool(self, x, marker, out=None, co=0):
        mode, k, stride, pad = _POOLS[marker]
        _, hin, win, c = x.shape
        hout, wout = (hin + 2 * pad - k) // stride + 1, (win + 2 * pad - k) // stride + 1
        if out is None:
            out = self._buf(hout, wout, c)
        a = L.Pool2dArgs(dtype=self.code, B=self.N, Hin=hin, Win=win, C=c, Hout=hout, Wout=wout, k=k, stride=stride, pad=pad, mode=mode,
                         x=x.data_ptr(), x_cs=c, y=out.data_ptr(), y_cs=out.shape[3], y_co=co)
        self.ops.append((self.lib.pd_pool2d, a, f"pd_pool2d {marker}"))
        return out

    def _block(self, P, name, branches, x):
        """One Inception block: each branch's last layer(s) write their channel slice of the concatenated output directly."""
        def heads(br):
            last = br[-1]
            return last if isinstance(last, list) else [last]
        widths = []
        for br in branches:
            for hd in heads(br):
                widths.append(x.shape[3] if isinstance(hd, str) else _c32(hd[2]))
        first = branches[0][0]
        stride = 2 if any(isinstance(e, tuple) and e[4] == 2 for br in branches for e in _flat(br)) else 1
        h = (x.shape[1] - 3) // 2 + 1 if stride == 2 else x.shape[1]
        w = (x.shape[2] - 3) // 2 + 1 if stride == 2 else x.shape[2]
        out = self._buf(h, w, sum(widths))
        co = 0
        for br in branches:
            t = x
            for e in br[:-1]:
                t = self._pool(t, e) if isinstance(e, str) else self._conv(P, f"{name}.{e[0]}", t)
            for hd in heads(br):
                if isinstance(hd, str):
                    self._pool(t, hd, out, co)
                    co += t.shape[3]
                else:
                    self._conv(P, f"{name}.{hd[0]}", t, out, co)
                    co += _c32(hd[2])
        assert co == out.shape[3] and first is not None
        return out

    def run(self, images_u8: torch.Tensor):
        st = torch.cuda.current_stream(self.dev).cuda_stream
        self.resize.x = images_u8.data_ptr()
        L.check(self.lib.pd_resize_tf1(C.byref(self.resize), st), "pd_resize_tf1")
        for fn, a, what in self.ops:
            L.check(fn(C.byref(a), st), what)
        self._keep = images_u8
        return {"2048": self.pool.clone(), "logits_unbiased": self.logits_u.clone(), "logits": self.logits.clone()}


# ---- the three metrics, fp64 on the host (torch_fidelity metric_fid.py / metric_isc.py / metric_kid.py) ---------------------------------
KEY_FID, KEY_ISC_MEAN, KEY_ISC_STD = "frechet_inception_distance", "inception_score_mean", "inception_score_std"
KEY_KID_MEAN, KEY_KID_STD = "kernel_inception_distance_mean", "kernel_inception_distance_std"


def fid_statistics(features) -> tuple:
    f = np.asarray(features, dtype=np.float64)
    return np.mean(f, axis=0), np.cov(f, rowvar=False)


def fid_from_statistics(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """||mu1 - mu2||^2 + Tr(S1) + Tr(S2) - 2 Tr(sqrtm(S1 S2)); a singular product is retried with eps on both diagonals."""
    import scipy.linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = scipy.linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = scipy.linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("FID: the matrix square root has an imaginary component")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def inception_score(logits_unbiased, splits: int = 10, shuffle: bool = True, rng_seed: int = 2020) -> dict:
    """exp(mean_x KL(p(y|x) || p(y))) per split of the (shuffled) samples; mean and std over the splits."""
    f = np.asarray(logits_unbiased, dtype=np.float64)
    N = f.shape[0]
    if shuffle:
        f = f[np.random.RandomState(rng_seed).permutation(N), :]
    z = f - f.max(axis=1, keepdims=True)
    log_p = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    p = np.exp(log_p)
    scores = []
    for i in range(splits):
        sl = slice(i * N // splits, (i + 1) * N // splits)
        q = p[sl].mean(axis=0, keepdims=True)
        scores.append(float(np.exp((p[sl] * (log_p[sl] - np.log(q))).sum(axis=1).mean())))
    return {KEY_ISC_MEAN: float(np.mean(scores)), KEY_ISC_STD: float(np.std(scores))}


def kernel_inception_distance(features_1, features_2, kid_subsets: int = 100, kid_subset_size: int = 1000, degree: int = 3,
                              gamma: Optional[float] = None, coef0: float = 1, rng_seed: int = 2020) -> dict:
    """Unbiased MMD^2 under the polynomial kernel (x . y / d + 1)^3 over ``kid_subsets`` random subsets of ``kid_subset_size`` samples."""
    f1, f2 = np.asarray(features_1, dtype=np.float64), np.asarray(features_2, dtype=np.float64)
    if kid_subset_size > len(f1) or kid_subset_size > len(f2):
        raise ValueError(f"kid_subset_size {kid_subset_size} exceeds the number of samples ({len(f1)}, {len(f2)})")
    rng = np.random.RandomState(rng_seed)
    gam = gamma if gamma is not None else 1.0 / f1.shape[1]
    mmds = np.zeros(kid_subsets)
    for i in range(kid_subsets):
        a = f1[rng.choice(len(f1), kid_subset_size, replace=False)]
        b = f2[rng.choice(len(f2), kid_subset_size, replace=False)]
        kxx, kxy, kyy = (a @ a.T * gam + coef0) ** degree, (a @ b.T * gam + coef0) ** degree, (b @ b.T * gam + coef0) ** degree
        m = kid_subset_size
        mmds[i] = ((kxx.sum() - np.trace(kxx)) + (kyy.sum() - np.trace(kyy))) / (m * (m - 1)) - 2 * kxy.sum() / (m * m)
    return {KEY_KID_MEAN: float(np.mean(mmds)), KEY_KID_STD: float(np.std(mmds))}


def to_uint8(images) -> np.ndarray:
    """What the reference's PNG files hold: ``(images * 255).round().astype("uint8")`` of the pipeline's float NHWC output in [0, 1]
    (utils_training.py:829,915); uint8 arrays pass through."""
    a = images.detach().cpu().numpy() if torch.is_tensor(images) else np.asarray(images)
    if a.dtype == np.uint8:
        return a
    return (a * 255).round().astype("uint8")


def extract_features(net: InceptionV3Features, images, batch_size: int = 64) -> dict:
    """All three feature sets of a set of images (uint8 or float NHWC in [0, 1]), fp64 on the host."""
    u8 = to_uint8(images)
    outs = []
    for i in range(0, len(u8), batch_size):
        o = net(torch.from_numpy(np.ascontiguousarray(u8[i:i + batch_size])).to(net.device))
        outs.append({k: v.double().cpu().numpy() for k, v in o.items()})
    return {k: np.concatenate([o[k] for o in outs]) for k in outs[0]}


def calculate_metrics(net: InceptionV3Features, input1, input2=None, *, isc: bool = True, fid: bool = True, kid: bool = False,
                      kid_subset_size: int = 1000, batch_size: int = 64, input2_features: Optional[dict] = None) -> dict:
    """``torch_fidelity.calculate_metrics(input1=generated, input2=real, isc=, fid=, kid=, kid_subset_size=)`` on image arrays: IS of
    ``input1``; FID / KID between ``input1`` and ``input2``.  ``input2_features``: features of the real images extracted once (what the
    reference gets from ``cache_root`` / ``input2_cache_name``)."""
    f1 = extract_features(net, input1, batch_size)
    out = {}
    if isc:
        out.update(inception_score(f1["logits_unbiased"]))
    if fid or kid:
        f2 = input2_features if input2_features is not None else extract_features(net, input2, batch_size)
        if fid:
            out[KEY_FID] = fid_from_statistics(*fid_statistics(f1["2048"]), *fid_statistics(f2["2048"]))
        if kid:
            out.update(kernel_inception_distance(f1["2048"], f2["2048"], kid_subset_size=kid_subset_size))
    return out


def class_metrics_hook(net: InceptionV3Features, real_images_by_class: Dict[int, Sequence], results: dict, *, isc: bool = True,
                       fid: bool = True, kid: bool = False, kid_subset_size: int = 1000, batch_size: int = 64):
    """``on_class_done`` callback for :func:`phendiff_amd.eval_generation.generate_samples`: what ``_compute_log_metrics`` does after a
    class's images were generated (utils_training.py:948-1001) -- metrics of that class's generated images against its real images
    (their features cached per class, like ``input2_cache_name``), stored as ``results[f"{metric}/{class_name}"]``."""
    cache: Dict[int, dict] = {}

    def done(class_label: int, class_name: str, batches):
        gen = np.concatenate([to_uint8(b.images) for b in batches])
        if class_label not in cache:
            cache[class_label] = extract_features(net, real_images_by_class[class_label], batch_size)
        m = calculate_metrics(net, gen, isc=isc, fid=fid, kid=kid, kid_subset_size=kid_subset_size, batch_size=batch_size,
                              input2_features=cache[class_label])
        for k, v in m.items():
            results[f"{k}/{class_name}"] = v
    return done
