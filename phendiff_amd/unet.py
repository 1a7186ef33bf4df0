"""``CustomCondUNet2DModel`` for MI355X: same call surface as the reference's
``src/cond_unet_2d/cond_unet_2d.py:29-362`` (diffusers ``UNet2DModel`` + class conditioning), executed by
hand-written HIP kernels through ``libphendiff_hip.so``.

The module tree (names, shapes, construction order = diffusers 0.18.2) only *holds parameters*; no torch
operator runs in ``forward``.  ``forward`` compiles, per input shape, a static launch plan
(:class:`UNetPlan`): pre-packed MFMA weights, pre-allocated NHWC activation buffers and pre-filled C-ABI
argument structs, so one UNet evaluation is ~120 asynchronous kernel launches on the current HIP stream
with no allocation and no host sync -- capturable in a hipGraph (``phendiff_amd.img2img``).
"""
from __future__ import annotations

import ctypes as C
import json
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import _lib as L
from .packing import pack_conv_weight, upsample_phase_weights, upsample_phase_weights_stacked
from .training import mark_requires_grad_calls

# compute_dtype -> (pd_dtype, storage dtype).  "fp16": the reference's `--mixed_precision fp16` (args_parser.py:381-390; img2img_comparison.py:57):
# fp16 storage + MFMA, fp32 accumulate / statistics / softmax; since round 5 it trains too (unet_train.UNetTrainer under training.LossScaler)
_DT = {"f32": (L.PD_F32, torch.float32), "bf16": (L.PD_BF16, torch.bfloat16), "fp16": (L.PD_F16, torch.float16)}


class UNet2DOutput(SimpleNamespace):
    """``diffusers.models.unet_2d.UNet2DOutput`` stand-in: ``.sample``."""


# ---------------------------------------------------------------------------------------------------
# parameter containers (diffusers state_dict names; cond_unet_2d.py:127-242 + diffusers unet_2d_blocks)
# ---------------------------------------------------------------------------------------------------
class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, tdim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, tdim)
        self.linear_2 = nn.Linear(tdim, tdim)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, tdim, groups, eps, scale_shift=False):
        super().__init__()
        self.scale_shift = scale_shift          # resnet_time_scale_shift = "scale_shift": the projection is [scale | shift]
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(tdim, cout * (2 if scale_shift else 1))
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.in_channels, self.out_channels = cin, cout


class _Attention(nn.Module):
    def __init__(self, ch, heads, groups, eps):
        super().__init__()
        self.heads = heads
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])


class _Sampler(nn.Module):
    def __init__(self, ch, stride=1, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=padding)
        self.padding = padding


class _Block(nn.Module):
    """Down / mid / up block container: ``.resnets``, ``.attentions``, ``.downsamplers`` / ``.upsamplers``."""

    def __init__(self):
        super().__init__()


def _down_block(cin, cout, tdim, n, groups, eps, add_down, down_pad, head_dim, ss=False):
    b = _Block()
    b.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, tdim, groups, eps, ss) for i in range(n)])
    if head_dim is not None:
        b.attentions = nn.ModuleList([_Attention(cout, cout // head_dim, groups, eps) for _ in range(n)])
    b.downsamplers = nn.ModuleList([_Sampler(cout, 2, down_pad)]) if add_down else None
    return b


def _up_block(cin, prev, cout, tdim, n, groups, eps, add_up, head_dim, ss=False):
    b = _Block()
    rs = []
    for i in range(n):
        skip = cin if i == n - 1 else cout
        rin = prev if i == 0 else cout
        rs.append(_Resnet(rin + skip, cout, tdim, groups, eps, ss))
    b.resnets = nn.ModuleList(rs)
    if head_dim is not None:
        b.attentions = nn.ModuleList([_Attention(cout, cout // head_dim, groups, eps) for _ in range(n)])
    b.upsamplers = nn.ModuleList([_Sampler(cout)]) if add_up else None
    return b


def _mid_block(ch, tdim, groups, eps, head_dim, add_attention, ss=False):
    b = _Block()
    b.resnets = nn.ModuleList([_Resnet(ch, ch, tdim, groups, eps, ss) for _ in range(2)])
    b.attentions = nn.ModuleList([_Attention(ch, ch // head_dim, groups, eps) if add_attention else None])
    return b


_CONFIG_DEFAULTS = dict(
    sample_size=None, in_channels=3, out_channels=3, center_input_sample=False, time_embedding_type="positional",
    freq_shift=0, flip_sin_to_cos=True,
    down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
    up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
    block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1, downsample_padding=1,
    act_fn="silu", attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5, resnet_time_scale_shift="default",
    add_attention=True, class_embed_type=None, num_class_embeds=None)


@mark_requires_grad_calls
class CustomCondUNet2DModel(nn.Module):
    """Drop-in for ``src.cond_unet_2d.CustomCondUNet2DModel`` (constructor kwargs = its ``register_to_config``
    keys, ``cond_unet_2d.py:74-107``).  ``compute_dtype``: ``"bf16"`` (bf16 activations/weights on MFMA, fp32
    accumulate, fp32 GroupNorm statistics / softmax) or ``"f32"`` (exact-fp32 MFMA; parity mode)."""

    def __init__(self, compute_dtype: str = "bf16", **kwargs):
        super().__init__()
        cfg = dict(_CONFIG_DEFAULTS)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError(f"unexpected config keys: {sorted(unknown)}")
        cfg.update(kwargs)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        cfg["down_block_types"] = tuple(cfg["down_block_types"])
        cfg["up_block_types"] = tuple(cfg["up_block_types"])
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if len(c.down_block_types) != len(c.up_block_types):  # cond_unet_2d.py:116-119
            raise ValueError("Must provide the same number of `down_block_types` as `up_block_types`.")
        if len(c.block_out_channels) != len(c.down_block_types):  # :121-124
            raise ValueError("Must provide the same number of `block_out_channels` as `down_block_types`.")
        # what the HIP path implements (everything the shipped configs use)
        if c.time_embedding_type != "positional" or c.act_fn != "silu" or c.mid_block_scale_factor != 1:
            raise NotImplementedError("phendiff_amd: positional time embedding, silu and mid_block_scale_factor = 1 are what the HIP "
                                      "path implements (no shipped config uses the fourier projection, other activations or a scale)")
        if c.class_embed_type not in (None, "timestep", "identity"):
            raise ValueError(f"unknown class_embed_type {c.class_embed_type}")
        if c.resnet_time_scale_shift not in ("default", "scale_shift"):
            raise ValueError(f"unknown resnet_time_scale_shift {c.resnet_time_scale_shift}")
        for t in c.down_block_types + c.up_block_types:
            if t not in ("DownBlock2D", "AttnDownBlock2D", "UpBlock2D", "AttnUpBlock2D"):
                raise NotImplementedError(f"block type {t}")
        boc = list(c.block_out_channels)
        if any(ch % 32 for ch in boc):
            raise NotImplementedError("block_out_channels must be multiples of 32 on the HIP path")
        if compute_dtype not in _DT:
            raise ValueError("compute_dtype must be 'bf16', 'fp16' or 'f32'")
        self.compute_dtype = compute_dtype
        self.sample_size = c.sample_size
        tdim = boc[0] * 4
        self.time_embed_dim = tdim  # cond_unet_2d.py:111-113
        g, eps = c.norm_num_groups, c.norm_eps
        self.conv_in = nn.Conv2d(c.in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], tdim)
        if c.class_embed_type is None and c.num_class_embeds is not None:      # cond_unet_2d.py:146-153
            self.class_embedding = nn.Embedding(c.num_class_embeds, tdim)
        elif c.class_embed_type == "timestep":
            self.class_embedding = _TimestepEmbedding(boc[0], tdim)
        elif c.class_embed_type == "identity":
            self.class_embedding = nn.Identity()
        else:
            self.class_embedding = None
        ss = c.resnet_time_scale_shift == "scale_shift"
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(c.down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            hd = None
            if t == "AttnDownBlock2D":
                hd = c.attention_head_dim if c.attention_head_dim is not None else out_ch
            self.down_blocks.append(_down_block(in_ch, out_ch, tdim, c.layers_per_block, g, eps,
                                                i != len(boc) - 1, c.downsample_padding, hd, ss))
        mid_hd = c.attention_head_dim if c.attention_head_dim is not None else boc[-1]
        self.mid_block = _mid_block(boc[-1], tdim, g, eps, mid_hd, c.add_attention, ss)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, t in enumerate(c.up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            hd = None
            if t == "AttnUpBlock2D":
                hd = c.attention_head_dim if c.attention_head_dim is not None else out_ch
            self.up_blocks.append(_up_block(in_ch, prev, out_ch, tdim, c.layers_per_block + 1, g, eps,
                                            i != len(boc) - 1, hd, ss))
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], c.out_channels, 3, padding=1)
        self._plans = {}
        self._weights = None
        self._grad_weights = None
        nn.Module.requires_grad_(self, False)  # no autograd graph: gradients come from the HIP backward plan (phendiff_amd.unet_train)

    # ---- diffusers-like conveniences ------------------------------------------------------------
    @classmethod
    def load_config(cls, path):
        with open(path) as f:
            d = json.load(f)
        return {k: v for k, v in d.items() if not k.startswith("_")}

    @classmethod
    def from_config(cls, config, compute_dtype="bf16", **overrides):
        d = dict(config) if isinstance(config, dict) else dict(vars(config))
        dropped = sorted(k for k in d if k not in _CONFIG_DEFAULTS and not k.startswith("_"))
        if dropped:      # diffusers' ConfigMixin.extract_init_dict: "... were passed to X, but are not expected and will be ignored"
            import warnings
            warnings.warn(f"The config attributes {dropped} were passed to {cls.__name__}, but are not expected and will be ignored "
                          "(e.g. models_configs/denoiser/SD_2-1_config.json carries UNet2DConditionModel keys).", stacklevel=2)
        d = {k: v for k, v in d.items() if k in _CONFIG_DEFAULTS}
        d.update(overrides)
        return cls(compute_dtype=compute_dtype, **d)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, compute_dtype="bf16", **overrides):
        """diffusers folder layout (``config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``)."""
        import os
        from .checkpoint import load_unet
        return load_unet(cls, os.path.join(path, subfolder) if subfolder else path, compute_dtype, **overrides)

    def save_pretrained(self, path, safe_serialization=True):
        from .checkpoint import save_unet
        save_unet(self, path, safe_serialization)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        """Drop packed weights / plans (call after changing parameters in place)."""
        self._plans = {}
        self._weights = None
        self._grad_weights = None

    def input_grad_plan(self, B, H, W, device):
        """Forward + input-gradient-only backward plan (d loss / d sample through the UNet, no parameter gradients): what
        ``torch.autograd.grad(losses, images)`` needs in the gradient-guided transfer (utils_Img2Img.py:744-745)."""
        from .unet_train import TrainWeights, UNetTrainPlan
        # compute_dtype='fp16' (the reference's `mixed_precision: fp16` img2img runs, general_config.yaml:46): the activation gradients
        # are fp16, so the CALLER scales `dout` and un-scales `dsample` (img2img.custom_guided_generation: a static power-of-two scale
        # on the Lp loss gradient, halved and the step redone when the result is not finite) -- round 6
        key = ("input_grad", B, H, W, str(device), self.compute_dtype)
        p = self._plans.get(key)
        if p is None:
            if self._weights is None:
                self._weights = _PackedWeights(self, device)
            if getattr(self, "_grad_weights", None) is None:
                self._grad_weights = TrainWeights(self, device, self._weights.tdt)
            p = UNetTrainPlan(self, self._weights, self._grad_weights, B, H, W, device, input_grad=True)
            self._plans[key] = p
        return p

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                class_labels: Optional[torch.Tensor] = None, class_emb: Optional[torch.Tensor] = None,
                return_dict: bool = True) -> Union[UNet2DOutput, Tuple]:
        if class_labels is not None and class_emb is not None:  # cond_unet_2d.py:268-269
            raise ValueError("Cannot specify both class_labels and class_emb")
        if self.class_embedding is not None and class_labels is None and class_emb is None:  # :298-301
            raise ValueError("either class_labels or class_emb should be provided when doing class conditioning")
        if not sample.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback): move the model and inputs to 'cuda'")
        B = sample.shape[0]
        dev = sample.device
        # cond_unet_2d.py:276-287: scalar / 0-dim / (B,) timesteps broadcast to the batch
        if not torch.is_tensor(timestep):
            ts = torch.full((B,), float(timestep), dtype=torch.float32, device=dev)
        else:
            ts = timestep.to(device=dev, dtype=torch.float32).reshape(-1)
            if ts.numel() == 1:
                ts = ts.expand(B)
            ts = ts.contiguous()
        mb = self.max_batch(sample.shape[2], sample.shape[3])
        if B > mb:      # the kernels address one tensor with 32-bit byte offsets (< 2 GiB): larger batches run in even slices
            outs = []
            n_sl = -(-B // mb)
            step = -(-B // n_sl)
            for b0 in range(0, B, step):
                sl = slice(b0, min(B, b0 + step))
                outs.append(self.forward(sample[sl], ts[sl], class_labels[sl] if class_labels is not None else None,
                                         class_emb[sl] if class_emb is not None else None, return_dict=False)[0])
            out = torch.cat(outs, 0)
            return UNet2DOutput(sample=out) if return_dict else (out,)
        plan = self.plan_for(B, sample.shape[2], sample.shape[3], dev)
        x = sample.contiguous().to(torch.float32)
        labels = None
        if class_labels is not None:      # nn.Embedding indices, or (class_embed_type timestep / identity) float values / rows
            labels = class_labels.to(device=dev, dtype=torch.int64 if self.config.class_embed_type is None else torch.float32).contiguous()
        cemb = class_emb.to(device=dev, dtype=torch.float32).contiguous() if class_emb is not None else None
        stream = torch.cuda.current_stream(dev).cuda_stream
        temb = plan.temb_rows(ts, labels, cemb, stream)
        out = torch.empty_like(x)
        plan.run(x.data_ptr(), temb.data_ptr(), out.data_ptr(), stream)
        plan.keepalive = (x, ts, labels, cemb)
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)

    def max_batch(self, H, W) -> int:
        """Largest batch one launch plan can hold: its widest materialised activation at full resolution (the last upsampler's
        output, block_out_channels[1] channels; channel concats are never materialised; the fp32 NCHW input / output) must stay
        below 2 GiB."""
        esz = 4 if self.compute_dtype == "f32" else 2
        boc = self.config.block_out_channels
        per_image = H * W * max(max(boc[0], boc[min(1, len(boc) - 1)]) * esz, 4 * max(self.config.in_channels, self.config.out_channels))
        return max(1, (2 ** 31 - 2 ** 20) // per_image)

    def new_plan(self, B, H, W, device):
        """A plan with its OWN activation buffers (for trajectories replayed concurrently on different streams)."""
        if self._weights is None:
            self._weights = _PackedWeights(self, device)
        return UNetPlan(self, self._weights, B, H, W, device)

    def plan_for(self, B, H, W, device):
        key = (B, H, W, str(device), self.compute_dtype)
        p = self._plans.get(key)
        if p is None:
            if self._weights is None:
                self._weights = _PackedWeights(self, device)
            p = UNetPlan(self, self._weights, B, H, W, device)
            self._plans[key] = p
        return p


# ---------------------------------------------------------------------------------------------------
class _PackedWeights:
    """Weights in kernel layouts (device).  Built once per model / device / dtype."""

    def __init__(self, m: CustomCondUNet2DModel, device):
        self.code, self.tdt = _DT[m.compute_dtype]
        dev = device
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        self.conv_in_w, self.conv_in_b = f32(m.conv_in.weight), f32(m.conv_in.bias)
        cin = m.conv_in.weight.shape[1]
        if cin > 3:
            raise NotImplementedError("conv_in on the HIP path takes <= 3 input channels (pixel-space UNet)")
        # conv_in as a 1x1 conv over 32 virtual channels k = ci*9 + ky*3 + kx (pd_conv im2col3 mode)
        wv = torch.zeros((m.conv_in.weight.shape[0], 32, 1, 1), dtype=torch.float32, device=dev)
        wv[:, :cin * 9, 0, 0] = f32(m.conv_in.weight).reshape(-1, cin * 9)
        self.conv_in_wv = pack_conv_weight(wv, self.tdt)
        te = m.time_embedding
        self.w1T, self.b1 = f32(te.linear_1.weight.t()), f32(te.linear_1.bias)
        self.w2T, self.b2 = f32(te.linear_2.weight.t()), f32(te.linear_2.bias)
        self.class_table = f32(m.class_embedding.weight) if isinstance(m.class_embedding, nn.Embedding) else None
        # class_embed_type: "timestep" = a second TimestepEmbedding over the sinusoid of the labels; "identity" = rows given as is
        self.class_mode = m.config.class_embed_type
        if self.class_mode == "timestep":
            ce = m.class_embedding
            self.cw1T, self.cb1 = f32(ce.linear_1.weight.t()), f32(ce.linear_1.bias)
            self.cw2T, self.cb2 = f32(ce.linear_2.weight.t()), f32(ce.linear_2.bias)
        self.resnets = {}
        self.attns = {}
        self.samplers = {}
        proj_w, proj_b, off = [], [], 0
        for name, r in self._iter(m, _Resnet):
            e = SimpleNamespace()
            e.cin, e.cout = r.in_channels, r.out_channels
            e.g1, e.be1 = f32(r.norm1.weight), f32(r.norm1.bias)
            e.g2, e.be2 = f32(r.norm2.weight), f32(r.norm2.bias)
            e.w1, e.b1 = self._pack(r.conv1.weight), f32(r.conv1.bias)
            e.w2, e.b2 = self._pack(r.conv2.weight), f32(r.conv2.bias)
            if r.conv_shortcut is not None:
                # conv_shortcut is folded into conv2 (pd_conv tail): per 32-co tile the tail's fragments follow conv2's
                ws = self._pack(r.conv_shortcut.weight)
                ct = e.w2.shape[0]
                e.w2 = torch.cat([e.w2.reshape(ct, -1, 64, 8), ws.reshape(ct, -1, 64, 8)], 1).contiguous()
                e.b2 = e.b2 + f32(r.conv_shortcut.bias)
                e.fused_shortcut = True
            else:
                e.fused_shortcut = False
            e.eps = r.norm1.eps
            e.temb_off = off
            e.scale_shift = r.scale_shift
            off += r.time_emb_proj.weight.shape[0]
            proj_w.append(r.time_emb_proj.weight.detach())
            proj_b.append(r.time_emb_proj.bias.detach())
            self.resnets[name] = e
        self.proj_dim = off
        self.wpT = f32(torch.cat(proj_w, 0).t())
        self.bp = f32(torch.cat(proj_b, 0))
        for name, a in self._iter(m, _Attention):
            e = SimpleNamespace()
            e.heads = a.heads
            e.g, e.be, e.eps = f32(a.group_norm.weight), f32(a.group_norm.bias), a.group_norm.eps
            wqkv = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0).detach()
            e.wqkv = self._pack(wqkv[:, :, None, None])
            e.bqkv = f32(torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias], 0))
            e.wo, e.bo = self._pack(a.to_out[0].weight.detach()[:, :, None, None]), f32(a.to_out[0].bias)
            self.attns[name] = e
        for name, s in self._iter(m, _Sampler):
            e = SimpleNamespace()
            e.w, e.b, e.padding = self._pack(s.conv.weight), f32(s.conv.bias), s.padding
            if ".upsamplers." in name:
                # Upsample2D as four 2x2 convolutions over the low-resolution tensor (pd_conv phase 1..4): 4 / 9 of the FLOPs of the 3x3
                # convolution over the nearest-upsampled tensor (round 4: the training plans too -- their input gradient runs the four phases backwards, pd_conv phase_in; the weight gradient stays on the 3x3 form)
                # (w4_src: the four fp32 phase kernels stacked, on the device -- the training re-pack refreshes it and re-packs w4 every step)
                e.w4_src = upsample_phase_weights_stacked(s.conv.weight.detach().to(device=device, dtype=torch.float32))
                e.w4 = tuple(self._pack(e.w4_src[p]) for p in range(4))      # (a tuple: _copy_into refreshes tuples of tensors in place)
            self.samplers[name] = e
        self.gn_out = (f32(m.conv_norm_out.weight), f32(m.conv_norm_out.bias), m.conv_norm_out.eps)
        co = m.conv_out.weight.shape[0]
        self.conv_out_pad = ((co + 31) // 32) * 32
        self.conv_out_w = self._pack(m.conv_out.weight, self.conv_out_pad)
        b = torch.zeros(self.conv_out_pad, dtype=torch.float32, device=dev)
        b[:co] = f32(m.conv_out.bias)
        self.conv_out_b = b

    @staticmethod
    def _iter(m, cls):
        for name, mod in m.named_modules():
            if isinstance(mod, cls):
                yield name, mod

    def _pack(self, w, cout_pad=None):
        # packed on the device the plan runs on (a training step re-packs after every optimizer update)
        return pack_conv_weight(w.detach().to(device=self.conv_in_w.device, dtype=torch.float32), self.tdt, cout_pad)

    def refresh(self, m):
        """Re-derive every kernel-layout tensor from ``m``'s current parameters IN PLACE (device pointers held by launch
        plans stay valid).  Called after each optimizer step of a training run."""
        _copy_into(self, type(self)(m, self.conv_in_w.device))


def _copy_into(dst, src):
    items = src.items() if isinstance(src, dict) else vars(src).items()
    for k, v in items:
        d = dst[k] if isinstance(dst, dict) else getattr(dst, k)
        if torch.is_tensor(v):
            d.copy_(v)
        elif isinstance(v, (dict, SimpleNamespace)):
            _copy_into(d, v)
        elif isinstance(v, tuple):
            for dd, vv in zip(d, v):
                if torch.is_tensor(vv):
                    dd.copy_(vv)


class _Op:
    """One kernel launch of the plan; ``flops`` / ``bytes`` are its ALGORITHMIC work (logical shapes, each tensor
    read or written once), used by bench.py's roofline leg."""
    __slots__ = ("fn", "args", "what", "flops", "bytes", "ctx", "side")

    def __init__(self, fn, args, what, flops=0.0, nbytes=0.0):
        self.fn, self.args, self.what, self.flops, self.bytes = fn, args, what, float(flops), float(nbytes)
        self.ctx = False        # True: depends on the conditioning context only (the SD plan's cross-attention k / v projections)
        self.side = False       # True (training plans): the fold of a weight-gradient slab -- may run on the plan's second stream


class UNetPlan:
    """Static launch plan of one UNet forward for a fixed (B, H, W)."""

    def __init__(self, m: CustomCondUNet2DModel, w: _PackedWeights, B, H, W, device):
        self.lib = L.lib()
        self.m, self.w = m, w
        self.B, self.H, self.W, self.device = B, H, W, device
        self.code, self.tdt = w.code, w.tdt
        c = m.config
        nlev = len(c.block_out_channels)
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise ValueError(f"sample size {(H, W)} must be a multiple of {1 << (nlev - 1)}")
        self.train = bool(getattr(self, "train", False))   # set by UNetTrainPlan before the forward is laid out
        self.tape = []          # block-level records of the forward (what a backward plan walks in reverse)
        self.gn_saved = {}      # id(scale) -> GroupNorm inputs / statistics kept for the backward (train plans only)
        self.ops = []
        self.bufs = []          # keep every device buffer alive
        self.stats = {}         # id(NHWC activation) -> (per-tile channel sums [B][T][C][2], T) written by its producer
        self._kmax_arena, self._kmax_used = None, 0   # max |k|^2 per (sample, head) of every d = 8 attention (zeroed once per forward)
        self.groups = c.norm_num_groups
        self.temb_args = None
        self._temb_ptr_fields = []
        self._in_args = None
        self._out_args = None
        self._build()

    # ---- buffers ---------------------------------------------------------------------------------
    def _act(self, h, w, ch):
        t = torch.empty((self.B, h, w, ch), dtype=self.tdt, device=self.device)
        self.bufs.append(t)
        return t

    def _f32(self, *shape):
        t = torch.empty(shape, dtype=torch.float32, device=self.device)
        self.bufs.append(t)
        return t

    # ---- op emitters -----------------------------------------------------------------------------
    def _gn(self, x0, x1, gamma, beta, eps, temb_off=None):
        """GroupNorm(32) of [x0 | x1] -> per-(sample, channel) scale/shift, from the statistics the producers of x0 / x1
        emitted in their epilogues (no pass over the tensors)."""
        B, h, w, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        C_ = c0 + c1
        st0, t0 = self.stats[id(x0)]
        st1, t1 = self.stats[id(x1)] if x1 is not None else (None, 0)
        scale, shift = self._f32(self.B, C_), self._f32(self.B, C_)
        a = L.GnFinalizeArgs(B=self.B, HW=h * w, groups=self.groups, eps=eps, C0=c0, T0=t0, stats0=st0.data_ptr(),
                             C1=c1, T1=t1, stats1=L.ptr(st1), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
                             scale=scale.data_ptr(), shift=shift.data_ptr())
        if temb_off is not None:      # scale_shift ResNets: this norm's affine is modulated by the step's [scale | shift] row
            a.temb_stride = self.w.proj_dim
            self._temb_ptr_fields.append((a, temb_off))
        if self.train:
            mean, rstd = self._f32(self.B, self.groups), self._f32(self.B, self.groups)
            a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
            self.gn_saved[id(scale)] = SimpleNamespace(mean=mean, rstd=rstd, gamma=gamma, beta=beta, x0=x0, x1=x1)
        self.ops.append(_Op(self.lib.pd_gn_finalize, a, "gn_finalize", 0.0, self.B * (t0 * c0 + t1 * c1) * 8.0))
        return scale, shift

    def _conv(self, x0, x1, wpk, bias, cout, *, ksize=3, stride=1, pad=1, upsample=0, silu=0, gn=None, temb_off=None,
              residual=None, out_mode=L.PD_OUT_NHWC, heads=0, cout_pad=None, y=None, stats=True, im2col3=0, src_ptr=None,
              src_shape=None, tail=None):
        B, hin, win, c0 = src_shape if src_shape is not None else x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        hc, wc = (2 * hin, 2 * win) if upsample else (hin, win)
        extra = 1 if (ksize == 3 and pad == 0) else 0
        hout = (hc + 2 * pad + extra - ksize) // stride + 1
        wout = (wc + 2 * pad + extra - ksize) // stride + 1
        cout_pad = cout_pad or cout
        if y is None:
            if out_mode == L.PD_OUT_NHWC:
                y = self._act(hout, wout, cout)
            elif out_mode == L.PD_OUT_QKV_HEADS:
                y = torch.empty((3, B, heads, hout * wout, 8), dtype=self.tdt, device=self.device)
                self.bufs.append(y)
        st = None
        if stats and out_mode == L.PD_OUT_NHWC:
            T = self.lib.pd_conv_stat_tiles(hout, wout, ksize, stride)
            st = self._f32(B, T, cout, 2)
            self.stats[id(y)] = (st, T)
        a = L.ConvArgs(dtype=self.code, B=B, Hin=hin, Win=win, Hout=hout, Wout=wout, C0=c0, C1=c1, Cout=cout,
                       Cout_pad=cout_pad, ksize=ksize, stride=stride, pad=pad, upsample=upsample, silu=silu,
                       out_mode=out_mode, heads=heads, x0=(x0.data_ptr() if x0 is not None else src_ptr), x1=L.ptr(x1),
                       scale=L.ptr(gn[0]) if gn else None, shift=L.ptr(gn[1]) if gn else None,
                       w_packed=wpk.data_ptr(), bias=bias.data_ptr(), temb=None, temb_stride=self.w.proj_dim,
                       residual=L.ptr(residual), y=L.ptr(y), stats_out=L.ptr(st), im2col3=im2col3,
                       tail_x0=L.ptr(tail[0]) if tail else None, tail_x1=L.ptr(tail[1]) if tail else None,
                       tail_C0=tail[0].shape[3] if tail else 0,
                       tail_C1=(tail[1].shape[3] if (tail and tail[1] is not None) else 0))
        if temb_off is not None:
            self._temb_ptr_fields.append((a, temb_off))
        esz = 4 if self.code == L.PD_F32 else 2
        cin = c0 + c1
        flops = 2.0 * B * hout * wout * cout * cin * ksize * ksize
        tail_c = (tail[0].shape[3] + (tail[1].shape[3] if tail[1] is not None else 0)) if tail else 0
        flops += 2.0 * B * hout * wout * cout * tail_c
        nbytes = (B * hin * win * cin + B * hout * wout * cout * (2 if residual is not None else 1)) * esz \
            + cout * cin * ksize * ksize * esz + (B * hout * wout * tail_c + cout * tail_c) * esz
        if out_mode == L.PD_OUT_NCHW_F32:
            nbytes += B * hout * wout * cout * (4 - esz)
        self.ops.append(_Op(self.lib.pd_conv, a, f"conv{ksize}x{ksize}", flops, nbytes))
        return y, a

    @property
    def SUBPIXEL_UP(self):      # diagnostic (read when a plan is built): same-box A/B against the 3x3-over-upsampled form
        return __import__("os").environ.get("PD_SUBPIXEL_UP", "1") != "0"


    def _subpixel_up_ok(self, x):
        B, h, w, ch = x.shape
        esz = 4 if self.code == L.PD_F32 else 2
        return (self.SUBPIXEL_UP and (not getattr(self, "train", False) or getattr(self, "subpixel_in_training", False)) and w >= 16 and ch % 32 == 0
                and B * 4 * h * w * ch * esz < (1 << 31))

    def _upconv_subpixel(self, x, s):
        """Upsample2D (nearest x2 + conv 3x3 pad 1; diffusers resnet.py, reached from cond_unet_2d.py:200-228) as four 2x2 convolutions
        over the low-resolution tensor, one per output phase (``pd_conv_args.phase``): each launch writes every other pixel of every other
        row of the upsampled output and its share of the output's GroupNorm statistic tiles."""
        B, h, w, ch = x.shape
        y = self._act(2 * h, 2 * w, ch)
        T = self.lib.pd_conv_stat_tiles(h, w, 2, 1)
        st = self._f32(B, 4 * T, ch, 2)
        self.stats[id(y)] = (st, 4 * T)
        esz = 4 if self.code == L.PD_F32 else 2
        for ph in range(4):
            a = L.ConvArgs(dtype=self.code, B=B, Hin=h, Win=w, Hout=h, Wout=w, C0=ch, C1=0, Cout=ch, Cout_pad=ch, ksize=2, stride=1, pad=0,
                           upsample=0, silu=0, out_mode=L.PD_OUT_NHWC, heads=0, x0=x.data_ptr(), x1=None, scale=None, shift=None,
                           w_packed=s.w4[ph].data_ptr(), bias=s.b.data_ptr(), temb=None, temb_stride=self.w.proj_dim, residual=None,
                           y=y.data_ptr(), stats_out=st.data_ptr(), im2col3=0, tail_x0=None, tail_x1=None, tail_C0=0, tail_C1=0, phase=1 + ph)
            # FLOPs / bytes of the LOGICAL layer (3x3 over the upsampled tensor: what the roofline accounting quotes) shared by the four launches
            self.ops.append(_Op(self.lib.pd_conv, a, "conv3x3", 2.0 * B * 4 * h * w * ch * ch * 9 / 4.0,
                                (B * h * w * ch + B * 4 * h * w * ch) * esz / 4.0 + ch * ch * 4 * esz))
        return y

    # pd_conv applies GroupNorm + SiLU while staging, once per 64-channel output tile; from this many output channels on
    # the input is normalised ONCE by pd_gn_apply instead and the convolution (and its weight gradient) runs without a prologue
    PREAPPLY_MIN_COUT = int(__import__("os").environ.get("PD_PREAPPLY_MIN_COUT", 320))      # measured: a win from 5 output tiles on (SD UNet: 7.7 vs 9.4 ms of 3x3 convs per forward), neutral at 4

    def _gn_apply(self, x0, x1, gn, silu):
        B, h, w, c0 = x0.shape
        c1 = x1.shape[3] if x1 is not None else 0
        y = self._act(h, w, c0 + c1)
        a = L.GnApplyArgs(dtype=self.code, B=B, HW=h * w, C0=c0, C1=c1, silu=silu, x0=x0.data_ptr(), x1=L.ptr(x1),
                          scale=gn[0].data_ptr(), shift=gn[1].data_ptr(), y=y.data_ptr())
        self.ops.append(_Op(self.lib.pd_gn_apply, a, "gn_apply", 0.0, 2.0 * y.numel() * (4 if self.code == L.PD_F32 else 2)))
        return y

    def _linear(self, x, wpk, bias, cout, residual=None, y=None, gn=None, stats=False, what="linear", glu=False):
        """nn.Linear over the tokens of an NHWC tensor through the GEMM kernel (``pd_linear``); ``wpk`` is pd_conv's packed 1x1
        layout, so forward and input-gradient weights are shared with the convolution path.  ``gn``: GroupNorm apply fused into
        the staging; ``stats``: emit the output's per-tile GroupNorm statistics.  Both need tokens-per-sample % 128 == 0
        (:meth:`_linear_ok`).  ``glu``: fused GEGLU epilogue (``wpk`` with value / gate tiles interleaved) -> ``cout // 2`` channels."""
        B, h, w, K = x.shape
        if y is None:
            y = self._act(h, w, cout // 2 if glu else cout)
        M, esz = B * h * w, (4 if self.code == L.PD_F32 else 2)
        st = None
        if stats:
            T = (h * w) // 128
            st = self._f32(B, T, cout, 2)
            self.stats[id(y)] = (st, T)
        a = L.LinearArgs(dtype=self.code, M=M, K=K, N=cout, N_pad=((cout + 31) // 32) * 32, x=x.data_ptr(), x_stride=K,
                         w_packed=wpk.data_ptr(), bias=bias.data_ptr(), residual=L.ptr(residual), y=y.data_ptr(),
                         scale=L.ptr(gn[0]) if gn else None, shift=L.ptr(gn[1]) if gn else None, rows_per_sample=h * w, qkv_heads=0,
                         stats_out=L.ptr(st), glu=int(glu))
        self.ops.append(_Op(self.lib.pd_linear, a, what, 2.0 * M * K * cout,
                            (M * K + M * (cout // 2 if glu else cout) * (2 if residual is not None else 1) + K * cout) * esz))
        return y

    @staticmethod
    def _linear_ok(x):
        """pd_linear's fused GroupNorm prologue / statistics epilogue work on whole 128-token tiles of one sample."""
        return (x.shape[1] * x.shape[2]) % 128 == 0 and x.shape[3] % 64 == 0

    def _resnet(self, name, x0, x1=None):
        e = self.w.resnets[name]
        pre = e.cout >= self.PREAPPLY_MIN_COUT
        z1 = z2 = None
        ss = getattr(e, "scale_shift", False)       # the latent-diffusion plan's entries have no such switch
        # "default": the projected embedding is added by conv1's epilogue; "scale_shift": it modulates norm2's affine instead
        t1 = None if ss else e.temb_off
        gn1 = self._gn(x0, x1, e.g1, e.be1, e.eps)
        if pre:
            z1 = self._gn_apply(x0, x1, gn1, 1)
            h1, _ = self._conv(z1, None, e.w1, e.b1, e.cout, temb_off=t1)
        else:
            h1, _ = self._conv(x0, x1, e.w1, e.b1, e.cout, silu=1, gn=gn1, temb_off=t1)
        gn2 = self._gn(h1, None, e.g2, e.be2, e.eps, temb_off=e.temb_off if ss else None)
        if pre:
            z2 = self._gn_apply(h1, None, gn2, 1)
        src, kw = (z2, dict()) if pre else (h1, dict(silu=1, gn=gn2))
        if e.fused_shortcut:
            out, _ = self._conv(src, None, e.w2, e.b2, e.cout, tail=(x0, x1), **kw)
        else:
            assert x1 is None
            out, _ = self._conv(src, None, e.w2, e.b2, e.cout, residual=x0, **kw)
        # z1 / z2: the normalised conv inputs when they were materialised (the weight gradients read them instead of
        # rebuilding GroupNorm + SiLU per tile)
        self.tape.append(SimpleNamespace(kind="resnet", name=name, x0=x0, x1=x1, h1=h1, out=out, gn1=gn1, gn2=gn2, e=e, z1=z1, z2=z2))
        return out

    def _kmax_slot(self, n):
        """Device address of ``n`` fp32 slots for a q/k/v projection's max |k|^2 (``pd_linear`` kmax2_out -> ``pd_attn_d8``
        kmax2).  The slots of one forward share an arena that a single ``pd_zero`` launch, placed before its first user,
        resets."""
        ARENA = 1 << 15
        if n > ARENA:
            return None
        if self._kmax_arena is None or self._kmax_used + n > ARENA:
            self._kmax_arena = torch.zeros(ARENA, dtype=torch.float32, device=self.device)
            self._kmax_used = 0
            self.bufs.append(self._kmax_arena)
            self.ops.append(_Op(self.lib.pd_zero, L.ZeroArgs(ptr=self._kmax_arena.data_ptr(), bytes=ARENA * 4), "zero", 0.0, ARENA * 4))
        ptr = self._kmax_arena.data_ptr() + 4 * self._kmax_used
        self._kmax_used += n
        return ptr

    def _attn(self, name, x):
        e = self.w.attns[name]
        B, h, w, ch = x.shape
        if ch != e.heads * 8:
            # attention_head_dim = None (one head over all channels: orig_google_ddpm_model_denoiser.json) or 64
            return self._attn_nhwc(name, x)
        gn = self._gn(x, None, e.g, e.be, e.eps)
        if (h * w) % 128 == 0 and ch % 64 == 0:
            # fused q/k/v projection through the GEMM kernel (GroupNorm apply while staging, head-major output): a 1x1 pd_conv
            # stages 32-channel chunks with a barrier per 8 MFMAs and runs at about half its rate
            qkv = torch.empty((3, B, e.heads, h * w, 8), dtype=self.tdt, device=self.device)
            self.bufs.append(qkv)
            M = B * h * w
            kmax2 = self._kmax_slot(B * e.heads) if self.code != L.PD_F32 else None
            a = L.LinearArgs(dtype=self.code, M=M, K=ch, N=3 * ch, N_pad=3 * ch, x=x.data_ptr(), x_stride=ch, w_packed=e.wqkv.data_ptr(),
                             bias=e.bqkv.data_ptr(), residual=None, y=qkv.data_ptr(), scale=gn[0].data_ptr(), shift=gn[1].data_ptr(),
                             rows_per_sample=h * w, qkv_heads=e.heads, kmax2_out=kmax2)
            # 16-bit engines, OPT-IN (PD_LIN_FOLD=1): the GroupNorm affine folded into per-sample weights, the projection through the
            # DMA-staged GEMM (pd_linear's `fold_ws` route); the attentions of one forward run one after the other and share the
            # workspace.  Round 5's one-process 2x2x2 factorial on the whole headline workload (profiles/r5_ab_factorial.log): the route
            # is faster per layer (0.106 -> 0.08 ms) and SLOWER per trajectory -- 15.844 images/s with it, 16.043 without (-1.2 %, spread
            # +-0.1 %; neutral on configs[1]) -- the denser MFMA issue costs the attention next to it more clock than the layer saves.
            # Decisions are taken on the whole workload: the staged route is the default.
            need = int(self.lib.pd_linear_fold_workspace(C.byref(a))) if __import__("os").environ.get("PD_LIN_FOLD", "0") != "0" else 0
            if need > 0:
                ws = getattr(self, "_fold_ws", None)
                if ws is None or ws.numel() < need:
                    ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                    self._fold_ws = ws
                    self.bufs.append(ws)
                a.fold_ws, a.fold_ws_bytes = ws.data_ptr(), need
            esz_ = 4 if self.code == L.PD_F32 else 2
            self.ops.append(_Op(self.lib.pd_linear, a, "conv1x1", 2.0 * M * ch * 3 * ch, (M * ch * 4 + 3 * ch * ch) * esz_))
        else:
            kmax2 = None
            qkv, _ = self._conv(x, None, e.wqkv, e.bqkv, 3 * ch, ksize=1, pad=0, gn=gn, out_mode=L.PD_OUT_QKV_HEADS,
                                heads=e.heads)
        o = self._act(h, w, ch)
        lse = self._f32(B, e.heads, h * w) if self.train else None
        a = L.AttnArgs(dtype=self.code, B=B, heads=e.heads, N=h * w, q=qkv[0].data_ptr(), k=qkv[1].data_ptr(),
                       v=qkv[2].data_ptr(), out=o.data_ptr(), lse=L.ptr(lse), kmax2=kmax2)
        esz = 4 if self.code == L.PD_F32 else 2
        N = h * w
        self.ops.append(_Op(self.lib.pd_attn_d8, a, "attn_d8", 4.0 * B * e.heads * N * N * 8, 4.0 * B * N * ch * esz))
        if self._linear_ok(o):
            out = self._linear(o, e.wo, e.bo, ch, residual=x, stats=True, what="conv1x1")
        else:
            out, _ = self._conv(o, None, e.wo, e.bo, ch, ksize=1, pad=0, residual=x)
        self.tape.append(SimpleNamespace(kind="attn", name=name, x=x, qkv=qkv, o=o, out=out, lse=lse, gn=gn, e=e))
        return out

    def _attn_nhwc(self, name, x):
        """GroupNorm -> fused q|k|v Linear (NHWC, no head-major copy) -> attention with head_dim 64 (``pd_attn_d64``) or one wide
        head of 128 / 256 / 512 channels (``pd_attn_wide``) -> out Linear + residual: diffusers ``Attention`` with
        ``residual_connection=True`` for every head_dim other than 8 (the VAE mid block; ``attention_head_dim = None``)."""
        e = self.w.attns[name]
        B, h, w, ch = x.shape
        N, esz = h * w, (4 if self.code == L.PD_F32 else 2)
        d = ch // e.heads
        if d != 64 and d not in (128, 256, 512):
            raise NotImplementedError(f"attention head_dim {d}: implemented are 8, 64 and one wide head of 128 / 256 / 512 channels")
        gn = self._gn(x, None, e.g, e.be, e.eps)
        qkv, _ = self._conv(x, None, e.wqkv, e.bqkv, 3 * ch, ksize=1, pad=0, gn=gn, stats=False)
        o = self._act(h, w, ch)
        p = qkv.data_ptr()
        lse = self._f32(B, e.heads, N) if self.train else None        # kept for the backward (pd_attn_d64_bwd / pd_attn_wide_bwd)
        if d == 64:
            a = L.AttnD64Args(dtype=self.code, B=B, heads=e.heads, Nq=N, Nkv=N, q=p, q_stride=3 * ch, k=p + ch * esz,
                              v=p + 2 * ch * esz, kv_stride=3 * ch, out=o.data_ptr(), out_stride=ch, lse=L.ptr(lse))
            fn, what = self.lib.pd_attn_d64, "attn_d64"
        else:
            a = L.AttnWideArgs(dtype=self.code, B=B, heads=e.heads, D=d, Nq=N, Nkv=N, scale=float(d) ** -0.5, q=p, q_stride=3 * ch,
                               k=p + ch * esz, v=p + 2 * ch * esz, kv_stride=3 * ch, out=o.data_ptr(), out_stride=ch, lse=L.ptr(lse))
            fn, what = self.lib.pd_attn_wide, "attn_wide"
        self.ops.append(_Op(fn, a, what, 4.0 * B * N * N * ch, 4.0 * B * N * ch * esz))
        if self._linear_ok(o):
            out = self._linear(o, e.wo, e.bo, ch, residual=x, stats=True, what="conv1x1")
        else:
            out, _ = self._conv(o, None, e.wo, e.bo, ch, ksize=1, pad=0, residual=x)
        # recorded on inference plans too (lse = None), like every other block: diagnostics.assert_finite_activations walks the tape,
        # and these are the buffers most likely to overflow in fp16
        self.tape.append(SimpleNamespace(kind="attn_nhwc", name=name, x=x, qkv=qkv, o=o, out=out, lse=lse, gn=gn, e=e, d=d))
        return out

    def _build(self):
        m, w, c = self.m, self.w, self.m.config
        boc = c.block_out_channels
        B, H, W = self.B, self.H, self.W
        # temb (filled per call)
        self.temb_args = L.TembArgs(rows=B, c0=boc[0], tdim=m.time_embed_dim, proj_dim=w.proj_dim,
                                    flip_sin_to_cos=int(c.flip_sin_to_cos), freq_shift=float(c.freq_shift),
                                    num_classes=(c.num_class_embeds or 0), w1=w.w1T.data_ptr(), b1=w.b1.data_ptr(),
                                    w2=w.w2T.data_ptr(), b2=w.b2.data_ptr(), class_table=L.ptr(w.class_table),
                                    wp=w.wpT.data_ptr(), bp=w.bp.data_ptr())
        # conv_in (cond_unet_2d.py:127-129,313): NCHW fp32 sample -> NHWC, MFMA 1x1 conv over 32 im2col channels
        self._in_field = "x0"
        centered = None
        if c.center_input_sample:      # cond_unet_2d.py:272-273: sample = 2 * sample - 1.0 (before conv_in's zero padding)
            centered = self._f32(B, c.in_channels, H, W)
            self._centered = centered      # (training: conv_in's weight gradient reads THIS tensor; the input gradient is doubled)
            self._center_const = (torch.ones_like(centered), torch.full((B,), 2.0, device=self.device), torch.full((B,), -1.0, device=self.device))
            ones, two, neg = self._center_const
            ca = L.AddNoiseArgs(numel=centered.numel(), per_sample=centered[0].numel(), velocity=0, x=None, noise=ones.data_ptr(),
                                sa=two.data_ptr(), sb=neg.data_ptr(), out=centered.data_ptr())
            self.ops.append(_Op(self.lib.pd_add_noise, ca, "center", 0.0, 2.0 * centered.numel() * 4))
        a0, conv_in_args = self._conv(None, None, w.conv_in_wv, w.conv_in_b, boc[0], ksize=1, pad=0,
                                      im2col3=c.in_channels, src_shape=(B, H, W, 32))
        if centered is not None:
            conv_in_args.x0 = centered.data_ptr()
            self._in_args, self._in_field = ca, "x"
        else:
            self._in_args = conv_in_args
        self.ops[-1].what = "conv_in"
        self.ops[-1].flops = 2.0 * B * H * W * boc[0] * c.in_channels * 9
        self.ops[-1].bytes = B * H * W * (c.in_channels * 4 + boc[0] * (4 if self.code == L.PD_F32 else 2))
        self.tape.append(SimpleNamespace(kind="conv_in", out=a0))
        h = a0
        skips = [a0]
        for i, blk in enumerate(m.down_blocks):
            has_attn = hasattr(blk, "attentions")
            for j in range(len(blk.resnets)):
                h = self._resnet(f"down_blocks.{i}.resnets.{j}", h)
                if has_attn:
                    h = self._attn(f"down_blocks.{i}.attentions.{j}", h)
                skips.append(h)
            if blk.downsamplers is not None:
                s = w.samplers[f"down_blocks.{i}.downsamplers.0"]
                hd, _ = self._conv(h, None, s.w, s.b, h.shape[3], stride=2, pad=s.padding)
                self.tape.append(SimpleNamespace(kind="down", name=f"down_blocks.{i}.downsamplers.0", x=h, out=hd, e=s))
                h = hd
                skips.append(h)
        h = self._resnet("mid_block.resnets.0", h)
        if m.mid_block.attentions[0] is not None:
            h = self._attn("mid_block.attentions.0", h)
        h = self._resnet("mid_block.resnets.1", h)
        for i, blk in enumerate(m.up_blocks):
            has_attn = hasattr(blk, "attentions")
            for j in range(len(blk.resnets)):
                skip = skips.pop()
                h = self._resnet(f"up_blocks.{i}.resnets.{j}", h, skip)
                if has_attn:
                    h = self._attn(f"up_blocks.{i}.attentions.{j}", h)
            if blk.upsamplers is not None:
                s = w.samplers[f"up_blocks.{i}.upsamplers.0"]
                if self._subpixel_up_ok(h):
                    hu = self._upconv_subpixel(h, s)
                else:
                    hu, _ = self._conv(h, None, s.w, s.b, h.shape[3], upsample=1)
                self.tape.append(SimpleNamespace(kind="up", name=f"up_blocks.{i}.upsamplers.0", x=h, out=hu, e=s))
                h = hu
        g, be, eps = w.gn_out
        gn = self._gn(h, None, g, be, eps)
        _, self._out_args = self._conv(h, None, w.conv_out_w, w.conv_out_b, c.out_channels, silu=1, gn=gn,
                                       out_mode=L.PD_OUT_NCHW_F32, cout_pad=w.conv_out_pad, y=None)
        self.tape.append(SimpleNamespace(kind="conv_out", x=h, gn=gn))
        self._cur = (None, None, None)

    # ---- execution -------------------------------------------------------------------------------
    def temb_rows(self, ts, labels, class_emb, stream, rows=None, out=None):
        """Launch pd_temb for ``rows`` (default B) (timestep, class) rows -> [rows][proj_dim] fp32 table."""
        rows = rows or self.B
        if out is None:
            out = torch.empty((rows, self.w.proj_dim), dtype=torch.float32, device=self.device)
        a = self.temb_args
        a.rows = rows
        w = self.w
        class_mode = getattr(w, "class_mode", None)
        tdim = self.m.time_embed_dim
        if class_mode == "identity":        # class_embed_type = "identity": the "labels" ARE the embedding rows
            if class_emb is None and labels is not None:
                class_emb = labels.to(device=self.device, dtype=torch.float32).contiguous()
            labels = None
        elif class_mode == "timestep":      # labels -> sinusoid -> class_embedding (a second TimestepEmbedding), :301-305
            if class_emb is None and labels is not None:
                class_emb = self._class_rows_timestep(labels, rows, stream)
            labels = None
        elif w.class_table is None:           # no class conditioning: labels / class_emb are ignored (cond_unet_2d.py:297-309)
            labels = class_emb = None
        # pd_temb reads rows timesteps, rows labels and rows * time_embed_dim class_emb floats: refuse anything shorter
        if ts.numel() < rows:
            raise ValueError(f"temb_rows: {ts.numel()} timesteps for {rows} rows")
        if labels is not None and labels.numel() != rows:
            raise ValueError(f"temb_rows: {labels.numel()} class labels for {rows} rows")
        if class_emb is not None and class_emb.numel() != rows * tdim:
            raise ValueError(f"temb_rows: class_emb has {class_emb.numel()} elements, needs rows x time_embed_dim = {rows} x {tdim}"
                             + (" (class_embed_type='identity': pass the embedding rows, not integer labels)" if class_mode == "identity" else ""))
        self._temb_keep = (labels, class_emb)
        a.timesteps, a.labels, a.class_emb = ts.data_ptr(), L.ptr(labels), L.ptr(class_emb)
        a.emb, a.proj = None, out.data_ptr()
        L.check(self.lib.pd_temb(C.byref(a), stream), "pd_temb")
        return out

    def _class_rows_timestep(self, labels, rows, stream, emb=None, scratch=None, vals=None):
        """``class_embedding(time_proj(class_labels))`` through ``pd_temb`` with the class MLP's weights: its ``emb`` output is the
        row block the main call adds (the projections it also computes land in a scratch table).  ``emb`` / ``scratch`` / ``vals``:
        pre-allocated buffers ([rows][time_embed_dim], [rows][proj_dim], the labels as fp32 [rows]) -- a hipGraph capture must not
        allocate (img2img._ClassRows)."""
        w, c = self.w, self.m.config
        if emb is None:
            emb = torch.empty((rows, self.m.time_embed_dim), dtype=torch.float32, device=self.device)
        if scratch is None:
            scratch = torch.empty((rows, w.proj_dim), dtype=torch.float32, device=self.device)
        if vals is None:
            vals = labels.to(device=self.device, dtype=torch.float32).contiguous()
        a = L.TembArgs(rows=rows, c0=c.block_out_channels[0], tdim=self.m.time_embed_dim, proj_dim=w.proj_dim,
                       flip_sin_to_cos=int(c.flip_sin_to_cos), freq_shift=float(c.freq_shift), num_classes=0,
                       timesteps=vals.data_ptr(), labels=None, class_emb=None, w1=w.cw1T.data_ptr(), b1=w.cb1.data_ptr(),
                       w2=w.cw2T.data_ptr(), b2=w.cb2.data_ptr(), class_table=None, wp=w.wpT.data_ptr(), bp=w.bp.data_ptr(),
                       emb=emb.data_ptr(), proj=scratch.data_ptr())
        L.check(self.lib.pd_temb(C.byref(a), stream), "pd_temb")
        self._class_keep = (vals, scratch)
        return emb

    def profile(self, x_ptr, temb_ptr, out_ptr, stream, reps=3):
        """Per-kernel-kind device time of one UNet evaluation, measured with HIP events recorded on the launch
        stream between consecutive launches (``pd_event_*``).  Returns {kind: dict(ms, launches, flops, bytes)}
        averaged over ``reps`` evaluations."""
        self.run(x_ptr, temb_ptr, out_ptr, stream)  # warm (also sets the pointers)
        return self._profile_ops(self.ops, stream, reps)

    def _profile_ops(self, ops, stream, reps=3):
        lib = self.lib
        nev = len(ops) + 1
        evs = []
        for _ in range(nev):
            e = C.c_void_p()
            L.check(lib.pd_event_create(C.byref(e)), "pd_event_create")
            evs.append(e)
        # per launch the MEDIAN over the passes (round 6: a mean let one multi-millisecond hiccup of the box -- another process's burst, a clock
        # ramp -- triple a kind's figure on the driver's line); at least three passes
        reps = max(int(reps), 3)
        per_op = [[] for _ in ops]
        for _ in range(reps):
            for i, op in enumerate(ops):
                L.check(lib.pd_event_record(evs[i], stream), "pd_event_record")
                L.check(op.fn(C.byref(op.args), stream), op.what)
            L.check(lib.pd_event_record(evs[-1], stream), "pd_event_record")
            ms = C.c_float()
            for i, op in enumerate(ops):
                L.check(lib.pd_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)), "pd_event_elapsed_ms")
                per_op[i].append(ms.value)
        acc = {}
        for op, samples in zip(ops, per_op):
            d = acc.setdefault(op.what, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            d["ms"] += sorted(samples)[len(samples) // 2]
            d["launches"] += 1.0
            d["flops"] += op.flops
            d["bytes"] += op.bytes
        for e in evs:
            lib.pd_event_destroy(e)
        return acc

    def run(self, x_ptr, temb_ptr, out_ptr, stream):
        """One UNet evaluation: NCHW fp32 at ``x_ptr`` -> NCHW fp32 prediction at ``out_ptr``; ``temb_ptr`` is the
        [B][proj_dim] fp32 table of this step.  Asynchronous, allocation-free."""
        if self._cur != (x_ptr, temb_ptr, out_ptr):
            setattr(self._in_args, self._in_field, x_ptr)
            self._out_args.y = out_ptr
            for a, off in self._temb_ptr_fields:
                a.temb = temb_ptr + 4 * off
            self._cur = (x_ptr, temb_ptr, out_ptr)
        byref, check = C.byref, L.check
        for op in self.ops:
            rc = op.fn(byref(op.args), stream)
            if rc:
                check(rc, op.what)
