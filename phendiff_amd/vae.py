"""Stable-Diffusion ``AutoencoderKL`` and ``VaeImageProcessor`` for MI355X: what
``CustomStableDiffusionImg2ImgPipeline`` calls as ``vae.encode(image).latent_dist.sample(generator)``
(``custom_pipeline_stable_diffusion_img2img.py:431``), ``vae.decode(latents / scaling_factor, return_dict=False)[0]``
(``:709-711``), ``vae.config.scaling_factor`` / ``.block_out_channels`` (``:144,433``) and what ``_encode_to_latents`` /
``_decode_to_images`` wrap (``utils_Img2Img.py:827-847``).

Same engine as :mod:`phendiff_amd.unet`: the module tree only holds parameters under diffusers' ``state_dict`` names; per
(batch, size) a static launch plan runs every ResnetBlock2D (no time embedding here), GroupNorm, sampling conv and Linear
through ``pd_conv`` / ``pd_gn_finalize`` and the one-head mid-block attention through ``pd_attn_wide`` (head_dim = 128 / 256 /
512 channels) or ``pd_attn_d64``.  No torch operator runs in a plan; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib as L
from .packing import pack_conv_weight, upsample_phase_weights
from .training import mark_requires_grad_calls
from .unet import UNetPlan, _Attention, _Block, _DT, _Op, _Sampler

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                     layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512,
                     down_block_types=("DownEncoderBlock2D",) * 4, up_block_types=("UpDecoderBlock2D",) * 4, act_fn="silu")


class _VaeResnet(nn.Module):
    """``ResnetBlock2D(temb_channels=None, eps=1e-6)`` parameters."""

    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.in_channels, self.out_channels = cin, cout


def _mid(ch, groups):
    b = _Block()
    b.resnets = nn.ModuleList([_VaeResnet(ch, ch, groups), _VaeResnet(ch, ch, groups)])
    b.attentions = nn.ModuleList([_Attention(ch, 1, groups, 1e-6)])
    return b


class _Encoder(nn.Module):
    def __init__(self, cin, latent, boc, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, ch in enumerate(boc):
            prev, out = out, ch
            b = _Block()
            b.resnets = nn.ModuleList([_VaeResnet(prev if j == 0 else out, out, groups) for j in range(layers)])
            b.downsamplers = nn.ModuleList([_Sampler(out, 2, 0)]) if i != len(boc) - 1 else None
            self.down_blocks.append(b)
        self.mid_block = _mid(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent, 3, padding=1)


class _Decoder(nn.Module):
    def __init__(self, cout, latent, boc, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(latent, boc[-1], 3, padding=1)
        self.mid_block = _mid(boc[-1], groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out = rev[0]
        for i, ch in enumerate(rev):
            prev, out = out, ch
            b = _Block()
            b.resnets = nn.ModuleList([_VaeResnet(prev if j == 0 else out, out, groups) for j in range(layers + 1)])
            b.upsamplers = nn.ModuleList([_Sampler(out)]) if i != len(boc) - 1 else None
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cout, 3, padding=1)


class DiagonalGaussianDistribution:
    """``diffusers.models.vae.DiagonalGaussianDistribution`` over device-resident moments (B, 2C, h, w) fp32."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self._C = parameters.shape[1] // 2

    @property
    def mean(self):
        return self.parameters[:, :self._C]

    @property
    def logvar(self):
        return self.parameters[:, self._C:].clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def _launch(self, noise, scale):
        p = self.parameters
        B, _, h, w = p.shape
        out = torch.empty((B, self._C, h, w), dtype=torch.float32, device=p.device)
        a = L.LatentSampleArgs(B=B, C=self._C, HW=h * w, scale=float(scale), moments=p.data_ptr(), noise=L.ptr(noise),
                               out=out.data_ptr())
        L.check(L.lib().pd_latent_sample(C.byref(a), torch.cuda.current_stream(p.device).cuda_stream), "pd_latent_sample")
        return out

    def sample(self, generator=None, noise=None, scale: float = 1.0):
        """``mean + std * randn_tensor(mean.shape, generator)``; ``scale`` folds the pipeline's ``scaling_factor *`` in.
        ``noise`` (B, C, h, w) overrides the draw (parity tests feed the oracle's noise)."""
        p = self.parameters
        if noise is None:
            shape = (p.shape[0], self._C, p.shape[2], p.shape[3])
            if generator is not None and generator.device.type == "cpu":      # diffusers randn_tensor: CPU generator -> CPU draw
                noise = torch.randn(shape, generator=generator, dtype=torch.float32).to(p.device)
            else:
                noise = torch.randn(shape, generator=generator, dtype=torch.float32, device=p.device)
        noise = noise.to(device=p.device, dtype=torch.float32).contiguous()
        return self._launch(noise, scale)

    def mode(self, scale: float = 1.0):
        return self._launch(None, scale)


class AutoencoderKLOutput(SimpleNamespace):
    """``.latent_dist``"""


class DecoderOutput(SimpleNamespace):
    """``.sample``"""


@mark_requires_grad_calls
class AutoencoderKL(nn.Module):
    """Drop-in for diffusers ``AutoencoderKL`` (SD configuration: ``DownEncoderBlock2D`` / ``UpDecoderBlock2D`` stages, one
    single-head attention in each mid block).  ``compute_dtype``: "bf16" (fast) or "f32" (exact-fp32 MFMA, parity mode)."""

    def __init__(self, compute_dtype: str = "bf16", **kwargs):
        super().__init__()
        cfg = dict(SD_VAE_CONFIG)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError(f"unexpected config keys: {sorted(unknown)}")
        cfg.update(kwargs)
        boc = tuple(cfg["block_out_channels"])
        cfg.update(block_out_channels=boc, down_block_types=tuple(cfg["down_block_types"])[:len(boc)] or ("DownEncoderBlock2D",) * len(boc),
                   up_block_types=tuple(cfg["up_block_types"])[:len(boc)] or ("UpDecoderBlock2D",) * len(boc))
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if any(t != "DownEncoderBlock2D" for t in c.down_block_types) or any(t != "UpDecoderBlock2D" for t in c.up_block_types) \
                or c.act_fn != "silu":
            raise NotImplementedError("phendiff_amd: DownEncoderBlock2D / UpDecoderBlock2D / silu only")
        if any(ch % 32 for ch in boc) or boc[-1] not in (64, 128, 256, 512):
            raise NotImplementedError("block_out_channels must be multiples of 32 and the last one 64 / 128 / 256 / 512 "
                                      "(pd_attn_d64 / pd_attn_wide)")
        if c.in_channels > 3 or c.latent_channels > 16:
            raise NotImplementedError("in_channels <= 3 (im2col conv_in), latent_channels <= 16")
        if compute_dtype not in _DT:
            raise ValueError("compute_dtype must be 'bf16', 'fp16' or 'f32'")
        self.compute_dtype = compute_dtype
        g = c.norm_num_groups
        self.encoder = _Encoder(c.in_channels, c.latent_channels, boc, c.layers_per_block, g)
        self.decoder = _Decoder(c.out_channels, c.latent_channels, boc, c.layers_per_block, g)
        self.quant_conv = nn.Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(c.latent_channels, c.latent_channels, 1)
        self._plans, self._weights = {}, None
        nn.Module.requires_grad_(self, False)

    # ---- diffusers conveniences ------------------------------------------------------------------------------------
    @classmethod
    def from_config(cls, config, compute_dtype="bf16", **overrides):
        d = dict(config) if isinstance(config, dict) else dict(vars(config))
        d = {k: v for k, v in d.items() if k in SD_VAE_CONFIG}
        d.update(overrides)
        return cls(compute_dtype=compute_dtype, **d)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, compute_dtype="bf16", **overrides):
        """diffusers folder layout (``config.json`` + ``diffusion_pytorch_model.{safetensors,bin}``); accepts the 0.18
        on-disk attention names ``query/key/value/proj_attn``."""
        from .checkpoint import load_weights_file
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, "config.json")) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        m = cls.from_config(cfg, compute_dtype=compute_dtype, **overrides)
        m.load_state_dict(load_weights_file(folder))
        return m

    def save_pretrained(self, path, safe_serialization=True):
        from .checkpoint import save_weights_file
        os.makedirs(path, exist_ok=True)
        cfg = dict(vars(self.config), _class_name="AutoencoderKL", _diffusers_version="0.18.2")
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2)
        save_weights_file(self.state_dict(), path, safe_serialization)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)

    def invalidate(self):
        self._plans, self._weights = {}, None

    # ---- execution -------------------------------------------------------------------------------------------------
    def _plan(self, kind, B, H, W, dev):
        key = (kind, B, H, W, str(dev), self.compute_dtype)
        p = self._plans.get(key)
        if p is None:
            if self._weights is None:
                self._weights = _VaeWeights(self, dev)
            p = (VaeEncodePlan if kind == "enc" else VaeDecodePlan)(self, self._weights, B, H, W, dev)
            self._plans[key] = p
        return p

    def _max_batch(self, H, W, kind="dec"):
        """pd_conv addresses a source tensor with 32-bit byte offsets: the widest full-resolution activation of a chunk
        must stay below 2 GiB.  Encoder: block_out_channels[0] channels at H x W; decoder: the last up block's first ResNet
        reads block_out_channels[1] channels at full resolution (the previous stage's upsampled output)."""
        esz = 4 if self.compute_dtype == "f32" else 2
        boc = self.config.block_out_channels
        ch = boc[0] if kind == "enc" else max(boc[0], boc[min(1, len(boc) - 1)])
        return max(1, ((1 << 31) - 1) // (H * W * max(ch, 64) * esz))

    def _run(self, kind, x, out_shape):
        if not x.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback): move the model and inputs to 'cuda'")
        x = x.contiguous().to(torch.float32)
        B, dev = x.shape[0], x.device
        s = 1 << (len(self.config.block_out_channels) - 1)
        H, W = (x.shape[2], x.shape[3]) if kind == "enc" else (x.shape[2] * s, x.shape[3] * s)
        out = torch.empty((B,) + out_shape, dtype=torch.float32, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        step = self._max_batch(H, W, kind)
        for b0 in range(0, B, step):
            nb = min(step, B - b0)
            self._plan(kind, nb, x.shape[2], x.shape[3], dev).run(x[b0:b0 + nb].data_ptr(), out[b0:b0 + nb].data_ptr(), stream)
        self._keepalive = x
        return out

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        c = self.config
        if x.ndim != 4 or x.shape[1] != c.in_channels:
            raise ValueError(f"expected (B, {c.in_channels}, H, W), got {tuple(x.shape)}")
        h, w = x.shape[2], x.shape[3]
        for _ in range(len(c.block_out_channels) - 1):
            h, w = h // 2, w // 2
        moments = self._run("enc", x, (2 * c.latent_channels, h, w))
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        c = self.config
        if z.ndim != 4 or z.shape[1] != c.latent_channels:
            raise ValueError(f"expected (B, {c.latent_channels}, h, w), got {tuple(z.shape)}")
        s = 1 << (len(c.block_out_channels) - 1)
        dec = self._run("dec", z, (c.out_channels, z.shape[2] * s, z.shape[3] * s))
        return DecoderOutput(sample=dec) if return_dict else (dec,)


# ---- kernel-layout weights ----------------------------------------------------------------------------------------------
class _VaeWeights:
    def __init__(self, m: AutoencoderKL, device):
        self.code, self.tdt = _DT[m.compute_dtype]
        self.proj_dim = 0
        dev, c = device, m.config
        f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        pk = lambda w, cp=None: pack_conv_weight(w.detach().to(device=dev, dtype=torch.float32), self.tdt, cp)

        def padded(conv, cin_pad, cout_pad):
            """conv weight / bias zero-padded to (cout_pad, cin_pad) channels."""
            co, ci, k, _ = conv.weight.shape
            w = torch.zeros((cout_pad, cin_pad, k, k), dtype=torch.float32, device=dev)
            w[:co, :ci] = f32(conv.weight)
            b = torch.zeros(cout_pad, dtype=torch.float32, device=dev)
            b[:co] = f32(conv.bias)
            return pk(w), b

        enc, dec = m.encoder, m.decoder
        # encoder conv_in: 3x3 over <= 3 NCHW planes = a 1x1 conv over 32 im2col channels k = ci*9 + ky*3 + kx
        cin = enc.conv_in.weight.shape[1]
        wv = torch.zeros((enc.conv_in.weight.shape[0], 32, 1, 1), dtype=torch.float32, device=dev)
        wv[:, :cin * 9, 0, 0] = f32(enc.conv_in.weight).reshape(-1, cin * 9)
        self.enc_in_w, self.enc_in_b = pk(wv), f32(enc.conv_in.bias)
        self.enc_out_w, self.enc_out_b = padded(enc.conv_out, enc.conv_out.weight.shape[1], 32)     # 2*latent -> 32 channels (zeros)
        self.quant_w, self.quant_b = padded(m.quant_conv, 32, 32)
        self.post_quant_w, self.post_quant_b = padded(m.post_quant_conv, 32, 32)
        self.dec_in_w, self.dec_in_b = padded(dec.conv_in, 32, dec.conv_in.weight.shape[0])
        self.dec_out_w, self.dec_out_b = padded(dec.conv_out, dec.conv_out.weight.shape[1], 32)
        self.enc_gn = (f32(enc.conv_norm_out.weight), f32(enc.conv_norm_out.bias), enc.conv_norm_out.eps)
        self.dec_gn = (f32(dec.conv_norm_out.weight), f32(dec.conv_norm_out.bias), dec.conv_norm_out.eps)
        self.resnets, self.attns, self.samplers = {}, {}, {}
        for name, r in m.named_modules():
            if isinstance(r, _VaeResnet):
                e = SimpleNamespace(cin=r.in_channels, cout=r.out_channels, eps=r.norm1.eps, temb_off=None)
                e.g1, e.be1, e.g2, e.be2 = f32(r.norm1.weight), f32(r.norm1.bias), f32(r.norm2.weight), f32(r.norm2.bias)
                e.w1, e.b1, e.w2, e.b2 = pk(r.conv1.weight), f32(r.conv1.bias), pk(r.conv2.weight), f32(r.conv2.bias)
                e.fused_shortcut = r.conv_shortcut is not None
                if e.fused_shortcut:                          # conv_shortcut folded into conv2 (pd_conv tail chunks)
                    ws = pk(r.conv_shortcut.weight)
                    ct = e.w2.shape[0]
                    e.w2 = torch.cat([e.w2.reshape(ct, -1, 64, 8), ws.reshape(ct, -1, 64, 8)], 1).contiguous()
                    e.b2 = e.b2 + f32(r.conv_shortcut.bias)
                self.resnets[name] = e
            elif isinstance(r, _Attention):
                e = SimpleNamespace(heads=r.heads, g=f32(r.group_norm.weight), be=f32(r.group_norm.bias), eps=r.group_norm.eps)
                wqkv = torch.cat([r.to_q.weight, r.to_k.weight, r.to_v.weight], 0).detach()
                e.wqkv = pk(wqkv[:, :, None, None])
                e.bqkv = f32(torch.cat([r.to_q.bias, r.to_k.bias, r.to_v.bias], 0))
                e.wo, e.bo = pk(r.to_out[0].weight.detach()[:, :, None, None]), f32(r.to_out[0].bias)
                self.attns[name] = e
            elif isinstance(r, _Sampler):
                self.samplers[name] = SimpleNamespace(w=pk(r.conv.weight), b=f32(r.conv.bias), padding=r.padding)
                if ".upsamplers." in name:      # Upsample2D as four 2x2 phase convolutions (UNetPlan._upconv_subpixel; inference plans)
                    self.samplers[name].w4 = tuple(pk(k) for k in upsample_phase_weights(r.conv.weight))


class _VaePlan(UNetPlan):
    """Launch plan over the UNet engine's emitters (``_gn`` / ``_conv`` / ``_resnet``), without a time embedding."""

    def __init__(self, m, w, B, H, W, device):
        self.lib = L.lib()
        self.m, self.w = m, w
        self.B, self.H, self.W, self.device = B, H, W, device      # H, W: spatial size of the plan's INPUT tensor
        self.code, self.tdt = w.code, w.tdt
        self.train = False
        self.tape, self.gn_saved, self.ops, self.bufs, self.stats = [], {}, [], [], {}
        self.groups = m.config.norm_num_groups
        self._temb_ptr_fields = []
        self._in_args = self._out_args = None
        self._cur = (None, None)
        self._build()

    def _mid(self, prefix, h):
        h = self._resnet(f"{prefix}.mid_block.resnets.0", h)
        h = self._attn_nhwc(f"{prefix}.mid_block.attentions.0", h)
        return self._resnet(f"{prefix}.mid_block.resnets.1", h)

    def run(self, x_ptr, out_ptr, stream):
        if self._cur != (x_ptr, out_ptr):
            self._set_io(x_ptr, out_ptr)
            self._cur = (x_ptr, out_ptr)
        byref, check = C.byref, L.check
        for op in self.ops:
            rc = op.fn(byref(op.args), stream)
            if rc:
                check(rc, op.what)


class VaeEncodePlan(_VaePlan):
    """(B, 3, H, W) fp32 NCHW image -> (B, 2*latent, H/8, W/8) fp32 NCHW moments (``quant_conv(encoder(x))``)."""

    def _build(self):
        m, w, c = self.m, self.w, self.m.config
        boc, B, H, W = c.block_out_channels, self.B, self.H, self.W
        h, self._in_args = self._conv(None, None, w.enc_in_w, w.enc_in_b, boc[0], ksize=1, pad=0, im2col3=c.in_channels,
                                      src_shape=(B, H, W, 32))
        self.ops[-1].what = "conv_in"
        self.ops[-1].flops = 2.0 * B * H * W * boc[0] * c.in_channels * 9
        for i, blk in enumerate(m.encoder.down_blocks):
            for j in range(len(blk.resnets)):
                h = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", h)
            if blk.downsamplers is not None:
                s = w.samplers[f"encoder.down_blocks.{i}.downsamplers.0"]
                h, _ = self._conv(h, None, s.w, s.b, h.shape[3], stride=2, pad=0)
        h = self._mid("encoder", h)
        g, be, eps = w.enc_gn
        gn = self._gn(h, None, g, be, eps)
        z, _ = self._conv(h, None, w.enc_out_w, w.enc_out_b, 32, silu=1, gn=gn, stats=False)
        _, self._out_args = self._conv(z, None, w.quant_w, w.quant_b, 2 * c.latent_channels, ksize=1, pad=0,
                                       out_mode=L.PD_OUT_NCHW_F32, cout_pad=32, y=None)

    def _set_io(self, x_ptr, out_ptr):
        self._in_args.x0 = x_ptr
        self._out_args.y = out_ptr


class VaeDecodePlan(_VaePlan):
    """(B, latent, h, w) fp32 NCHW latents -> (B, 3, 8h, 8w) fp32 NCHW image (``decoder(post_quant_conv(z))``)."""

    def _build(self):
        m, w, c = self.m, self.w, self.m.config
        boc, B, H, W = c.block_out_channels, self.B, self.H, self.W
        lat = self._act(H, W, 32)
        self._in_args = L.NchwToNhwcArgs(dtype=self.code, B=B, C=c.latent_channels, HW=H * W, Cpad=32, x=None, out=lat.data_ptr())
        self.ops.append(_Op(self.lib.pd_nchw_to_nhwc, self._in_args, "nchw_to_nhwc", 0.0, B * H * W * c.latent_channels * 4.0))
        z, _ = self._conv(lat, None, w.post_quant_w, w.post_quant_b, 32, ksize=1, pad=0, stats=False)
        h, _ = self._conv(z, None, w.dec_in_w, w.dec_in_b, boc[-1])
        h = self._mid("decoder", h)
        for i, blk in enumerate(m.decoder.up_blocks):
            for j in range(len(blk.resnets)):
                h = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", h)
            if blk.upsamplers is not None:
                s = w.samplers[f"decoder.up_blocks.{i}.upsamplers.0"]
                if self._subpixel_up_ok(h):
                    h = self._upconv_subpixel(h, s)
                else:
                    h, _ = self._conv(h, None, s.w, s.b, h.shape[3], upsample=1)
        g, be, eps = w.dec_gn
        gn = self._gn(h, None, g, be, eps)
        _, self._out_args = self._conv(h, None, w.dec_out_w, w.dec_out_b, c.out_channels, silu=1, gn=gn,
                                       out_mode=L.PD_OUT_NCHW_F32, cout_pad=32, y=None)

    def _set_io(self, x_ptr, out_ptr):
        self._in_args.x = x_ptr
        self._out_args.y = out_ptr


# ---- VaeImageProcessor (diffusers image_processor.py; the members the reference pipeline touches) -----------------------
class VaeImageProcessor:
    """``preprocess`` (PIL / numpy / tensor inputs and lists of them, ``custom_pipeline_stable_diffusion_img2img.py:638``) and ``postprocess``
    (``:717-721``: ``(x/2+.5).clamp(0,1)`` -> "pt" | "np" NHWC float | "pil"; "latent" passes through)."""

    def __init__(self, vae_scale_factor: int = 8, do_normalize: bool = True, do_resize: bool = True):
        self.config = SimpleNamespace(vae_scale_factor=vae_scale_factor, do_normalize=do_normalize, do_resize=do_resize)

    def preprocess(self, image, height=None, width=None):
        """Every input kind diffusers 0.18.2 accepts (image_processor.py ``preprocess``; the reference calls it at
        custom_pipeline_stable_diffusion_img2img.py:638): a PIL image / numpy array / tensor, or a list of one kind.  Host-side
        formatting only -- PIL images are resized down to multiples of ``vae_scale_factor`` (lanczos) and become NCHW float in [0, 1];
        numpy arrays are NHWC (or HWC each); 3-d tensors are stacked, 4-d ones concatenated; 4-channel tensors (latents) pass
        untouched; numpy / tensor sizes that are not multiples of the scale factor are refused.  Then [0, 1] -> [-1, 1], unless the
        data already holds negative values.  The result stays on the device of a tensor input and is a CPU tensor otherwise (the
        pipeline moves it, ``prepare_latents``)."""
        import numpy as np
        from PIL import Image
        kinds = (Image.Image, np.ndarray, torch.Tensor)
        if isinstance(image, kinds):
            image = [image]
        elif not (isinstance(image, list) and len(image) > 0 and all(isinstance(i, kinds) for i in image)):
            raise ValueError("Input is in incorrect format: PIL image, numpy array, tensor or a list of them")
        f = self.config.vae_scale_factor
        resize = getattr(self.config, "do_resize", True)
        if isinstance(image[0], Image.Image):
            if resize:
                image = [im.resize(((width or im.width) - (width or im.width) % f, (height or im.height) - (height or im.height) % f),
                                   resample=Image.LANCZOS) for im in image]
            arr = np.stack([np.array(im).astype(np.float32) / 255.0 for im in image], axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            image = torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 1, 2)))
        elif isinstance(image[0], np.ndarray):
            arr = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
            if arr.ndim == 3:
                arr = arr[..., None]
            image = torch.from_numpy(np.ascontiguousarray(arr.transpose(0, 3, 1, 2)))
            if resize and (image.shape[2] % f or image.shape[3] % f):
                raise ValueError(f"images must have height and width divisible by {f}, got {tuple(image.shape[2:])}")
        else:
            # (one 4-d tensor: torch.cat of a single tensor would only copy it)
            image = (image[0] if len(image) == 1 else torch.cat(image, dim=0)) if image[0].ndim == 4 else torch.stack(image, dim=0)
            if image.shape[1] == 4:                           # latents pass untouched
                return image
            if resize and (image.shape[2] % f or image.shape[3] % f):
                raise ValueError(f"images must have height and width divisible by {f}, got {tuple(image.shape[2:])}")
        if self.config.do_normalize and float(image.min()) >= 0:   # [0, 1] -> [-1, 1]; already-normalised data is left alone
            image = 2.0 * image - 1.0
        return image

    def postprocess(self, image: torch.Tensor, output_type: str = "pil", do_denormalize=None):
        if output_type == "latent":
            return image
        if output_type not in ("pt", "np", "pil"):
            raise ValueError(f"output_type {output_type!r} not in ('latent', 'pt', 'np', 'pil')")
        B, Cc, H, W = image.shape
        x = image.contiguous().to(torch.float32)
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=x.device)
        a = L.PostprocArgs(B=B, C=Cc, H=H, W=W, x=x.data_ptr(), out_f32=out.data_ptr(), out_u8=None)
        L.check(L.lib().pd_postproc(C.byref(a), torch.cuda.current_stream(x.device).cuda_stream), "pd_postproc")
        if output_type == "pt":
            return out.permute(0, 3, 1, 2)
        arr = out.cpu().numpy()
        if output_type == "np":
            return arr
        from .pipeline import numpy_to_pil
        return numpy_to_pil(arr)
