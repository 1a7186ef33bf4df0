"""CPU fp32 restatement of diffusers 0.18.2 ``AutoencoderKL`` in its Stable-Diffusion configuration and of
``VaeImageProcessor`` -- what ``CustomStableDiffusionImg2ImgPipeline`` calls as ``vae.encode(image).latent_dist.sample(g)``
(``custom_pipeline_stable_diffusion_img2img.py:431``), ``vae.decode(latents / scaling_factor, return_dict=False)[0]``
(``:709-711``), ``image_processor.preprocess / postprocess`` (``:638,717-721``) and what ``_encode_to_latents`` /
``_decode_to_images`` wrap (``utils_Img2Img.py:827-847``).  Test infrastructure only (see ``oracle/__init__.py``).

**Parity unpinned** like the rest of the oracle: diffusers is not importable here.  The structure follows SURVEY.md
Appendix A.11 and is pinned by the public parameter counts of the SD VAE, 34 163 664 (encoder + quant_conv) and
49 490 199 (decoder + post_quant_conv) = 83 653 863 (``tests/test_oracle_vae.py``).  ``state_dict`` names are diffusers'
in-memory names (``encoder.down_blocks.i.resnets.j.conv1.weight``, ``encoder.mid_block.attentions.0.to_q.weight`` ...).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import AttentionRef, Downsample2DRef, Upsample2DRef

SD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                     layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, sample_size=512)


class VaeResnetRef(nn.Module):
    """``ResnetBlock2D(temb_channels=None, eps=1e-6)``: GN -> SiLU -> conv -> GN -> SiLU -> conv, + (1x1-projected) input."""

    def __init__(self, cin, cout, groups, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _BlockRef(nn.Module):
    pass


class _MidRef(nn.Module):
    """``UNetMidBlock2D(temb_channels=None, attention_head_dim=channels)``: ResNet, one-head attention, ResNet."""

    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnetRef(ch, ch, groups), VaeResnetRef(ch, ch, groups)])
        self.attentions = nn.ModuleList([AttentionRef(ch, 1, groups, 1e-6)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class EncoderRef(nn.Module):
    def __init__(self, in_channels, latent_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, ch in enumerate(boc):
            cin, out = out, ch
            b = _BlockRef()
            b.resnets = nn.ModuleList([VaeResnetRef(cin if j == 0 else out, out, groups) for j in range(layers)])
            b.downsamplers = nn.ModuleList([Downsample2DRef(out, 0)]) if i != len(boc) - 1 else None
            self.down_blocks.append(b)
        self.mid_block = _MidRef(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            for r in b.resnets:
                x = r(x)
            if b.downsamplers is not None:
                x = b.downsamplers[0](x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DecoderRef(nn.Module):
    def __init__(self, out_channels, latent_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(latent_channels, boc[-1], 3, padding=1)
        self.mid_block = _MidRef(boc[-1], groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out = rev[0]
        for i, ch in enumerate(rev):
            prev, out = out, ch
            b = _BlockRef()
            b.resnets = nn.ModuleList([VaeResnetRef(prev if j == 0 else out, out, groups) for j in range(layers + 1)])
            b.upsamplers = nn.ModuleList([Upsample2DRef(out)]) if i != len(boc) - 1 else None
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            for r in b.resnets:
                x = r(x)
            if b.upsamplers is not None:
                x = b.upsamplers[0](x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class DiagonalGaussianRef:
    """``DiagonalGaussianDistribution``: moments (B, 2C, h, w) -> mean | logvar.clamp(-30, 20); ``sample`` draws with
    ``randn_tensor(mean.shape, generator)`` on the parameters' device."""

    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class AutoencoderKLRef(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(SD_VAE_CONFIG)
        cfg.update(kwargs)
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        self.config = SimpleNamespace(**cfg)
        c = self.config
        self.encoder = EncoderRef(c.in_channels, c.latent_channels, c.block_out_channels, c.layers_per_block, c.norm_num_groups)
        self.decoder = DecoderRef(c.out_channels, c.latent_channels, c.block_out_channels, c.layers_per_block, c.norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * c.latent_channels, 2 * c.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(c.latent_channels, c.latent_channels, 1)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianRef(self.quant_conv(self.encoder(x))))

    def decode(self, z, return_dict=True):
        dec = self.decoder(self.post_quant_conv(z))
        return SimpleNamespace(sample=dec) if return_dict else (dec,)


# ---- VaeImageProcessor (diffusers image_processor.py; the members the reference pipeline touches) -----------------------
def vae_preprocess_ref(image, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True) -> torch.Tensor:
    """``VaeImageProcessor.preprocess`` of diffusers 0.18.2 (image_processor.py; called at
    custom_pipeline_stable_diffusion_img2img.py:638), every accepted input: a PIL image / numpy array / tensor or a list of
    one kind.  PIL: resized DOWN to multiples of ``vae_scale_factor`` (lanczos), uint8 / 255, NHWC -> NCHW.  numpy: NHWC (or HWC
    each) float in [0, 1] -> NCHW.  Tensors: 3-d ones are stacked, 4-d ones concatenated; 4-channel tensors (latents) are returned
    untouched.  Sizes that are not multiples of the scale factor are refused (numpy / tensor).  Then [0, 1] -> [-1, 1] unless the
    data already holds negative values (returned as they are, with a deprecation warning in diffusers)."""
    import numpy as np
    try:
        from PIL import Image
        pil_t = Image.Image
    except Exception:      # pragma: no cover
        pil_t, Image = (), None
    kinds = (pil_t, np.ndarray, torch.Tensor) if pil_t else (np.ndarray, torch.Tensor)
    if isinstance(image, kinds):
        image = [image]
    elif not (isinstance(image, list) and len(image) > 0 and all(isinstance(i, kinds) for i in image)):
        raise ValueError("Input is in incorrect format: PIL image, numpy array, tensor or a list of them")
    f = vae_scale_factor
    if pil_t and isinstance(image[0], pil_t):
        if do_resize:
            image = [im.resize((im.width - im.width % f, im.height - im.height % f), resample=Image.LANCZOS) for im in image]
        arr = np.stack([np.array(im).astype(np.float32) / 255.0 for im in image], axis=0)
        if arr.ndim == 3:
            arr = arr[..., None]
        image = torch.from_numpy(arr.transpose(0, 3, 1, 2))
    elif isinstance(image[0], np.ndarray):
        arr = np.concatenate(image, axis=0) if image[0].ndim == 4 else np.stack(image, axis=0)
        if arr.ndim == 3:
            arr = arr[..., None]
        image = torch.from_numpy(arr.transpose(0, 3, 1, 2))
        if do_resize and (image.shape[2] % f or image.shape[3] % f):
            raise ValueError(f"images must have height and width divisible by {f}, got {tuple(image.shape[2:])}")
    else:
        # (one 4-d tensor: torch.cat of a single tensor would only copy it)
        image = (image[0] if len(image) == 1 else torch.cat(image, dim=0)) if image[0].ndim == 4 else torch.stack(image, dim=0)
        if image.shape[1] == 4:
            return image
        if do_resize and (image.shape[2] % f or image.shape[3] % f):
            raise ValueError(f"images must have height and width divisible by {f}, got {tuple(image.shape[2:])}")
    if float(image.min()) < 0:
        do_normalize = False
    return 2.0 * image - 1.0 if do_normalize else image


def vae_postprocess_ref(image: torch.Tensor, output_type: str = "np"):
    """``(x / 2 + 0.5).clamp(0, 1)`` -> "pt" tensor | "np" NHWC float32 array | "latent" (untouched)."""
    if output_type == "latent":
        return image
    image = (image / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return image
    return image.permute(0, 2, 3, 1).float().numpy()
