"""CPU fp32 restatement of diffusers 0.18.2 ``UNet2DConditionModel`` in its Stable-Diffusion-2.1 configuration (what
``CustomStableDiffusionImg2ImgPipeline`` drives: ``custom_pipeline_stable_diffusion_img2img.py:680-686``, and what
``_SD_prediction_wrapper`` trains: ``utils_training.py:459-496``) and of the reference's ``CustomEmbedding``
(``src/custom_embedding/custom_embedding.py:36-47``).  Test infrastructure only (see ``oracle/__init__.py``).

**Parity unpinned** like the rest of the oracle: diffusers is not importable here.  The structure follows SURVEY.md
Appendix A.9 / A.10 and is pinned by the public parameter count of the SD-2.1 UNet, 865 910 724
(``tests/test_oracle_sd_unet.py``).  ``state_dict`` names are diffusers' (``down_blocks.i.attentions.j.transformer_blocks.0
.attn1.to_q.weight`` ...), so weights are exchangeable with the HIP engine.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet_ref import ResnetBlock2DRef, Downsample2DRef, Upsample2DRef, timestep_embedding_ref

SD21_UNET_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5,
    flip_sin_to_cos=True, freq_shift=0, use_linear_projection=True, sample_size=96)


class AttentionRef(nn.Module):
    """diffusers ``Attention`` as used by ``BasicTransformerBlock`` (q/k/v without bias, out projection with bias;
    ``AttnProcessor2_0`` -> ``F.scaled_dot_product_attention``)."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])

    def forward(self, x, context=None):
        ctx = x if context is None else context
        B, N, _ = x.shape
        split = lambda t: t.reshape(B, t.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(x)), split(self.to_k(ctx)), split(self.to_v(ctx)))
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, -1))


class GEGLURef(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForwardRef(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLURef(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlockRef(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = AttentionRef(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = AttentionRef(dim, heads, dim_head, cross_attention_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForwardRef(dim)

    def forward(self, x, ehs):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), ehs) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DRef(nn.Module):
    """``Transformer2DModel(use_linear_projection=True, num_layers=1)``."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlockRef(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, ehs):
        B, C, H, W = x.shape
        res = x
        h = self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, ehs)
        h = self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2)
        return h + res


class _TimestepEmbeddingRef(nn.Module):
    def __init__(self, cin, tdim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, tdim)
        self.linear_2 = nn.Linear(tdim, tdim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class _BlockRef(nn.Module):
    pass


class UNet2DConditionRef(nn.Module):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 down_block_types=SD21_UNET_CONFIG["down_block_types"], up_block_types=SD21_UNET_CONFIG["up_block_types"],
                 attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, norm_num_groups=32, norm_eps=1e-5,
                 flip_sin_to_cos=True, freq_shift=0, use_linear_projection=True, sample_size=None):
        super().__init__()
        assert use_linear_projection, "SD-2.1 form only"
        boc = tuple(block_out_channels)
        heads = (attention_head_dim,) * len(boc) if isinstance(attention_head_dim, int) else tuple(attention_head_dim)
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                                      layers_per_block=layers_per_block, down_block_types=tuple(down_block_types),
                                      up_block_types=tuple(up_block_types), attention_head_dim=heads,
                                      cross_attention_dim=cross_attention_dim, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                                      flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift, sample_size=sample_size)
        g, eps, tdim = norm_num_groups, norm_eps, boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbeddingRef(boc[0], tdim)
        tf = lambda ch, nh: Transformer2DRef(nh, ch // nh, ch, cross_attention_dim, g)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, t in enumerate(down_block_types):
            in_ch, out_ch = out_ch, boc[i]
            b = _BlockRef()
            b.resnets = nn.ModuleList([ResnetBlock2DRef(in_ch if j == 0 else out_ch, out_ch, tdim, g, eps) for j in range(layers_per_block)])
            if t == "CrossAttnDownBlock2D":
                b.attentions = nn.ModuleList([tf(out_ch, heads[i]) for _ in range(layers_per_block)])
            b.downsamplers = nn.ModuleList([Downsample2DRef(out_ch, padding=1)]) if i != len(boc) - 1 else None
            self.down_blocks.append(b)
        self.mid_block = _BlockRef()
        self.mid_block.resnets = nn.ModuleList([ResnetBlock2DRef(boc[-1], boc[-1], tdim, g, eps) for _ in range(2)])
        self.mid_block.attentions = nn.ModuleList([tf(boc[-1], heads[-1])])
        self.up_blocks = nn.ModuleList()
        rev, rheads = list(reversed(boc)), list(reversed(heads))
        out_ch = rev[0]
        for i, t in enumerate(up_block_types):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            b = _BlockRef()
            n = layers_per_block + 1
            b.resnets = nn.ModuleList([ResnetBlock2DRef((prev if j == 0 else out_ch) + (in_ch if j == n - 1 else out_ch), out_ch, tdim, g, eps)
                                       for j in range(n)])
            if t == "CrossAttnUpBlock2D":
                b.attentions = nn.ModuleList([tf(out_ch, rheads[i]) for _ in range(n)])
            b.upsamplers = nn.ModuleList([Upsample2DRef(out_ch)]) if i != len(boc) - 1 else None
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True):
        c = self.config
        B = sample.shape[0]
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.reshape(-1).expand(B) if t.numel() == 1 else t.reshape(-1)
        emb = self.time_embedding(timestep_embedding_ref(t, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift))
        ehs = encoder_hidden_states
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, emb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, ehs)
                skips.append(h)
            if blk.downsamplers is not None:
                h = blk.downsamplers[0](h)
                skips.append(h)
        h = self.mid_block.resnets[0](h, emb)
        h = self.mid_block.attentions[0](h, ehs)
        h = self.mid_block.resnets[1](h, emb)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(torch.cat([h, skips.pop()], 1), emb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, ehs)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0](h)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return SimpleNamespace(sample=h) if return_dict else (h,)


class CustomEmbeddingRef(nn.Module):
    """``CustomEmbedding`` (src/custom_embedding/custom_embedding.py): ``inner_module = nn.Embedding(num_classes, dim)``,
    class id -> (B, dim)."""

    def __init__(self, num_classes=2, class_embedding_dim=1024):
        super().__init__()
        self.inner_module = nn.Embedding(num_classes, class_embedding_dim)

    def forward(self, x):
        return self.inner_module(x)


def class_emb_to_encoder_hidden_states(class_emb, seq_len=77):
    """``_SD_prediction_wrapper`` (utils_training.py:472-484): the class embedding as token 0 followed by 76 zero tokens;
    an unconditional pass uses ``zeros(B, 77, dim)``."""
    bs, ed = class_emb.shape
    class_emb = class_emb.reshape(bs, 1, ed)
    return torch.cat([class_emb, torch.zeros_like(class_emb).repeat(1, seq_len - 1, 1)], dim=1)
