"""Known answers for the SD-tier oracle: the restated ``UNet2DConditionModel`` must have
the public SD-2.1 UNet's parameter count, diffusers' state_dict names, and the 77-token cross-attention must reduce to the
per-head scalar gate of A.10."""
import torch

from oracle import CustomEmbeddingRef, SD21_UNET_CONFIG, UNet2DConditionRef, class_emb_to_encoder_hidden_states

TINY = dict(in_channels=4, out_channels=4, block_out_channels=(64, 128), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
            attention_head_dim=(1, 2), cross_attention_dim=96, norm_num_groups=32)


def test_sd21_unet_parameter_count_and_names():
    with torch.device("meta"):
        m = UNet2DConditionRef(**SD21_UNET_CONFIG)
    assert sum(p.numel() for p in m.parameters()) == 865_910_724        # public stabilityai/stable-diffusion-2-1 UNet
    names = dict(m.named_parameters())
    for n, shape in {"conv_in.weight": (320, 4, 3, 3), "time_embedding.linear_1.weight": (1280, 320),
                     "down_blocks.0.attentions.0.norm.weight": (320,), "down_blocks.0.attentions.0.proj_in.weight": (320, 320),
                     "down_blocks.1.attentions.1.transformer_blocks.0.attn1.to_q.weight": (640, 640),
                     "down_blocks.2.attentions.0.transformer_blocks.0.attn2.to_k.weight": (1280, 1024),
                     "mid_block.attentions.0.transformer_blocks.0.ff.net.0.proj.weight": (10240, 1280),
                     "mid_block.attentions.0.transformer_blocks.0.ff.net.2.weight": (1280, 5120),
                     "up_blocks.0.resnets.0.conv1.weight": (1280, 2560, 3, 3), "up_blocks.3.resnets.2.conv1.weight": (320, 640, 3, 3),
                     "up_blocks.1.attentions.2.transformer_blocks.0.attn1.to_out.0.bias": (1280,),
                     "conv_out.weight": (4, 320, 3, 3)}.items():
        assert tuple(names[n].shape) == shape, n
    assert "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.bias" not in names       # q/k/v carry no bias
    assert not hasattr(m.down_blocks[3], "attentions") and not hasattr(m.up_blocks[0], "attentions")
    assert m.down_blocks[3].downsamplers is None and m.up_blocks[3].upsamplers is None


def test_tiny_forward_and_cross_attention_gate():
    torch.manual_seed(0)
    m = UNet2DConditionRef(**TINY).eval()
    emb = CustomEmbeddingRef(2, 96)
    labels = torch.tensor([0, 1])
    ehs = class_emb_to_encoder_hidden_states(emb(labels))
    assert ehs.shape == (2, 77, 96) and float(ehs[:, 1:].abs().max()) == 0.0
    x = torch.randn(2, 4, 16, 16)
    with torch.no_grad():
        out = m(x, torch.tensor([10, 500]), ehs).sample
        assert out.shape == x.shape and torch.isfinite(out).all()
        # A.10: with k = v = 0 on the 76 padded tokens, cross attention = e^s / (e^s + 76) * v_class per head
        blk = m.down_blocks[0].attentions[0].transformer_blocks[0]
        h = torch.randn(2, 256, 64)
        ref = blk.attn2(h, ehs)
        a = blk.attn2
        q, k0, v0 = a.to_q(h), a.to_k(ehs[:, :1]), a.to_v(ehs[:, :1])
        d = 64 // a.heads
        s = (q.reshape(2, 256, a.heads, d) * k0.reshape(2, 1, a.heads, d)).sum(-1) / d ** 0.5
        gate = torch.exp(s) / (torch.exp(s) + 76.0)
        o = (gate[..., None] * v0.reshape(2, 1, a.heads, d)).reshape(2, 256, 64)
        assert float((a.to_out[0](o) - ref).abs().max()) < 1e-5
        # unconditional pass (all-zero tokens): cross attention contributes only its output bias
        z = blk.attn2(h, torch.zeros_like(ehs))
        assert float((z - a.to_out[0].bias).abs().max()) < 1e-6
