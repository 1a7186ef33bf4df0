"""Oracle known-answer checks (CPU).  The oracle is unpinned by the reference (no tests, no
importable diffusers); these are the independent facts it can be held to: SURVEY.md section 4."""
import pytest
import torch

from oracle import CondUNet2DRef, UNET_CONFIGS


@pytest.mark.parametrize("name,count", [
    ("super_small", 15_725_443),
    ("small_denoiser_config", 62_826_243),
    ("orig_google_ddpm", 113_673_219),      # public google/ddpm-celebahq-256 parameter count
    ("ddpm_cifar10", 35_746_307),           # public google/ddpm-cifar10-32 parameter count
])
def test_parameter_counts(name, count):
    m = CondUNet2DRef(**UNET_CONFIGS[name])
    assert sum(p.numel() for p in m.parameters()) == count


def test_state_dict_names_are_diffusers_names():
    m = CondUNet2DRef(**UNET_CONFIGS["super_small"])
    keys = set(m.state_dict().keys())
    for k in ["conv_in.weight", "time_embedding.linear_1.weight", "time_embedding.linear_2.bias",
              "class_embedding.weight", "down_blocks.0.resnets.0.norm1.weight",
              "down_blocks.0.resnets.1.time_emb_proj.weight", "down_blocks.0.downsamplers.0.conv.weight",
              "down_blocks.1.resnets.0.conv_shortcut.weight", "down_blocks.2.attentions.1.to_q.weight",
              "down_blocks.2.attentions.0.group_norm.bias", "mid_block.attentions.0.to_out.0.weight",
              "mid_block.resnets.1.conv2.bias", "up_blocks.0.attentions.2.to_v.bias",
              "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.2.resnets.2.conv_shortcut.bias",
              "conv_norm_out.weight", "conv_out.bias"]:
        assert k in keys, k
    assert m.time_embed_dim == 256


def test_forward_shapes_and_class_conditioning():
    torch.manual_seed(0)
    cfg = dict(UNET_CONFIGS["super_small"], sample_size=32)
    m = CondUNet2DRef(**cfg).eval()
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        a = m(x, 10, class_labels=torch.tensor([0, 1])).sample
        b = m(x, torch.tensor(10), class_labels=torch.tensor([1, 1])).sample
        z = m(x, 10, class_emb=torch.zeros(2, 256)).sample
        e = m(x, 10, class_emb=m.class_embedding(torch.tensor([0, 1]))).sample
    assert a.shape == (2, 3, 32, 32)
    assert torch.allclose(a[1], b[1], atol=1e-6) and not torch.allclose(a[0], b[0], atol=1e-4)
    assert torch.allclose(a, e, atol=1e-6) and not torch.allclose(a, z, atol=1e-4)
    with pytest.raises(ValueError):
        m(x, 10, class_labels=torch.tensor([0, 1]), class_emb=torch.zeros(2, 256))
    with pytest.raises(ValueError):
        m(x, 10)
