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


def closed_form_parameter_count(cfg):
    """Parameter count written out from the layer list of SURVEY.md Appendix A.3-A.6 (diffusers 0.18.2 `ResnetBlock2D`, `Attention`,
    `Downsample2D` / `Upsample2D`, `TimestepEmbedding`, `nn.Embedding`), independent of the oracle's module code."""
    boc, L = list(cfg["block_out_channels"]), cfg["layers_per_block"]
    tdim = 4 * boc[0]

    def resnet(cin, cout):
        n = 2 * cin + (cin * cout * 9 + cout) + (tdim * cout + cout) + 2 * cout + (cout * cout * 9 + cout)
        return n + ((cin * cout + cout) if cin != cout else 0)

    def attention(ch):
        return 2 * ch + 4 * (ch * ch + ch)

    def sampler(ch):
        return ch * ch * 9 + ch
    n = cfg["in_channels"] * boc[0] * 9 + boc[0]                                   # conv_in
    n += (boc[0] * tdim + tdim) + (tdim * tdim + tdim)                             # time_embedding
    if cfg.get("num_class_embeds"):
        n += cfg["num_class_embeds"] * tdim
    out = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        cin, out = out, boc[i]
        for j in range(L):
            n += resnet(cin if j == 0 else out, out) + (attention(out) if t.startswith("Attn") else 0)
        if i != len(boc) - 1:
            n += sampler(out)
    n += 2 * resnet(boc[-1], boc[-1]) + attention(boc[-1])                         # mid block
    rev = boc[::-1]
    out = rev[0]
    for i, t in enumerate(cfg["up_block_types"]):
        prev, out = out, rev[i]
        skip_in = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = skip_in if j == L else out
            n += resnet((prev if j == 0 else out) + skip, out) + (attention(out) if t.startswith("Attn") else 0)
        if i != len(boc) - 1:
            n += sampler(out)
    return n + 2 * boc[0] + (boc[0] * cfg["out_channels"] * 9 + cfg["out_channels"])  # conv_norm_out + conv_out


@pytest.mark.parametrize("name,count", [("super_small", 15_725_443), ("small_denoiser_config", 62_826_243), ("orig_google_ddpm", 113_673_219),
                                        ("ddpm_cifar10", 35_746_307), ("SD_2-1_config", 641_914_883)])
def test_parameter_counts_closed_form(name, count):
    """The closed form reproduces the four published / surveyed counts, which validates it; it then gives the known answer for
    models_configs/denoiser/SD_2-1_config.json (641 914 883), which the oracle (built on the meta device: no 2.6 GB allocation)
    must match."""
    assert closed_form_parameter_count(UNET_CONFIGS[name]) == count
    with torch.device("meta"):
        m = CondUNet2DRef(**UNET_CONFIGS[name])
    assert sum(p.numel() for p in m.parameters()) == count


def test_product_config_tables_equal_the_oracle_s():
    """phendiff_amd.configs carries the same VALUES as the oracle's copies for every key both know (two independent transcriptions of
    models_configs/denoiser/*.json), all four shipped denoisers included."""
    from phendiff_amd.configs import UNET_CONFIGS as PROD
    for prod_name, ref_name in (("super_small",) * 2, ("small_denoiser_config",) * 2, ("orig_google_ddpm_model_denoiser", "orig_google_ddpm"),
                                ("SD_2-1_config",) * 2):
        for k, v in UNET_CONFIGS[ref_name].items():
            if k in PROD[prod_name]:
                assert PROD[prod_name][k] == v, (prod_name, k)
            else:
                assert k == "num_class_embeds" and v is None, (prod_name, k)


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
