"""Known-answer checks that pin the AutoencoderKL restatement (oracle/vae_ref.py): the public SD-VAE parameter counts
(SURVEY.md A.11), shapes, the diagonal-Gaussian sampling rule and VaeImageProcessor's tensor paths."""
import numpy as np
import torch

from oracle import AutoencoderKLRef, vae_postprocess_ref, vae_preprocess_ref

TINY = dict(block_out_channels=(32, 64), layers_per_block=1)


def test_sd_vae_parameter_counts():
    m = AutoencoderKLRef()
    n = lambda mod: sum(p.numel() for p in mod.parameters())
    assert n(m.encoder) + n(m.quant_conv) == 34_163_664
    assert n(m.decoder) + n(m.post_quant_conv) == 49_490_199
    assert n(m) == 83_653_863
    names = set(m.state_dict())
    for k in ("encoder.down_blocks.0.resnets.1.conv2.weight", "encoder.down_blocks.2.downsamplers.0.conv.bias",
              "encoder.mid_block.attentions.0.to_q.weight", "decoder.up_blocks.3.resnets.2.norm1.weight",
              "decoder.up_blocks.0.upsamplers.0.conv.weight", "quant_conv.weight", "post_quant_conv.bias",
              "encoder.down_blocks.1.resnets.0.conv_shortcut.weight", "decoder.up_blocks.2.resnets.0.conv_shortcut.weight"):
        assert k in names, k
    assert not any("time_emb_proj" in k for k in names)


def test_tiny_shapes_and_sampling_rule():
    torch.manual_seed(0)
    m = AutoencoderKLRef(**TINY).eval()
    x = torch.rand(2, 3, 24, 16) * 2 - 1
    with torch.no_grad():
        d = m.encode(x).latent_dist
        assert d.mean.shape == (2, 4, 12, 8)
        noise = torch.randn(2, 4, 12, 8)
        z = d.sample(noise=noise)
        assert torch.equal(z, d.mean + torch.exp(0.5 * d.logvar) * noise)
        assert torch.equal(d.mode(), d.mean)
        g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
        assert torch.equal(d.sample(g1), d.mean + d.std * torch.randn(d.mean.shape, generator=g2))
        y = m.decode(z, return_dict=False)[0]
        assert y.shape == x.shape and m.decode(z).sample.equal(y)
    # clamp of the log-variance
    mom = torch.cat([torch.zeros(1, 4, 2, 2), torch.full((1, 4, 2, 2), 100.0)], 1)
    from oracle.vae_ref import DiagonalGaussianRef
    assert float(DiagonalGaussianRef(mom).logvar.max()) == 20.0


def test_image_processor():
    x = torch.rand(2, 3, 8, 8)
    assert torch.equal(vae_preprocess_ref(x), 2 * x - 1)
    assert torch.equal(vae_preprocess_ref(2 * x - 1), 2 * x - 1) or float((2 * x - 1).min()) >= 0
    lat = torch.randn(2, 4, 4, 4)
    assert vae_preprocess_ref(lat) is lat
    out = vae_postprocess_ref(torch.tensor([[[[-3.0, 0.0], [0.5, 3.0]]]]).repeat(1, 3, 1, 1), "np")
    assert out.shape == (1, 2, 2, 3) and np.allclose(out[0, :, :, 0], [[0, 0.5], [0.75, 1.0]])
    assert vae_postprocess_ref(lat, "latent") is lat


def test_image_processor_pil_numpy_and_list_inputs():
    """VaeImageProcessor.preprocess (diffusers 0.18.2 image_processor.py; custom_pipeline_stable_diffusion_img2img.py:638): PIL images
    are resized down to multiples of the VAE scale factor and land in [-1, 1] NCHW; numpy NHWC in [0, 1] likewise; lists are batched;
    sizes that are not multiples of 8 are refused for numpy / tensors; data that is already negative is not normalised again."""
    import pytest
    from PIL import Image
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, size=(19, 27, 3), dtype=np.uint8)          # H = 19, W = 27 -> resized to 16 x 24
    out = vae_preprocess_ref(Image.fromarray(u8))
    assert tuple(out.shape) == (1, 3, 16, 24) and float(out.min()) >= -1 and float(out.max()) <= 1
    u8b = rng.integers(0, 256, size=(16, 24, 3), dtype=np.uint8)          # no resize: exact values
    out = vae_preprocess_ref([Image.fromarray(u8b), Image.fromarray(u8b[::-1].copy())])
    want = torch.from_numpy(u8b.astype(np.float32) / 255.0).permute(2, 0, 1) * 2 - 1
    assert tuple(out.shape) == (2, 3, 16, 24) and torch.equal(out[0], want) and torch.equal(out[1], want.flip(1))
    arr = rng.random((2, 16, 8, 3), dtype=np.float32)
    out = vae_preprocess_ref(arr)
    assert torch.equal(out, torch.from_numpy(arr).permute(0, 3, 1, 2) * 2 - 1)
    out = vae_preprocess_ref([arr[0], arr[1]])                            # list of HWC arrays: stacked
    assert torch.equal(out, torch.from_numpy(arr).permute(0, 3, 1, 2) * 2 - 1)
    neg = arr * 2 - 1
    assert torch.equal(vae_preprocess_ref(neg), torch.from_numpy(neg).permute(0, 3, 1, 2))      # already in [-1, 1]
    t = torch.rand(3, 8, 8)
    assert torch.equal(vae_preprocess_ref([t, t]), torch.stack([t, t]) * 2 - 1)
    with pytest.raises(ValueError):
        vae_preprocess_ref(rng.random((1, 12, 8, 3), dtype=np.float32))   # 12 % 8 != 0
    with pytest.raises(ValueError):
        vae_preprocess_ref("not an image")
