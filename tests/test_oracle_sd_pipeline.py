"""Oracle drift + structural checks for the latent-diffusion pipeline restatement (oracle/sd_pipeline_ref.py)."""
import os
import sys

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
from make_golden import sd_tiny_pipe  # noqa: E402

from oracle import hack_class_embedding_ref, sd_cfg_forward_start_ref, sd_ddib_ref  # noqa: E402


def test_oracle_reproduces_golden_sd_transfers():
    d = np.load(os.path.join(GOLDEN, "sd_tiny_32_s4.npz"))
    pipe = sd_tiny_pipe()
    x, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
    out, inverted, latents = sd_ddib_ref(pipe, x, labels, 1 - labels, 4, generator=torch.Generator().manual_seed(11))
    assert np.allclose(latents.numpy(), d["latents"], atol=1e-5)
    assert np.allclose(inverted.numpy(), d["inverted"], atol=1e-5)
    assert np.allclose(out, d["ddib_out"], atol=1e-5) and out.shape == (4, 32, 32, 3) and out.min() >= 0 and out.max() <= 1
    cfg_out, cfg_lat = sd_cfg_forward_start_ref(pipe, x, 1 - labels, 3.0, 0.5, 4, generator=torch.Generator().manual_seed(12),
                                                output_type="np+latent")
    assert np.allclose(cfg_lat.numpy(), d["cfg_latents"], atol=1e-5) and np.allclose(cfg_out, d["cfg_out"], atol=1e-5)


def test_pipeline_structure():
    pipe = sd_tiny_pipe()
    assert pipe.vae_scale_factor == 2
    pipe.scheduler.set_timesteps(10)
    assert pipe.scheduler.timesteps.tolist() == [901, 801, 701, 601, 501, 401, 301, 201, 101, 1]     # leading + steps_offset 1
    ts, n = pipe.get_timesteps(10, 0.5)
    assert n == 5 and ts.tolist() == [401, 301, 201, 101, 1]
    ts, n = pipe.get_timesteps(10, 1)
    assert n == 10
    ts, n = pipe.get_timesteps(10, 0)
    assert n == 0 and len(ts) == 0
    e = hack_class_embedding_ref(torch.randn(3, 96))
    assert e.shape == (3, 77, 96) and float(e[:, 1:].abs().max()) == 0
    # guidance_scale <= 1 and None both disable guidance; class_labels as int / list / tensor agree
    lat = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    kw = dict(image=lat, strength=1, add_forward_noise_to_image=False, num_inference_steps=2, output_type="latent")
    a = pipe(class_labels=1, guidance_scale=None, **kw)
    b = pipe(class_labels=[1], guidance_scale=1.0, **kw)
    c = pipe(class_labels=torch.tensor([1]), guidance_scale=0, **kw)
    assert torch.equal(a, b) and torch.equal(a, c)
    # guidance with w = 1 reproduces the conditional prediction (uncond + 1 * (cond - uncond))
    g = pipe(class_labels=[1], guidance_scale=torch.tensor([1.0]), **kw)
    assert torch.allclose(g, a, atol=1e-5)
    # strength 0: nothing runs, latents are returned as they are
    z = pipe(class_labels=[1], image=lat, strength=0, add_forward_noise_to_image=False, num_inference_steps=4, output_type="latent")
    assert torch.equal(z, lat)
    # the constructor's deprecation fix-ups
    from oracle import DDIMSchedulerRef, SDImg2ImgPipelineRef
    p2 = SDImg2ImgPipelineRef(pipe.vae, pipe.unet, DDIMSchedulerRef(steps_offset=0, clip_sample=True), pipe.class_embedding)
    assert p2.scheduler.config.steps_offset == 1 and p2.scheduler.config.clip_sample is False
