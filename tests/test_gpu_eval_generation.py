"""Eval-time generation during training on MI355X (``utils_training.py:642-941``): train a few steps so the
EMA shadow differs from the weights, swap the EMA weights in, sample per class from one seeded generator through the product
pipeline, swap back — against the CPU oracle doing the same with diffusers' EMAModel.store / copy_to / restore semantics."""
import numpy as np
import pytest
import torch

from test_oracle_training import SCHED

pytestmark = pytest.mark.gpu


def _pair(mode):
    import phendiff_amd as P
    from oracle import CondUNet2DRef, DDIMSchedulerRef, TINY_CONFIG0_UNET, TrainingLoopRef, synthetic_two_class_batch
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in TINY_CONFIG0_UNET.items() if k in keys})
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **TINY_CONFIG0_UNET)
    m.load_state_dict(r.state_dict())
    loop = TrainingLoopRef(r, DDIMSchedulerRef(**SCHED), lr=2e-4, warmup=0, total_steps=100, proba_uncond=0.0, seed=5)
    sched = P.DDIMScheduler(**SCHED)
    tr = P.UNetTrainer(m.to("cuda:0"), sched, lr=2e-4, use_ema=True)
    clean, labels = synthetic_two_class_batch(8, 32, 77)
    for _ in range(3):
        noise, ts, _ = loop.sample(clean)
        loop.step(clean, labels, noise, ts, False)
        noisy = sched.add_noise(clean.cuda(), noise.cuda(), ts.cuda())
        tr.step(noisy, ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda(), lr=loop.lr)
    return r, loop, m, tr, sched


# f32: three optimisation steps (weights agree to ~2e-5) + 4 sampling steps, guidance w = 2 amplifies the difference ~2x
@pytest.mark.parametrize("mode,tol", [("f32", 6e-4), ("bf16", 4e-2)])
@pytest.mark.parametrize("w", [None, 2.0])
def test_ema_generation_matches_oracle_and_restores_training_weights(mode, tol, w):
    import phendiff_amd as P
    from phendiff_amd.eval_generation import generate_samples
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef, EMASwapRef, eval_generation_ddim_ref
    from test_gpu_unet_ddib import rel
    r, loop, m, tr, sched = _pair(mode)
    r.eval()
    flat_before = tr.opt.flat.clone()
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
    y_before = m(x, 100, torch.tensor([0, 1]).cuda()).sample.clone()

    seen = []
    pipe = P.ConditionalDDIMPipeline(unet=m, scheduler=sched)
    gen = generate_samples(pipe, nb_classes=2, nb_generated_images=5, eval_batch_size=3, num_inference_steps=4,
                           guidance_factor=w, trainer=tr, generator=torch.Generator().manual_seed(11),
                           class_names=["dmso", "latrunculin"], on_class_done=lambda c, n, b: seen.append((c, n, len(b))))
    assert seen == [(0, "dmso", 2), (1, "latrunculin", 2)]

    ref_pipe = ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**SCHED))
    with EMASwapRef(loop.ema, loop.params):
        want = eval_generation_ddim_ref(ref_pipe, 2, [3, 2], w, 4, torch.Generator().manual_seed(11), eval_batch_size=3)
    for c in (0, 1):
        got = gen.images_of(c)
        assert got.shape == want[c][1].shape == (5, 32, 32, 3)
        assert gen.files_of(c) == want[c][0]
        assert rel(torch.from_numpy(got), torch.from_numpy(want[c][1])) < tol
    assert gen.batches[0].uint8.dtype == np.uint8

    # the EMA weights really were the ones sampled with: the training weights give a different picture
    no_ema = generate_samples(pipe, nb_classes=1, nb_generated_images=3, eval_batch_size=3, num_inference_steps=4,
                              guidance_factor=w, generator=torch.Generator().manual_seed(11))
    assert rel(torch.from_numpy(no_ema.images_of(0)), torch.from_numpy(gen.batches[0].images)) > 10 * tol or mode == "bf16"

    # restore: flat parameters bit-identical, and so is a forward through the re-packed weights
    assert torch.equal(tr.opt.flat, flat_before)
    assert torch.equal(m(x, 100, torch.tensor([0, 1]).cuda()).sample, y_before)


def test_unconditional_model_generates_one_pass_with_the_reference_seed_on_device():
    import phendiff_amd as P
    from phendiff_amd.eval_generation import EVAL_SEED, generate_samples
    _, _, m, tr, sched = _pair("bf16")
    pipe = P.ConditionalDDIMPipeline(unet=m, scheduler=sched)
    a = generate_samples(pipe, nb_classes=2, nb_generated_images=4, eval_batch_size=4, num_inference_steps=3, proba_uncond=1,
                         trainer=tr)
    assert [b.class_name for b in a.batches] == ["unconditional"] and a.images_of(0).shape == (4, 32, 32, 3)
    b = generate_samples(pipe, nb_classes=2, nb_generated_images=4, eval_batch_size=4, num_inference_steps=3, proba_uncond=1,
                         trainer=tr, generator=torch.Generator(device="cuda").manual_seed(EVAL_SEED))
    assert np.array_equal(a.images_of(0), b.images_of(0))          # default generator = device generator, seed 5742877512
    assert np.isfinite(a.images_of(0)).all()


def test_sd_generation_from_noise_latents():
    import phendiff_amd as P
    from phendiff_amd.eval_generation import generate_samples, latents_preview
    from oracle import eval_generation_sd_ref
    from test_gpu_sd_pipeline import make_pipe
    from test_gpu_unet_ddib import rel
    ref, pipe = make_pipe("f32")
    # the starting latents come from the global device RNG (the reference's torch.randn without the generator): replay them
    torch.cuda.manual_seed(99)
    draws = [torch.randn(bs, 4, 8, 8, device="cuda").cpu() for _ in (0, 1) for bs in (2, 1)]
    torch.cuda.manual_seed(99)
    got = generate_samples(pipe, nb_classes=2, nb_generated_images=3, eval_batch_size=2, num_inference_steps=3,
                           guidance_factor=3.0, model_type="StableDiffusion", generator=torch.Generator().manual_seed(4),
                           latent_hw=(8, 8))
    want = eval_generation_sd_ref(ref, 2, [2, 1], 3.0, 3, torch.Generator().manual_seed(4), latent_hw=(8, 8),
                                  initial_latents=draws)
    for c in (0, 1):
        assert rel(torch.from_numpy(got.images_of(c)), torch.from_numpy(want[c][0])) < 2e-4
        lat = torch.cat([b.latents for b in got.batches if b.class_label == c]).cpu()
        assert rel(lat, want[c][1]) < 2e-4
        assert latents_preview(lat).shape == (3, 1, 8, 8)
