"""CustomStableDiffusionImg2ImgPipeline on MI355X (SD UNet + VAE + DDIM on the HIP engine) against the CPU oracle and the
committed golden vectors: DDIB through the latent space, CFG forward-start transfer, call-surface options."""
import os
import sys

import numpy as np
import pytest
import torch

from test_gpu_unet_ddib import rel

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, GOLDEN)
from make_golden import SD_SCHED, SD_TINY_UNET, SD_TINY_VAE, sd_tiny_pipe  # noqa: E402


def make_pipe(mode):
    import phendiff_amd as P
    ref = sd_tiny_pipe()
    unet = P.SDUNet2DConditionModel(compute_dtype=mode, **SD_TINY_UNET)
    unet.load_state_dict(ref.unet.state_dict())
    vae = P.AutoencoderKL(compute_dtype=mode, **SD_TINY_VAE)
    vae.load_state_dict(ref.vae.state_dict())
    emb = P.CustomEmbedding(2, SD_TINY_UNET["cross_attention_dim"])
    emb.load_state_dict(ref.class_embedding.state_dict())
    pipe = P.CustomStableDiffusionImg2ImgPipeline(vae.to("cuda:0"), unet.to("cuda:0"), P.DDIMScheduler(**SD_SCHED), emb.to("cuda:0"))
    return ref, pipe


# measured maxima (profiles/r2_parity_errors.json): f32 3.1e-6, bf16 1.5e-2, fp16 1.8e-3
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 3e-2), ("fp16", 4e-3)])
def test_sd_ddib_matches_golden(mode, tol):
    import phendiff_amd as P
    d = np.load(os.path.join(GOLDEN, "sd_tiny_32_s4.npz"))
    _, pipe = make_pipe(mode)
    x, labels = torch.from_numpy(d["images"]).cuda(), torch.from_numpy(d["labels"]).cuda()
    # the pieces, as _ddib strings them together (utils_Img2Img.py:575-607)
    lat, [cond] = P.LDM_preprocess(pipe, x, [labels], generator=torch.Generator().manual_seed(11))
    assert cond.shape == (4, 77, 96)
    assert rel(lat, torch.from_numpy(d["latents"])) < tol
    inv = P.inversion(pipe, lat, cond, 4)
    assert rel(inv, torch.from_numpy(d["inverted"])) < tol
    out = P.ddib(pipe, x, labels, 1 - labels, 4, generator=torch.Generator().manual_seed(11))
    assert isinstance(out, np.ndarray) and out.shape == (4, 32, 32, 3)
    # the decoded images carry the VAE decoder's own 16-bit error on top of the latent trajectory's (measured: f32 7.6e-6, bf16 3.2e-2, fp16 4.0e-3)
    assert rel(out, d["ddib_out"]) < {"f32": 2e-5, "bf16": 6e-2, "fp16": 8e-3}[mode]


# measured maxima (profiles/r2_parity_errors.json): f32 3.1e-6, bf16 1.5e-2, fp16 1.8e-3
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 3e-2), ("fp16", 4e-3)])
def test_sd_cfg_forward_start_matches_golden(mode, tol):
    import phendiff_amd as P
    d = np.load(os.path.join(GOLDEN, "sd_tiny_32_s4.npz"))
    _, pipe = make_pipe(mode)
    x, labels = torch.from_numpy(d["images"]).cuda(), torch.from_numpy(d["labels"]).cuda()
    out, lat = pipe(image=x, class_labels=1 - labels, strength=0.5, num_inference_steps=4, guidance_scale=3.0,
                    generator=torch.Generator().manual_seed(12), output_type="np+latent")
    assert rel(lat, torch.from_numpy(d["cfg_latents"])) < tol
    assert np.linalg.norm(out - d["cfg_out"]) / np.linalg.norm(d["cfg_out"]) < tol
    out2 = P.classifier_free_guidance_forward_start(pipe, x, 1 - labels, 3.0, 0.5, 4, generator=torch.Generator().manual_seed(12))
    assert np.array_equal(out2, out)


def test_sd_pipeline_call_surface():
    import phendiff_amd as P
    ref, pipe = make_pipe("f32")
    assert pipe.vae_scale_factor == 2 and set(pipe.components) == {"vae", "unet", "scheduler", "class_embedding"}
    g = torch.Generator().manual_seed(1)
    lat = torch.randn(2, 4, 8, 8, generator=g)
    kw = dict(strength=1, add_forward_noise_to_image=False, num_inference_steps=3)
    want = ref(image=lat, class_labels=[1, 0], guidance_scale=None, output_type="latent", **kw)
    a = pipe(image=lat.cuda(), class_labels=[1, 0], guidance_scale=None, output_type="latent", **kw)
    assert rel(a, want) < 2e-4
    b = pipe(image=lat.cuda(), class_labels=torch.tensor([1, 0]).cuda(), guidance_scale=1.0, output_type="latent", **kw)
    assert torch.equal(a, b)
    # per-sample guidance weights (1-D tensor), pre-computed embeddings, pt / pil outputs
    w = torch.tensor([1.5, 4.0])
    emb = ref.class_embedding(torch.tensor([1, 0])).detach()
    want = ref(image=lat, class_labels_embeds=emb, guidance_scale=w, output_type="pt", **kw)
    got = pipe(image=lat.cuda(), class_labels_embeds=emb.cuda(), guidance_scale=w.cuda(), output_type="pt", **kw)
    assert got.shape == (2, 3, 16, 16) and rel(got, want) < 2e-4
    pil = pipe(image=lat.cuda(), class_labels=[1, 0], output_type="pil", **kw)
    assert len(pil) == 2 and pil[0].size == (16, 16)
    # int label, generation from noise (image=None, latent_shape), strength 0 returns the latents untouched
    one = pipe(latent_shape=(1, 4, 8, 8), class_labels=1, strength=1, num_inference_steps=2, output_type="latent")
    assert one.shape == (1, 4, 8, 8) and bool(torch.isfinite(one).all())
    z = pipe(image=lat.cuda(), class_labels=[1, 0], strength=0, add_forward_noise_to_image=False, num_inference_steps=4, output_type="latent")
    assert torch.equal(z.cpu(), lat)
    # callback protocol
    seen = []
    pipe(image=lat.cuda(), class_labels=[1, 0], output_type="latent", callback=lambda i, t, l: seen.append((i, int(t))), **kw)
    assert [s[0] for s in seen] == [0, 1, 2]
    # input checks of the reference
    with pytest.raises(ValueError):
        pipe(class_labels=[1])
    with pytest.raises(ValueError):
        pipe(image=lat.cuda(), class_labels=[1, 0], strength=1.5)
    with pytest.raises(ValueError):
        pipe(image=lat.cuda(), class_labels=[1, 0], class_labels_embeds=emb.cuda())
    with pytest.raises(ValueError):
        pipe(image=lat.cuda())


def test_sd_pipeline_save_and_reload(tmp_path):
    import phendiff_amd as P
    _, pipe = make_pipe("f32")
    pipe.save_pretrained(str(tmp_path / "sd"))
    for sub in ("vae", "unet", "scheduler", "class_embedding"):
        assert (tmp_path / "sd" / sub).is_dir()
    p2 = P.CustomStableDiffusionImg2ImgPipeline.from_pretrained(str(tmp_path / "sd"), compute_dtype="f32").to("cuda:0")
    lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(2)).cuda()
    kw = dict(image=lat, class_labels=[0, 1], strength=1, add_forward_noise_to_image=False, num_inference_steps=2, output_type="np")
    assert np.array_equal(pipe(**kw), p2(**kw))


def test_tensor_to_pil():
    import phendiff_amd as P
    from oracle import tensor_to_uint8_ref
    g = torch.Generator().manual_seed(9)
    x = torch.rand(3, 3, 8, 10, generator=g) * 2 - 1
    pil = P.tensor_to_PIL(x.cuda())
    got = np.stack([np.asarray(im) for im in pil])
    want = tensor_to_uint8_ref(x)
    assert got.shape == want.shape == (3, 8, 10, 3) and int(np.abs(got.astype(int) - want.astype(int)).max()) <= 1     # <= 1 LSB (A20)
    lat = torch.randn(2, 4, 6, 6, generator=g)
    for ch in ("mean", 2):
        pil = P.tensor_to_PIL(lat.cuda(), ch)
        got = np.stack([np.asarray(im) for im in pil])
        assert pil[0].mode == "L" and np.array_equal(got, tensor_to_uint8_ref(lat, ch)[..., 0])
    assert P.tensor_to_PIL(x[:1].cuda()).size == (10, 8)        # a single image is returned bare


def test_sd_pipeline_eta_and_seeded_variance_noise():
    """eta > 0 (stochastic DDIM) with a CPU generator: the variance noise is drawn on the CPU like diffusers' randn_tensor, so the
    HIP pipeline and the oracle consume identical draws -- with and without classifier-free guidance."""
    ref, pipe = make_pipe("f32")
    lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(21))
    for gs in (None, 2.5):
        kw = dict(class_labels=[0, 1], strength=1, add_forward_noise_to_image=False, num_inference_steps=3, eta=0.7,
                  guidance_scale=gs, output_type="latent")
        want = ref(image=lat, generator=torch.Generator().manual_seed(22), **kw)
        got = pipe(image=lat.cuda(), generator=torch.Generator().manual_seed(22), **kw)
        assert rel(got, want) < 2e-4, (gs, rel(got, want))
    # per-sample generators (list): the VAE posterior draws and the forward noise follow the list; eta stays deterministic here
    x = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(23)) * 2 - 1
    gens = lambda: [torch.Generator().manual_seed(31), torch.Generator().manual_seed(32)]
    a = pipe(image=x.cuda(), class_labels=[1, 0], strength=0.5, num_inference_steps=4, generator=gens(), output_type="latent")
    b = pipe(image=x.cuda(), class_labels=[1, 0], strength=0.5, num_inference_steps=4, generator=gens(), output_type="latent")
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


@pytest.mark.skipif(bool(os.environ.get("PD_SKIP_LONG_TESTS")), reason="full-size SD stack on the CPU oracle (~2 minutes)")
def test_sd_img2img_full_size_stack_vs_oracle():
    """BASELINE configs[4] at its real sizes: the SD-2.1 UNet (865.9 M) + the SD VAE (83.7 M) + CustomEmbedding(2, 1024), one
    512x512 image -> 64x64 latents, DDIB with 2 + 2 DDIM steps: exact-fp32 engine against the CPU oracle (random init)."""
    import phendiff_amd as P
    from oracle import (AutoencoderKLRef, CustomEmbeddingRef, DDIMSchedulerRef, SDImg2ImgPipelineRef, UNet2DConditionRef,
                        sd_ddib_ref)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    torch.manual_seed(0)
    r_unet = UNet2DConditionRef(**P.SD21_UNET_CONFIG).eval()
    r_vae = AutoencoderKLRef(**P.SD_VAE_CONFIG).eval()
    r_emb = CustomEmbeddingRef(2, 1024)
    sched_cfg = P.SCHEDULER_CONFIGS["SD_orig_config"]
    ref_pipe = SDImg2ImgPipelineRef(r_vae, r_unet, DDIMSchedulerRef(**sched_cfg), r_emb)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(1, 3, 512, 512, generator=g) * 2 - 1)
    labels = torch.tensor([0])
    want, inverted, latents = sd_ddib_ref(ref_pipe, x, labels, 1 - labels, 2, generator=torch.Generator().manual_seed(5))
    unet = P.SDUNet2DConditionModel(compute_dtype="f32", **P.SD21_UNET_CONFIG)
    unet.load_state_dict(r_unet.state_dict())
    vae = P.AutoencoderKL(compute_dtype="f32", **P.SD_VAE_CONFIG)
    vae.load_state_dict(r_vae.state_dict())
    emb = P.CustomEmbedding(2, 1024)
    emb.load_state_dict(r_emb.state_dict())
    pipe = P.CustomStableDiffusionImg2ImgPipeline(vae.to("cuda:0"), unet.to("cuda:0"), P.DDIMScheduler(**sched_cfg), emb.to("cuda:0"))
    lat, [cond] = P.LDM_preprocess(pipe, x.cuda(), [labels.cuda()], generator=torch.Generator().manual_seed(5))
    assert lat.shape == (1, 4, 64, 64) and rel(lat, latents) < 1e-4
    got = P.ddib(pipe, x.cuda(), labels.cuda(), (1 - labels).cuda(), 2, generator=torch.Generator().manual_seed(5))
    assert got.shape == want.shape == (1, 512, 512, 3)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-3
    # the same stack on the 16-bit engines: fp16 is configs[4]'s stated dtype (img2img_comparison.py:56-59) -- 512x512 VAE
    # activations and 1 280-channel UNet activations in fp16 storage must stay finite (a failure names the buffer), then parity
    from conftest import record_error
    from phendiff_amd.diagnostics import assert_finite_activations
    del pipe, unet, vae
    torch.cuda.empty_cache()
    for mode, tol_lat, tol_img in (("fp16", 1e-3, 6e-3), ("bf16", 6e-3, 5e-2)):     # measured 3.7e-4 / 3.1e-3 and 3.0e-3 / 2.5e-2
        unet = P.SDUNet2DConditionModel(compute_dtype=mode, **P.SD21_UNET_CONFIG)
        unet.load_state_dict(r_unet.state_dict())
        vae = P.AutoencoderKL(compute_dtype=mode, **P.SD_VAE_CONFIG)
        vae.load_state_dict(r_vae.state_dict())
        pipe = P.CustomStableDiffusionImg2ImgPipeline(vae.to("cuda:0"), unet.to("cuda:0"), P.DDIMScheduler(**sched_cfg), emb)
        lat, _ = P.LDM_preprocess(pipe, x.cuda(), [labels.cuda()], generator=torch.Generator().manual_seed(5))
        got = P.ddib(pipe, x.cuda(), labels.cuda(), (1 - labels).cuda(), 2, generator=torch.Generator().manual_seed(5))
        rep = assert_finite_activations(list(unet._plans.values()) + list(vae._plans.values()), what=f"SD img2img stack, {mode} engine",
                                        limit=65504.0 if mode == "fp16" else float("inf"))
        assert np.isfinite(got).all()
        e_lat = record_error(float((lat.cpu() - latents).norm() / latents.norm()))
        e_img = record_error(float(np.linalg.norm(got - want) / np.linalg.norm(want)))
        assert e_lat < tol_lat and e_img < tol_img, (mode, e_lat, e_img, rep["__max__"])
        del pipe, unet, vae
        torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_sd_ddib_graph_replays_the_eager_transfer_bit_for_bit(mode):
    """SDDDIBGraph (VAE encode -> sample -> S inversion + S denoising SD-UNet steps -> VAE decode -> post-processing in ONE hipGraph)
    == the eager `ddib(pipe, ...)` (utils_Img2Img.py:566-612) on the same posterior noise; a second replay with other inputs reuses
    the graph."""
    import phendiff_amd as P
    d = np.load(os.path.join(GOLDEN, "sd_tiny_32_s4.npz"))
    _, pipe = make_pipe(mode)
    x, labels = torch.from_numpy(d["images"]).cuda(), torch.from_numpy(d["labels"]).cuda()
    eager = P.ddib(pipe, x, labels, 1 - labels, 4, generator=torch.Generator().manual_seed(11))
    g = P.SDDDIBGraph(pipe, batch_size=4, num_inference_steps=4, height=32, width=32)
    out = g.run(x, labels, 1 - labels, generator=torch.Generator().manual_seed(11))
    torch.cuda.synchronize()
    assert torch.equal(out.images.cpu(), torch.from_numpy(eager))
    if mode == "f32":
        assert rel(out.inverted, torch.from_numpy(d["inverted"])) < 2e-5 and rel(out.images, d["ddib_out"]) < 2e-5
    x2 = (x.flip(0) * 0.9).contiguous()
    eager2 = P.ddib(pipe, x2, 1 - labels, labels, 4, generator=torch.Generator().manual_seed(5))
    out2 = g.run(x2, 1 - labels, labels, generator=torch.Generator().manual_seed(5))
    torch.cuda.synchronize()
    assert torch.equal(out2.images.cpu(), torch.from_numpy(eager2))


# ---- gradient-guided transfer, latent-diffusion branch (utils_Img2Img.py:651-760 with a CustomStableDiffusionImg2ImgPipeline) ----------
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 3e-2), ("fp16", 5e-3)])      # measured 2.4e-6 / 1.2e-2
def test_sd_guidance_gradient_through_unet_matches_autograd(mode, tol):
    """d Lp(x0_pred, target) / d latents through the SD UNet (input-gradient-only backward plan) and the scheduler's x0 formula, one
    step: what ``torch.autograd.grad(losses_seq, images)`` returns with ``pipe.unet(images, t, target_class_embeds)`` (:718-745)."""
    import ctypes as C
    import phendiff_amd as P
    import phendiff_amd._lib as L
    from oracle import hack_class_embedding_ref, lp_loss_ref
    ref, pipe = make_pipe(mode)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 4, 16, 16, generator=g)
    target = lat + 0.3 * torch.randn(lat.shape, generator=g)
    labels = torch.tensor([0, 1])
    ehs = hack_class_embedding_ref(ref._encode_class(labels, False)).detach()
    images = lat.clone().requires_grad_(True)
    ref.scheduler.set_timesteps(4)
    t = ref.scheduler.timesteps[1]
    mo = ref.unet(images, t, ehs).sample
    x0 = ref.scheduler.step(mo, t, images).pred_original_sample
    losses = lp_loss_ref(x0, target, 2)
    (want,) = torch.autograd.grad([losses[0], losses[1]], images)

    dev = torch.device("cuda:0")
    plan = pipe.unet.input_grad_plan(2, 16, 16, 77, dev)
    st = torch.cuda.current_stream().cuda_stream
    im, tg = lat.to(dev).contiguous(), target.to(dev)
    out, d_out, d_dir = (torch.empty_like(im) for _ in range(3))
    pipe.scheduler.set_timesteps(4)
    plan.forward(im, torch.full((2,), float(t), device=dev), ehs.to(dev), out, st)
    sa, sb, _, _, _ = pipe.scheduler.step_coefficients(t)
    c = pipe.scheduler.config
    partial = torch.empty(2 * 2, dtype=torch.float64, device=dev)
    ls = torch.empty(2, device=dev)
    a = L.LpGuidanceArgs(numel=im.numel(), per_sample=im[0].numel(), pred_type=L.PD_PRED[c.prediction_type], clip=int(bool(c.clip_sample)),
                         clip_range=float(c.clip_sample_range), sqrt_a=sa, sqrt_b=sb, p=2.0, sample=im.data_ptr(), model_out=out.data_ptr(),
                         target=tg.data_ptr(), partial=partial.data_ptr(), splits=2, d_model_out=d_out.data_ptr(),
                         d_sample_direct=d_dir.data_ptr(), losses=ls.data_ptr())
    L.check(L.lib().pd_lp_guidance(C.byref(a), st), "pd_lp_guidance")
    from phendiff_amd.img2img import GUIDANCE_GRAD_SCALE
    S = float(GUIDANCE_GRAD_SCALE) if mode == "fp16" else 1.0      # fp16 engine: the static scale of the guided transfer (round 6)
    plan.backward(d_out * S, st)
    torch.cuda.synchronize()
    assert torch.isfinite(plan.dsample).all()
    assert rel(out, mo.detach()) < {"f32": 2e-5, "bf16": 3e-2, "fp16": 4e-3}[mode]
    assert rel(ls, losses.detach()) < {"f32": 1e-5, "bf16": 2e-2, "fp16": 3e-3}[mode]
    assert rel(d_dir + plan.dsample / S, want) < tol


@pytest.mark.parametrize("mode,tol_lat,tol_img", [("f32", 5e-5, 2e-4), ("bf16", 8e-2, 1e-1), ("fp16", 1e-2, 2e-2)])
def test_sd_gradient_guided_transfer_matches_golden(mode, tol_lat, tol_img):
    """_linear_interp_custom_guidance_inverted_start with the latent-diffusion pipeline, end to end, against the committed oracle
    vectors (tests/golden/make_golden.py --sd-guided: tiny stack, 32x32 images = 16x16 latents, S = 3, p = 2, loss scale 0.5 -- raised
    from the reference default 1e-3 so that a wrong gradient could not hide inside the tolerance): _LDM_preprocess -> inversion ->
    per-step UNet forward + input-gradient backward + Lp push in latent space -> _decode_to_images -> min-max renormalisation."""
    import phendiff_amd as P
    d = np.load(os.path.join(GOLDEN, "guided_sd_tiny_32_s3.npz"))
    _, pipe = make_pipe(mode)
    x, labels = torch.from_numpy(d["images"]).cuda(), torch.from_numpy(d["labels"]).cuda()
    S, p, scale = 3, float(d["p"]), float(d["guidance_loss_scale"])
    # the pieces (utils_Img2Img.py:663-696)
    lat, (c_orig, c_target) = P.LDM_preprocess(pipe, x, [labels, 1 - labels], generator=torch.Generator().manual_seed(13))
    inv = P.inversion(pipe, lat, c_orig, S)
    assert rel(inv, torch.from_numpy(d["inverted"])) < tol_lat
    guided, step_losses = P.custom_guided_generation(pipe, inv, c_target, p, scale, S, return_losses=True)
    assert rel(guided, torch.from_numpy(d["guided_latents"])) < tol_lat
    assert len(step_losses) == S and all(float(l.min()) >= 0 for l in step_losses)
    unguided = P.custom_guided_generation(pipe, inv, c_target, p, 0.0, S)                    # the guidance must have had an effect
    assert rel(guided, unguided) > 0.15
    # ... and the whole function
    out = P.linear_interp_custom_guidance_inverted_start(pipe, x, labels, 1 - labels, p, scale, S, output_type="pt",
                                                         generator=torch.Generator().manual_seed(13))
    assert tuple(out.shape) == (2, 3, 32, 32) and float(out.min()) == -1.0 and float(out.max()) == 1.0      # min-max renormalised
    ref = torch.from_numpy(d["out"])
    if mode == "f32":
        assert rel(out, ref) < tol_img
    else:
        # the min-max renormalisation (image - min) / max is an affine map fixed by two EXTREME pixels: a 16-bit engine moves those
        # by its ordinary error and with them offset and scale of the whole image (measured: 0.28 relative although the latents
        # agree to 8e-2).  Compare up to that affine map: centred, unit-norm images.
        a, b = out.cpu() - out.mean().cpu(), ref - ref.mean()
        assert float((a / a.norm() - b / b.norm()).norm()) < tol_img
    arr = P.linear_interp_custom_guidance_inverted_start(pipe, x, labels, 1 - labels, p, scale, S, generator=torch.Generator().manual_seed(13))
    assert isinstance(arr, np.ndarray) and arr.shape == (2, 32, 32, 3) and arr.min() >= 0 and arr.max() <= 1


def test_cross_attention_context_is_projected_once_per_context():
    """Round 6: the cross-attention k / v of the SD UNet depend on the class context only; an inference plan keeps them while the SAME
    encoder_hidden_states tensor (unmodified) comes back -- every step after the first of a sampling loop -- and re-projects on a new
    tensor, an in-place change, or re-packed weights.  Outputs are bit-identical to recomputing every time."""
    _, pipe = make_pipe("bf16")
    unet = pipe.unet
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(2, 4, 16, 16, generator=g).cuda()
    ehs_a = pipe._encode_class(class_labels=torch.tensor([0, 1]).cuda(), device=lat.device, do_classifier_free_guidance=False)
    from phendiff_amd.sd_pipeline import hack_class_embedding
    ehs_a = hack_class_embedding(ehs_a).contiguous()
    ehs_b = hack_class_embedding(pipe._encode_class(class_labels=torch.tensor([1, 0]).cuda(), device=lat.device,
                                                    do_classifier_free_guidance=False)).contiguous()
    plan = unet.plan_for(2, 16, 16, 77, lat.device)
    n_ctx = sum(1 for op in plan.ops if op.ctx)
    assert n_ctx == sum(1 for _ in unet._weights.transformers) > 0
    calls = []
    orig = plan.run
    plan.run = lambda *a, context=True: (calls.append(context), orig(*a, context=context))[1]
    o1 = unet(lat, 500, ehs_a, return_dict=False)[0].clone()
    o2 = unet(lat, 500, ehs_a, return_dict=False)[0].clone()             # same tensor: cached
    o3 = unet(lat, 500, ehs_a.clone(), return_dict=False)[0].clone()     # equal values, another object: projected again
    assert calls == [True, False, True] and torch.equal(o1, o2) and torch.equal(o1, o3)
    ob = unet(lat, 500, ehs_b, return_dict=False)[0].clone()
    assert not torch.equal(ob, o1)
    ehs_b.copy_(ehs_a)                                                   # in-place change of the cached tensor: version bump -> fresh
    o4 = unet(lat, 500, ehs_b, return_dict=False)[0].clone()
    assert calls[-2:] == [True, True] and torch.equal(o4, o1)
