"""End-to-end parity on MI355X: the HIP UNet / pipeline / DDIB against the CPU oracle on identical seeded
weights and inputs, and against the committed golden fixtures (tests/golden, made by the oracle)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    from conftest import record_error
    return record_error(float((a - b).norm() / (b.norm() + 1e-30)))


def make_pair(name, size, mode, seed=0):
    import phendiff_amd as P
    from oracle import CondUNet2DRef
    torch.manual_seed(seed)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in dict(P.UNET_CONFIGS[name], sample_size=size).items() if k in keys}).eval()
    # default init leaves the class table N(0,1) and convs small; make every path matter
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS[name], sample_size=size))
    m.load_state_dict(r.state_dict())
    return r, m.to("cuda:0")


def synth_batch(B, size, seed=1234):
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(B) % 2
    x = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    x = (x + 0.25 * (2 * labels.float() - 1).view(B, 1, 1, 1)).clamp(-1, 1)
    return x, labels


# one UNet evaluation: fp32 engine within fp32 round-off of the oracle; 16-bit engines within their storage round-off.
# Tolerances = ~2x the maxima measured on MI355X (profiles/r2_parity_errors.json: f32 4.0e-6, bf16 1.21e-2, fp16 1.48e-3)
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 2.5e-2), ("fp16", 3e-3)])
@pytest.mark.parametrize("size", [32, 64])
def test_unet_forward_super_small(mode, tol, size):
    r, m = make_pair("super_small", size, mode)
    x, labels = synth_batch(3, size)
    for t in (2999, 640, 0):
        with torch.no_grad():
            ref = r(x, t, class_labels=labels).sample
        got = m(x.cuda(), t, class_labels=labels.cuda()).sample
        assert got.shape == ref.shape and got.dtype == torch.float32
        assert rel(got, ref) < tol, (mode, size, t, rel(got, ref))
    # class_emb path (zeros = unconditional) and positional call form unet(x, t, labels)
    with torch.no_grad():
        ref = r(x, 100, class_emb=torch.zeros(3, 256)).sample
    got = m(x.cuda(), torch.tensor(100), class_emb=torch.zeros(3, 256, device="cuda")).sample
    assert rel(got, ref) < tol
    got2 = m(x.cuda(), torch.tensor(100), labels.cuda()).sample
    with torch.no_grad():
        ref2 = r(x, 100, labels).sample
    assert rel(got2, ref2) < tol


VARIANTS = [dict(center_input_sample=True), dict(resnet_time_scale_shift="scale_shift"), dict(class_embed_type="timestep"),
            dict(class_embed_type="identity", num_class_embeds=None),
            dict(center_input_sample=True, resnet_time_scale_shift="scale_shift", class_embed_type="timestep")]


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 2.5e-2)])
@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "+".join(sorted(v)))
def test_unet_forward_config_variants(mode, tol, variant):
    """The constructor switches of CustomCondUNet2DModel that no shipped config sets (cond_unet_2d.py:103,146-153,272-273,
    301-302): ``center_input_sample`` (2x - 1 as one elementwise launch in front of conv_in), ``resnet_time_scale_shift =
    "scale_shift"`` (the projected embedding modulates norm2's affine inside pd_gn_finalize instead of being added by conv1),
    ``class_embed_type`` "timestep" (a second TimestepEmbedding over the sinusoid of the labels) and "identity" (rows as given)."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef
    torch.manual_seed(11)
    cfg = dict(P.UNET_CONFIGS["super_small"], sample_size=32, **variant)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in cfg.items() if k in keys}).eval()
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **cfg)
    assert sum(p.numel() for p in m.parameters()) == sum(p.numel() for p in r.parameters())
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    x, labels = synth_batch(3, 32)
    if variant.get("class_embed_type") == "identity":
        labels = torch.randn(3, m.time_embed_dim, generator=torch.Generator().manual_seed(5))
    elif variant.get("class_embed_type") == "timestep":
        labels = torch.tensor([0.0, 1.0, 7.0])
    for t in (2999, 17):
        with torch.no_grad():
            ref = r(x, t, class_labels=labels).sample
        got = m(x.cuda(), t, class_labels=labels.cuda()).sample
        assert rel(got, ref) < tol, (mode, variant, t)
    # the same model through the DDIB transfer, eager and as one hipGraph -- every class_embed_type (round 6: the captured trajectory
    # holds its conditioning in static row buffers, img2img._ClassRows: int64 labels | fp32 "identity" rows | fp32 label values that go
    # through the class MLP inside the graph)
    from oracle import DDIMSchedulerRef, ConditionalDDIMPipelineRef, ddib_ref
    scfg = dict(P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    kind = variant.get("class_embed_type")
    if kind == "identity":
        g5 = torch.Generator().manual_seed(6)
        orig, target = labels, torch.randn(3, m.time_embed_dim, generator=g5)
    elif kind == "timestep":
        orig, target = labels, torch.tensor([1.0, 0.0, 3.0])
    else:
        orig = synth_batch(3, 32)[1]
        target = 1 - orig
    rpipe = ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**scfg))
    pipe = P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**scfg))
    if kind == "identity":
        # the pipeline takes embedding ROWS as `class_emb` (check_inputs: class_labels is 1-D, pipeline_conditionial_ddim.py:99-105), so
        # `_ddib` is spelled out: inversion under the original rows, then the pipeline call with the target rows
        from oracle import inversion_ref
        kw = dict(w=0, num_inference_steps=2, add_forward_noise_to_image=False, frac_diffusion_skipped=0)
        want = rpipe(class_labels=target, start_image=inversion_ref(rpipe, x, orig, 2), **kw).images      # (the oracle's pipeline has no check_inputs)
        eager = lambda o, t: pipe(None, class_emb=t.cuda(), start_image=P.inversion(pipe, x.cuda(), o.cuda(), 2), output_type="numpy", **kw).images
    else:
        want, _ = ddib_ref(rpipe, x, orig, target, 2)
        eager = lambda o, t: P.ddib(pipe, x.cuda(), o.cuda(), t.cuda(), 2)
    got = eager(orig, target)
    assert rel(got, want) < (2e-5 if mode == "f32" else 2e-2)
    runner = P.DDIBGraph(pipe, batch_size=3, num_inference_steps=2, height=32, width=32)
    got_g = runner.run(x.cuda(), orig.cuda(), target.cuda()).images
    assert np.array_equal(got_g.cpu().numpy(), got)
    got_g2 = runner.run(x.cuda(), target.cuda(), orig.cuda()).images            # a replay with other conditioning: the buffers, not the nodes
    assert np.array_equal(got_g2.cpu().numpy(), eager(target, orig))
    # CFG forward-start as one graph against its eager form (cond + uncond evaluations, fused guidance step)
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(8)).cuda()
    cfgg = P.CFGForwardStartGraph(pipe, batch_size=3, num_inference_steps=4, height=32, width=32, frac_diffusion_skipped=0.5, guidance_scale=2.5)
    a1 = cfgg.run(x.cuda(), target.cuda(), noise).images.cpu().numpy()
    eager = P.CFGForwardStartGraph(pipe, batch_size=3, num_inference_steps=4, height=32, width=32, frac_diffusion_skipped=0.5, guidance_scale=2.5,
                                   use_graph=False)
    assert np.array_equal(a1, eager.run(x.cuda(), target.cuda(), noise).images.cpu().numpy())
    if kind == "identity":
        plan = m.plan_for(3, 32, 32, torch.device("cuda:0"))
        with pytest.raises(ValueError, match="time_embed_dim"):      # a short row buffer is refused, not read past its end
            plan.temb_rows(torch.zeros(3, device="cuda"), torch.zeros(3, dtype=torch.int64, device="cuda"), None,
                           torch.cuda.current_stream().cuda_stream)
        with pytest.raises(ValueError, match="identity"):
            runner.run(x.cuda(), torch.zeros(3, dtype=torch.int64).cuda(), target.cuda())


def test_unet_forward_small_denoiser_f32():
    r, m = make_pair("small_denoiser_config", 32, "f32")
    x, labels = synth_batch(2, 32)
    with torch.no_grad():
        ref = r(x, 1500, class_labels=labels).sample
    got = m(x.cuda(), 1500, class_labels=labels.cuda()).sample
    assert rel(got, ref) < 5e-5


# models_configs/denoiser/orig_google_ddpm_model_denoiser.json on the HIP path: attention_head_dim = None (ONE head over 512
# channels -> pd_attn_wide), six levels, eps 1e-6, freq_shift 1 / sin before cos, pad-0 downsamplers, no class table
# (cond_unet_2d.py:132-153,176-197).  Same 2x-measured bounds as super_small.
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 2.5e-2), ("fp16", 3e-3)])
def test_unet_forward_orig_google_ddpm(mode, tol):
    import phendiff_amd as P
    from oracle import CondUNet2DRef, UNET_CONFIGS as REF_CONFIGS
    torch.manual_seed(0)
    r = CondUNet2DRef(**dict(REF_CONFIGS["orig_google_ddpm"], sample_size=64)).eval()
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS["orig_google_ddpm_model_denoiser"], sample_size=64))
    assert sum(p.numel() for p in m.parameters()) == 113_673_219
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    x, _ = synth_batch(2, 64)
    for t in (2999, 37):
        with torch.no_grad():
            ref = r(x, t).sample
        got = m(x.cuda(), t).sample
        assert got.shape == ref.shape and rel(got, ref) < tol, (mode, t, rel(got, ref))
    if mode == "f32":           # committed oracle vectors (tests/golden/make_golden.py --google): forward + DDIB round trip, S = 2
        d = np.load(os.path.join(GOLDEN, "ddib_google_ddpm_64_s2.npz"))
        xg, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
        assert rel(m(xg.cuda(), 1500).sample, d["unet_out_t1500"]) < tol
        pipe = P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
        graph = P.DDIBGraph(pipe, batch_size=2, num_inference_steps=2, height=64, width=64)
        out = graph.run(xg.cuda(), labels.cuda(), (1 - labels).cuda())
        torch.cuda.synchronize()
        assert rel(out.inverted, d["inverted"]) < 2e-5 and rel(out.images, d["out_images"]) < 2e-5


# models_configs/denoiser/SD_2-1_config.json -- the fourth shipped denoiser: pixel-space class-conditional UNet with
# SD-2.1 widths (320 / 640 / 1280 / 1280: GroupNorm groups of 10 / 20 / 40 channels), d = 8 attention on three levels + the mid block
# (40 / 80 / 160 heads), 641 914 883 parameters.  One oracle instance (17 s to initialise on the host) serves the three engines.
@pytest.fixture(scope="module")
def sd21_denoiser_oracle():
    from oracle import CondUNet2DRef, UNET_CONFIGS as REF_CONFIGS
    torch.manual_seed(0)                      # the seed / construction order of tests/golden/make_golden.py --sd21-denoiser
    return CondUNet2DRef(**dict(REF_CONFIGS["SD_2-1_config"], sample_size=32)).eval()


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 2.5e-2), ("fp16", 3e-3)])
def test_unet_forward_sd21_denoiser_config(sd21_denoiser_oracle, mode, tol):
    import warnings
    import phendiff_amd as P
    r = sd21_denoiser_oracle
    # from_config takes the file's key set: the six UNet2DConditionModel keys are dropped with a warning, as diffusers does
    file_cfg = dict(P.UNET_CONFIGS["SD_2-1_config"], **{k: None for k in P.configs.UNET_CONFIG_IGNORED_KEYS}, sample_size=32)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m = P.CustomCondUNet2DModel.from_config(file_cfg, compute_dtype=mode)
    assert any("will be ignored" in str(x.message) and "use_linear_projection" in str(x.message) for x in w)
    assert sum(p.numel() for p in m.parameters()) == 641_914_883
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    assert [len(b.attentions) for b in m.down_blocks if hasattr(b, "attentions")] == [2, 2, 2]
    for size, t in ((32, 2999), (64, 40)):
        x, labels = synth_batch(2, size)
        with torch.no_grad():
            ref = r(x, t, class_labels=labels).sample
        got = m(x.cuda(), t, class_labels=labels.cuda()).sample
        assert got.shape == ref.shape and rel(got, ref) < tol, (mode, size, rel(got, ref))
    # committed oracle vectors (tests/golden/make_golden.py --sd21-denoiser): one evaluation + a DDIB class transfer, S = 2, 32x32
    d = np.load(os.path.join(GOLDEN, "ddib_sd21_denoiser_32_s2.npz"))
    xg, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
    assert rel(m(xg.cuda(), 1500, class_labels=labels.cuda()).sample, d["unet_out_t1500"]) < tol
    pipe = P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    graph = P.DDIBGraph(pipe, batch_size=2, num_inference_steps=2, height=32, width=32)
    out = graph.run(xg.cuda(), labels.cuda(), (1 - labels).cuda())
    torch.cuda.synchronize()
    traj_tol = {"f32": 2e-5, "bf16": 1.5e-2, "fp16": 2e-3}[mode]
    assert rel(out.inverted, d["inverted"]) < traj_tol and rel(out.images, d["out_images"]) < traj_tol
    del m, graph, pipe
    torch.cuda.empty_cache()


def test_unet_rejects_bad_calls():
    import phendiff_amd as P
    _, m = make_pair("super_small", 32, "f32")
    x, labels = synth_batch(2, 32)
    with pytest.raises(ValueError):
        m(x.cuda(), 1, class_labels=labels.cuda(), class_emb=torch.zeros(2, 256, device="cuda"))
    with pytest.raises(ValueError):
        m(x.cuda(), 1)
    with pytest.raises(P.PhenDiffHipError):
        m(x, 1, class_labels=labels)   # CPU tensors: no fallback


def _pipes(mode, size=32):
    import phendiff_amd as P
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef
    r, m = make_pair("super_small", size, mode)
    cfg = P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]
    return ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**cfg)), P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**cfg))


# measured maxima (profiles/r2_parity_errors.json): f32 1.6e-6, bf16 7.2e-3, fp16 8.9e-4
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 1.5e-2), ("fp16", 2e-3)])
def test_ddib_eager_and_graph_vs_oracle(mode, tol):
    """invert (orig class) -> denoise (target = 1 - orig), S = 4, 32x32: utils_Img2Img._ddib semantics."""
    import phendiff_amd as P
    from oracle import ddib_ref, inversion_ref
    pref, pgot = _pipes(mode)
    x, labels = synth_batch(4, 32)
    target = P.swap_binary_labels(labels)
    ref_img, ref_inv = ddib_ref(pref, x, labels, target, 4)
    inv = P.inversion(pgot, x.cuda(), labels.cuda(), 4)
    assert rel(inv, ref_inv) < tol
    img = P.ddib(pgot, x.cuda(), labels.cuda(), target.cuda(), 4)
    assert img.shape == ref_img.shape == (4, 32, 32, 3) and img.dtype == np.float32
    assert rel(img, ref_img) < tol
    # hipGraph replay of the same trajectory: identical to the eager product path, bit for bit
    graph = P.DDIBGraph(pgot, batch_size=4, num_inference_steps=4)
    out = graph.run(x.cuda(), labels.cuda(), target.cuda())
    torch.cuda.synchronize()
    assert torch.equal(out.images.cpu(), torch.from_numpy(img))
    assert torch.equal(out.inverted.cpu(), inv.cpu())
    u8 = out.images_u8.cpu().numpy()
    # uint8 quantisation (<= 1 LSB is the exact-fp32 engine's contract; the 16-bit engines carry their trajectory
    # error of 7e-3 / 9e-4 into the image: measured worst pixel 8 / 1 LSB, bounds below)
    lsb = int(np.abs(u8.astype(int) - (ref_img * 255).round().astype(int)).max())
    from conftest import record_error
    record_error(float(lsb))
    assert lsb <= {"f32": 1, "fp16": 2, "bf16": 12}[mode], lsb
    # replay with other inputs reuses the graph
    x2, l2 = synth_batch(4, 32, seed=99)
    out2 = graph.run(x2.cuda(), l2.cuda(), (1 - l2).cuda())
    torch.cuda.synchronize()
    ref2, _ = ddib_ref(pref, x2, l2, 1 - l2, 4)
    assert rel(out2.images, ref2) < tol


def test_ddib_graph_slices_batches_beyond_one_plan(monkeypatch):
    """SURVEY 8(d) sweeps batch_size up to 128; one launch plan holds <= 127 images at 256 x 256 (32-bit byte offsets).  DDIBGraph replays a
    larger batch as even slices (equal slices share one captured graph): same result as the oracle, and per-sample identical to the
    trajectories of the slices run on their own.  The plan limit is lowered to 3 images here (7 = 3 + 2 + 2: two graphs, three replays)."""
    import phendiff_amd as P
    from oracle import ddib_ref
    pref, pgot = _pipes("f32")
    monkeypatch.setattr(type(pgot.unet), "max_batch", lambda self, H, W: 3)
    x, labels = synth_batch(7, 32, seed=5)
    target = 1 - labels
    g = P.DDIBGraph(pgot, batch_size=7, num_inference_steps=3)
    assert [b1 - b0 for b0, b1 in g._bounds] == [3, 3, 1] or [b1 - b0 for b0, b1 in g._bounds] == [3, 2, 2]
    out = g.run(x.cuda(), labels.cuda(), target.cuda())
    torch.cuda.synchronize()
    ref_img, ref_inv = ddib_ref(pref, x, labels, target, 3)
    assert rel(out.images, ref_img) < 2e-5 and rel(out.inverted, ref_inv) < 2e-5
    assert out.images_u8.shape == (7, 32, 32, 3) and int(np.abs(out.images_u8.cpu().numpy().astype(int) - (ref_img * 255).round().astype(int)).max()) <= 1
    b0, b1 = g._bounds[1]
    solo = P.DDIBGraph(pgot, batch_size=b1 - b0, num_inference_steps=3, private_plan=True).run(x[b0:b1].cuda(), labels[b0:b1].cuda(), target[b0:b1].cuda())
    torch.cuda.synchronize()
    assert torch.equal(solo.images, out.images[b0:b1]) and torch.equal(solo.inverted, out.inverted[b0:b1])
    with pytest.raises(ValueError):
        g.run(x[:6].cuda(), labels[:6].cuda(), target[:6].cuda())


def test_golden_fixture_ddib_f32():
    """Committed oracle vectors (tests/golden/make_golden.py): weights seed 0, super_small @32, S=4."""
    import phendiff_amd as P
    d = np.load(os.path.join(GOLDEN, "ddib_super_small_32_s4.npz"))
    _, pgot = _pipes("f32")
    x, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
    inv = P.inversion(pgot, x.cuda(), labels.cuda(), 4)
    assert rel(inv, d["inverted"]) < 2e-4
    img = P.ddib(pgot, x.cuda(), labels.cuda(), (1 - labels).cuda(), 4)
    assert rel(img, d["out_images"]) < 2e-4
    eps = pgot.unet(x.cuda(), 1500, class_labels=labels.cuda()).sample
    assert rel(eps, d["unet_out_t1500"]) < 5e-5


def test_pipeline_cfg_and_forward_noise_f32():
    """Pipeline options off the DDIB path: from-noise sampling with a generator, CFG (both equations, tensor w),
    forward-noising + frac_diffusion_skipped (classifier_free_guidance_forward_start semantics)."""
    pref, pgot = _pipes("f32")
    labels = torch.tensor([0, 1, 1])
    x, _ = synth_batch(3, 32)
    for kw in (dict(w=2.5, guidance_eqn="imagen"), dict(w=0.7, guidance_eqn="CFG"),
               dict(w=torch.tensor([1.5, 0.0, 3.0]), guidance_eqn="imagen")):
        ref = pref(class_labels=labels, num_inference_steps=3, start_image=x, add_forward_noise_to_image=False,
                   frac_diffusion_skipped=0.5, **kw).images
        kw2 = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
        got = pgot(class_labels=labels.cuda(), num_inference_steps=3, start_image=x.cuda(), add_forward_noise_to_image=False,
                   frac_diffusion_skipped=0.5, output_type="numpy", **kw2).images
        assert rel(got, ref) < 2e-4, kw
    # seeded CPU generator: same noise stream as the oracle (randn_tensor draws on the generator's device)
    ref = pref(class_labels=labels, num_inference_steps=2, generator=torch.Generator().manual_seed(5)).images
    got = pgot(class_labels=labels.cuda(), num_inference_steps=2, generator=torch.Generator().manual_seed(5),
               output_type="numpy").images
    assert rel(got, ref) < 2e-4
    ref = pref(class_labels=labels, num_inference_steps=4, start_image=x, frac_diffusion_skipped=0.5, w=2.5,
               generator=torch.Generator().manual_seed(6)).images
    got = pgot(class_labels=labels.cuda(), num_inference_steps=4, start_image=x.cuda(), frac_diffusion_skipped=0.5, w=2.5,
               generator=torch.Generator().manual_seed(6), output_type="pil").images
    assert len(got) == 3 and got[0].size == (32, 32)
    assert np.abs(np.asarray(got[0]).astype(int) - (ref[0] * 255).round().astype(int)).max() <= 1


def test_roundtrip_property_full_size():
    """Size-independent property at the BASELINE size (256x256): a zero-output model makes inversion then
    regeneration closed-form, so the engine's scheduler path can be checked without the CPU oracle."""
    import phendiff_amd as P
    m = P.CustomCondUNet2DModel(compute_dtype="bf16", **dict(P.UNET_CONFIGS["super_small"], sample_size=256))
    for p in m.parameters():
        p.data.zero_()
    m.to("cuda:0")
    pipe = P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
    x, labels = synth_batch(2, 256)
    g = P.DDIBGraph(pipe, batch_size=2, num_inference_steps=5)
    out = g.run(x.cuda(), labels.cuda(), (1 - labels).cuda())
    torch.cuda.synchronize()
    # v = 0: inversion x <- (sqrt(a'a) + sqrt((1-a')(1-a))) x per step; the last inverse step has a' = 0 => x = sqrt(1-a) x
    inv = P.DDIMInverseScheduler.from_config(pipe.scheduler.config)
    inv.set_timesteps(5)
    f = 1.0
    ref = x.clone()
    for t in inv.timesteps:
        sa, sb, sap, dirc, _ = inv.step_coefficients(t)
        x0 = (sa * ref).clamp(-1, 1)
        ref = sap * x0 + dirc * (sb * ref)
    assert torch.allclose(out.inverted.cpu(), ref, atol=1e-5)
    assert torch.isfinite(out.images).all() and float(out.images.min()) >= 0 and float(out.images.max()) <= 1


@pytest.mark.parametrize("eqn,w", [("imagen", 2.5), ("CFG", 1.5)])
def test_cfg_forward_start_eager_and_graph_vs_oracle(eqn, w):
    """classifier_free_guidance_forward_start (utils_Img2Img.py:615-648; guidance_scale 2.5,
    frac_diffusion_skipped 0.5 in the shipped config): eager pipeline and hipGraph form against the oracle."""
    import phendiff_amd as P
    pref, pgot = _pipes("f32")
    x, labels = synth_batch(4, 32)
    target = 1 - labels
    S, frac = 6, 0.5
    ref = pref(class_labels=target, w=w, num_inference_steps=S, start_image=x, frac_diffusion_skipped=frac,
               guidance_eqn=eqn, generator=torch.Generator().manual_seed(7)).images
    if eqn == "imagen":
        eager = P.classifier_free_guidance_forward_start(pgot, x.cuda(), target.cuda(), w, frac, S,
                                                         generator=torch.Generator().manual_seed(7))
        assert rel(eager, ref) < 2e-4
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(7))   # the draw the pipeline makes
    g = P.CFGForwardStartGraph(pgot, batch_size=4, num_inference_steps=S, guidance_scale=w, frac_diffusion_skipped=frac,
                               guidance_eqn=eqn)
    assert g.ts == [t for t in pgot.scheduler.timesteps.tolist() if t <= 1500] and len(g.ts) == 3
    out = g.run(x.cuda(), target.cuda(), noise.cuda())
    torch.cuda.synchronize()
    assert rel(out.images, ref) < 2e-4
    out2 = g.run(x.cuda(), labels.cuda(), noise.cuda())      # replay with other labels
    torch.cuda.synchronize()
    ref2 = pref(class_labels=labels, w=w, num_inference_steps=S, start_image=x, frac_diffusion_skipped=frac,
                guidance_eqn=eqn, generator=torch.Generator().manual_seed(7)).images
    assert rel(out2.images, ref2) < 2e-4


def test_inverted_regeneration_reconstructs():
    """"inverted_regeneration" (utils_Img2Img.py:374-384) = DDIB with target = original class; more steps reconstruct
    the input at least as well (trend of saved_figures/reco_err_*.png), checked on the engine itself."""
    import phendiff_amd as P
    _, pgot = _pipes("f32")
    x, labels = synth_batch(2, 32)
    want = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    errs = [float(np.abs(P.inverted_regeneration(pgot, x.cuda(), labels.cuda(), S) - want).mean()) for S in (2, 10)]
    assert errs[1] <= errs[0] * 1.05


def test_nonsquare_and_batch1_f32():
    """Edge shapes: batch 1, non-square sample_size (tuple, pipeline_conditionial_ddim.py:222-234), sizes that are not
    multiples of the conv tile (masked tiles)."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    cfg = dict(P.UNET_CONFIGS["super_small"], sample_size=(24, 40))
    r = CondUNet2DRef(**{k: v for k, v in cfg.items() if k in keys}).eval()
    m = P.CustomCondUNet2DModel(compute_dtype="f32", **cfg)
    m.load_state_dict(r.state_dict())
    m.to("cuda:0")
    g = torch.Generator().manual_seed(4)
    x = torch.rand(1, 3, 24, 40, generator=g) * 2 - 1
    lab = torch.tensor([1])
    with torch.no_grad():
        ref = r(x, 77, class_labels=lab).sample
    got = m(x.cuda(), 77, class_labels=lab.cuda()).sample
    assert rel(got, ref) < 5e-5
    # per-sample timesteps (training-style call, utils_training.py:522-527)
    x3 = torch.rand(3, 3, 24, 40, generator=g) * 2 - 1
    ts = torch.tensor([5, 1500, 2999])
    l3 = torch.tensor([0, 1, 0])
    with torch.no_grad():
        ref = r(x3, ts, class_labels=l3).sample
    got = m(x3.cuda(), ts.cuda(), class_labels=l3.cuda()).sample
    assert rel(got, ref) < 5e-5
    # from-noise generation on a tuple sample_size
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef
    sc = P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]
    pref, pgot = ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**sc)), P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**sc))
    a = pref(class_labels=l3, num_inference_steps=3, generator=torch.Generator().manual_seed(2)).images
    b = pgot(class_labels=l3.cuda(), num_inference_steps=3, generator=torch.Generator().manual_seed(2), output_type="numpy").images
    assert b.shape == (3, 24, 40, 3) and rel(b, a) < 2e-4


@pytest.mark.parametrize("sched", ["1k_epsilon_pred", "SD_orig_config", "better_SD_config"])
def test_ddib_graph_other_scheduler_configs_f32(sched):
    """The other shipped scheduler configs (epsilon prediction with zero-SNR table: x0 = (x - sqrt(b) out)/sqrt(a) with
    a -> 0; leading spacing + steps_offset + set_alpha_to_one=False; no clipping) through eager + graph DDIB."""
    import phendiff_amd as P
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef, ddib_ref
    r, m = make_pair("super_small", 32, "f32")
    cfg = P.SCHEDULER_CONFIGS[sched]
    pref, pgot = ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**cfg)), P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**cfg))
    x, labels = synth_batch(2, 32)
    ref_img, ref_inv = ddib_ref(pref, x, labels, 1 - labels, 5)
    out = P.DDIBGraph(pgot, batch_size=2, num_inference_steps=5).run(x.cuda(), labels.cuda(), (1 - labels).cuda())
    torch.cuda.synchronize()
    fin = torch.isfinite(torch.from_numpy(ref_img))
    assert torch.equal(torch.isfinite(out.images.cpu()), fin)            # the epsilon/zero-SNR hazard reproduces
    assert rel(out.inverted, ref_inv) < 5e-4
    assert rel(out.images.cpu()[fin], torch.from_numpy(ref_img)[fin]) < 5e-4


def test_scheduler_step_options_f32():
    """eta > 0 with explicit variance noise, use_clipped_model_output, sample prediction, linspace spacing."""
    import phendiff_amd as P
    from oracle import DDIMSchedulerRef
    g = torch.Generator().manual_seed(21)
    x, out, vn = (torch.randn(2, 3, 8, 8, generator=g) for _ in range(3))
    for pt in ("epsilon", "sample", "v_prediction"):
        cfg = dict(num_train_timesteps=1000, beta_schedule="linear", prediction_type=pt, timestep_spacing="linspace",
                   clip_sample=True, clip_sample_range=0.8)
        a, b = DDIMSchedulerRef(**cfg), P.DDIMScheduler(**cfg)
        a.set_timesteps(7); b.set_timesteps(7)
        assert torch.equal(a.timesteps, b.timesteps)
        for t in a.timesteps[1:4]:
            ra = a.step(out, t, x, eta=0.7, use_clipped_model_output=True, variance_noise=vn)
            rb = b.step(out.cuda(), t, x.cuda(), eta=0.7, use_clipped_model_output=True, variance_noise=vn.cuda())
            assert torch.allclose(rb.prev_sample.cpu(), ra.prev_sample, rtol=1e-5, atol=1e-5), (pt, int(t))
            assert torch.allclose(rb.pred_original_sample.cpu(), ra.pred_original_sample, rtol=1e-5, atol=1e-6)


def test_full_size_forward_vs_oracle():
    """BASELINE size (256x256, super_small): one UNet evaluation against the CPU oracle, both engine modes."""
    import os as _os
    torch.set_num_threads(min(16, len(_os.sched_getaffinity(0))))
    r, m32 = make_pair("super_small", 256, "f32")
    x, labels = synth_batch(1, 256)
    with torch.no_grad():
        ref = r(x, 1500, class_labels=labels).sample
    got = m32(x.cuda(), 1500, class_labels=labels.cuda()).sample
    assert rel(got, ref) < 2e-5
    # measured (profiles/r2_parity_errors.json): bf16 1.12e-2; fp16 about 1/8 of it
    for mode, tol in (("bf16", 2.5e-2), ("fp16", 3e-3)):
        _, m16 = make_pair("super_small", 256, mode)
        got16 = m16(x.cuda(), 1500, class_labels=labels.cuda()).sample
        assert rel(got16, ref) < tol, (mode, rel(got16, ref))


def test_full_size_trajectory_bf16_vs_f32_engine():
    """The metric's own workload (256x256, 50 + 50 DDIM steps): the bf16 (bench) engine against the exact-fp32 engine on the
    same weights and images -- the oracle takes minutes per image here, the fp32 engine is its stand-in (it agrees with the
    oracle to 5e-5 per forward, test above)."""
    import phendiff_amd as P
    outs = {}
    x, labels = synth_batch(2, 256)
    for mode in ("f32", "bf16", "fp16"):
        torch.manual_seed(0)
        unet = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS["super_small"], sample_size=256)).to("cuda:0")
        pipe = P.ConditionalDDIMPipeline(unet, P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]))
        g = P.DDIBGraph(pipe, batch_size=2, num_inference_steps=50)
        o = g.run(x.cuda(), labels.cuda(), (1 - labels).cuda())
        torch.cuda.synchronize()
        outs[mode] = (o.images.cpu().clone(), o.inverted.cpu().clone())
        del g, pipe, unet
        torch.cuda.empty_cache()
    # measured: bf16 4.1e-3 on both (profiles/r2_parity_errors.json); the bench engine is the bf16 one, so its bound is ~2.5x that
    for mode, tol in (("bf16", 1.0e-2), ("fp16", 2.0e-3)):
        assert torch.isfinite(outs[mode][0]).all()
        assert rel(outs[mode][1], outs["f32"][1]) < tol, mode         # inverted latents after 50 steps
        assert rel(outs[mode][0], outs["f32"][0]) < tol, mode         # images after 100 steps
        assert float(outs[mode][0].min()) >= 0.0 and float(outs[mode][0].max()) <= 1.0


@pytest.mark.skipif(bool(os.environ.get("PD_SKIP_LONG_TESTS")), reason="one minute of CPU oracle on 16 cores (numbers of the last run: DESIGN.md section 2)")
def test_full_size_full_length_trajectory_vs_oracle():
    """The metric's workload end to end against the CPU oracle itself: 256x256, super_small, 50 inversion + 50 denoising
    steps, one image -- the exact-fp32 engine (and the bf16 engine) against oracle `ddib_ref` on identical weights."""
    import time
    import phendiff_amd as P
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef, ddib_ref
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    r, m32 = make_pair("super_small", 256, "f32")
    x, labels = synth_batch(1, 256)
    cfg = P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]
    t0 = time.time()
    with torch.no_grad():
        ref, ref_inv = ddib_ref(ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**cfg)), x, labels, 1 - labels, 50)
    t_cpu = time.time() - t0
    ref = torch.as_tensor(ref)
    res, res_inv, lsb, lsb_mean = {}, {}, {}, {}
    # A20: the metric's product is an 8-bit image -- uint8 = round(255 * clamp(x / 2 + 0.5, 0, 1)) of the oracle's output
    ref_u8 = (ref.numpy() * 255).round().astype(np.int16)
    for mode in ("f32", "bf16", "fp16"):
        m = m32 if mode == "f32" else make_pair("super_small", 256, mode)[1]
        pipe = P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**cfg))
        out = P.DDIBGraph(pipe, batch_size=1, num_inference_steps=50).run(x.cuda(), labels.cuda(), (1 - labels).cuda())
        torch.cuda.synchronize()
        res[mode], res_inv[mode] = rel(out.images.cpu(), ref), rel(out.inverted.cpu(), ref_inv)
        d = np.abs(out.images_u8.cpu().numpy().astype(np.int16) - ref_u8)
        lsb[mode], lsb_mean[mode] = int(d.max()), float(d.mean())
        from conftest import record_error
        record_error(float(lsb[mode]))
    print(f"full-length oracle trajectory: {t_cpu:.1f} s on the CPU; rel-L2 of the final images: f32 engine {res['f32']:.2e}, "
          f"bf16 engine {res['bf16']:.2e}, fp16 engine {res['fp16']:.2e}; of the inverted latents after 50 steps: {res_inv['f32']:.2e} / "
          f"{res_inv['bf16']:.2e} / {res_inv['fp16']:.2e}; worst uint8 pixel (LSB) {lsb}, mean |d| {lsb_mean}")
    # measured: f32 6.1e-7 / 5.6e-7, bf16 3.4e-3 / 4.1e-3 (images / inverted latents)
    assert res["f32"] < 2e-5 and res_inv["f32"] < 2e-5 and res["bf16"] < 1.0e-2 and res_inv["bf16"] < 1.0e-2
    assert res["fp16"] < 1e-3 and res_inv["fp16"] < 1e-3      # measured 4.3e-4 / 5.0e-4
    # worst 8-bit pixel of the 65 536 x 3 after 50 + 50 steps: the bench engine's (bf16) contract at full
    # length, stated as measured x ~2 (bounds set from the round-3 run recorded in profiles/r3_parity_errors.json)
    assert lsb["f32"] <= 1 and lsb["bf16"] <= 6 and lsb["fp16"] <= 2, lsb      # measured 1 / 3 / 1


def test_forward_slices_batches_beyond_the_2gib_tensor_limit(monkeypatch):
    """A batch whose widest activation would pass 2 GiB runs in even slices (the kernels use 32-bit byte offsets); per-sample
    results do not depend on the slicing (bitwise)."""
    import phendiff_amd as P
    torch.manual_seed(0)
    m = P.CustomCondUNet2DModel(compute_dtype="bf16", **dict(P.UNET_CONFIGS["super_small"], sample_size=32)).to("cuda:0")
    assert m.max_batch(256, 256) == 127 and m.max_batch(32, 32) > 4096
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7, 3, 32, 32, generator=g).cuda()
    ts = torch.tensor([5, 900, 33, 2999, 1500, 7, 64]).cuda()
    labels = torch.tensor([0, 1, 1, 0, 1, 0, 0]).cuda()
    whole = m(x, ts, labels).sample.clone()
    monkeypatch.setattr(type(m), "max_batch", lambda self, H, W: 3)
    sliced = m(x, ts, labels).sample
    assert sliced.shape == whole.shape and torch.equal(sliced, whole)
    emb = torch.randn(7, m.time_embed_dim, generator=g).cuda()
    assert torch.equal(m(x, 400, class_emb=emb, return_dict=False)[0][4:], m(x[4:], 400, class_emb=emb[4:]).sample)
