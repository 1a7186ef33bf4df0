"""SD UNet2DConditionModel backward / training step on MI355X against torch.autograd over the CPU oracle (same seeded weights,
latents and class conditioning): what `_SD_prediction_wrapper` + `accelerator.backward(loss)` + clip + AdamW produce in the
reference (utils_training.py:459-496,415-454; BASELINE configs[3])."""
import pytest
import torch

from test_gpu_sd_unet import SMALL, TINY, WIDE, make_pair
from test_gpu_unet_backward import compare
from test_gpu_unet_ddib import rel

pytestmark = pytest.mark.gpu


def batch(B, size, seed=6):
    import phendiff_amd as P
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["SD_orig_config"])
    g = torch.Generator().manual_seed(seed)
    clean = torch.randn(B, 4, size, size, generator=g) * 0.8
    noise = torch.randn(B, 4, size, size, generator=g)
    ts = torch.tensor([850, 300, 12, 999, 0, 501][:B])
    labels = torch.arange(B) % 2
    acp = sched.alphas_cumprod[ts]
    sa, sb = (acp ** 0.5).view(-1, 1, 1, 1), ((1 - acp) ** 0.5).view(-1, 1, 1, 1)
    return sched, clean, noise, ts, labels, sa * clean + sb * noise, sa * noise - sb * clean


def oracle_grads(r, emb, noisy, ts, target, labels, unconditional=False):
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    params = dict(r.named_parameters())
    params["class_embedding.inner_module.weight"] = emb.inner_module.weight
    for p in params.values():
        p.requires_grad_(True)
        p.grad = None
    ehs = torch.zeros(noisy.shape[0], 77, emb.inner_module.weight.shape[1]) if unconditional else ehs_ref(emb(labels))
    out = r(noisy, ts, ehs).sample
    loss = torch.nn.functional.mse_loss(out, target)     # v_prediction: utils_training.py:428-431
    loss.backward()
    return loss.detach(), {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in params.items()}


@pytest.mark.parametrize("mode,per_tol,glob_tol", [("f32", 3e-4, 3e-5), ("bf16", 1e-1, 2.5e-2)])
@pytest.mark.parametrize("cfg,size", [(TINY, 16), (SMALL, 32), (WIDE, 16), (TINY, 64)])      # (TINY, 64): Upsample2D 32 -> 64 as sub-pixel phases, forward and input gradient
def test_sd_unet_backward_matches_autograd(mode, per_tol, glob_tol, cfg, size):
    import phendiff_amd as P
    r, emb, m, e2 = make_pair(cfg, mode)
    B = 3
    sched, clean, noise, ts, labels, noisy, target = batch(B, size)
    loss_ref, ref = oracle_grads(r, emb, noisy, ts, target, labels)
    tr = P.SDUNetTrainer(m, e2, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < (1e-5 if mode == "f32" else 5e-3) * float(loss_ref)
    compare(ref, tr.grads, per_tol, glob_tol)
    assert float(tr.grads["class_embedding.inner_module.weight"].abs().max()) > 0
    # gradients ACCUMULATE across calls
    tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    compare({n: 2 * g for n, g in ref.items()}, tr.grads, per_tol, glob_tol)


@pytest.mark.skipif(bool(__import__("os").environ.get("PD_SKIP_LONG_TESTS")), reason="866 M-parameter oracle + autograd on the CPU (~1 minute)")
def test_sd21_unet_full_size_backward_matches_autograd_bf16():
    """BASELINE configs[3]'s model at its real widths: gradients of all 686 parameters of the SD-2.1 UNet (865.9 M) + the
    CustomEmbedding table from the bf16 engine (the one the fine-tuning bench times) against torch.autograd over the CPU oracle,
    2 samples at 32x32 latents."""
    import os
    import phendiff_amd as P
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    r, emb, m, e2 = make_pair(P.SD21_UNET_CONFIG, "bf16")
    B = 2
    sched, clean, noise, ts, labels, noisy, target = batch(B, 32)
    loss_ref, ref = oracle_grads(r, emb, noisy, ts, target, labels)
    tr = P.SDUNetTrainer(m, e2, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert len(ref) == 687
    assert abs(float(loss) - float(loss_ref)) < 5e-3 * float(loss_ref)
    compare(ref, tr.grads, 1e-1, 2.5e-2)


def test_sd_unet_backward_unconditional_step_f32():
    """All-zero context (utils_training.py:465-471): the class table and the cross-attention key / value projections get no
    gradient; everything else does."""
    import phendiff_amd as P
    r, emb, m, e2 = make_pair(TINY, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 16)
    _, ref = oracle_grads(r, emb, noisy, ts, target, labels, unconditional=True)
    tr = P.SDUNetTrainer(m, e2, sched, lr=1e-4, use_ema=False)
    tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda(), unconditional=True)
    torch.cuda.synchronize()
    assert float(tr.grads["class_embedding.inner_module.weight"].abs().max()) == 0.0
    compare(ref, tr.grads, 3e-4, 3e-5)


def test_sd_training_steps_follow_torch_adamw_f32():
    """Three optimisation steps (conditional, unconditional, conditional): clip_grad_norm_ 1.0 -> AdamW -> re-packed weights
    track the same steps done by torch on the oracle, including the trained CustomEmbedding."""
    import phendiff_amd as P
    r, emb, m, e2 = make_pair(TINY, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(4, 16)
    tr = P.SDUNetTrainer(m, e2, sched, lr=2e-4, use_ema=True)
    allp = list(r.parameters()) + list(emb.parameters())
    opt = torch.optim.AdamW(allp, lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    for uncond in (False, True, False):
        loss_ref, _ = oracle_grads(r, emb, noisy, ts, target, labels, unconditional=uncond)
        torch.nn.utils.clip_grad_norm_(allp, 1.0)
        opt.step()
        loss = tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), labels.cuda(), unconditional=uncond)
        assert abs(float(loss) - float(loss_ref)) < 2e-4 * abs(float(loss_ref))
    torch.cuda.synchronize()
    sd = dict(r.state_dict())
    num = den = 0.0
    for n, p in m.named_parameters():
        num += float((p.detach().cpu() - sd[n]).double().pow(2).sum())
        den += float(sd[n].double().pow(2).sum())
    assert (num / den) ** 0.5 < 1e-5
    assert rel(e2.inner_module.weight.detach(), emb.inner_module.weight.detach()) < 1e-5
    # the inference entry point sees the updated weights (shared, re-packed in place)
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    with torch.no_grad():
        ref_out = r(noisy, ts, ehs_ref(emb(labels))).sample
    got = m(noisy.cuda(), ts.cuda(), P.class_emb_to_encoder_hidden_states(e2(labels.cuda()))).sample
    assert rel(got, ref_out) < 1e-4


def test_sd_inference_after_training_steps_sees_the_updated_upsampler_phase_kernels():
    """As above at 64 x 64 latents, where the inference plan runs Upsample2D (32 -> 64) as four sub-pixel phases with PRE-SUMMED copies of
    the 3x3 weights: the fine-tuning re-pack refreshes those copies as well."""
    import phendiff_amd as P
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    r, emb, m, e2 = make_pair(TINY, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 64)
    tr = P.SDUNetTrainer(m, e2, sched, lr=2e-3, use_ema=False)
    up = [n for n, _ in m.named_parameters() if ".upsamplers." in n and n.endswith("conv.weight")]
    before = {n: p.detach().clone() for n, p in m.named_parameters() if n in up}
    for _ in range(2):
        tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), labels.cuda())
    torch.cuda.synchronize()
    assert up and all(float((p.detach() - before[n]).abs().max()) > 1e-4 for n, p in m.named_parameters() if n in before)
    r.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    emb.inner_module.weight.data.copy_(e2.inner_module.weight.detach().cpu())
    with torch.no_grad():
        ref_out = r(noisy, ts, ehs_ref(emb(labels))).sample
    got = m(noisy.cuda(), ts.cuda(), P.class_emb_to_encoder_hidden_states(e2(labels.cuda()))).sample
    assert rel(got, ref_out) < 1e-4


def test_sd_training_step_bf16_reduces_loss_and_overlapped_path():
    import phendiff_amd as P
    _, _, m, e2 = make_pair(TINY, "bf16")
    sched, clean, noise, ts, labels, noisy, _ = batch(4, 16)
    tr = P.SDUNetTrainer(m, e2, sched, lr=5e-4)
    losses = [float(tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), labels.cuda())) for _ in range(8)]
    assert losses[-1] < losses[0], losses
    # bucket schedule of the overlapped all-reduce: every parameter has a completion point, the class embedding is last
    plan = tr.plan_for(4, 16, 16)
    names = list(tr.grads)
    assert all(n in plan.grad_ready for n in names)
    assert plan.grad_ready["class_embedding.inner_module.weight"] == len(plan.bwd_ops) - 1
    # every completion point is a launch that writes into the flat gradient buffer (not a helper launch emitted before the writer)
    lo, hi = tr.opt.grad.data_ptr(), tr.opt.grad.data_ptr() + tr.opt.grad.numel() * 4
    for name, idx in plan.grad_ready.items():
        op = plan.bwd_ops[idx]
        ptrs = [getattr(op.args, f) for f, _ in op.args._fields_]
        assert any(isinstance(v, int) and lo <= v < hi for v in ptrs), (name, idx, op.what)


def test_sd_save_state_resume_continues_bitwise(tmp_path):
    """accelerate-layout checkpoint of a StableDiffusion run (unet = pytorch_model.bin, class_embedding = pytorch_model_2.bin, one
    optimizer over both, utils_misc.py:322-347): the CustomEmbedding's weights, Adam moments, its OWN step count (it is skipped on
    unconditional steps) and EMA shadow survive, so a resumed run reproduces the uninterrupted one exactly."""
    import os
    import phendiff_amd as P
    B, size = 2, 16
    sched, clean, noise, ts, labels, noisy, _ = batch(B, size)
    args = [t.cuda() for t in (noisy, ts, clean, noise, labels)]

    def fresh():
        _, _, m, e2 = make_pair(TINY, "bf16")
        return P.SDUNetTrainer(m, e2, sched, lr=3e-4)
    a = fresh()
    uncond = [False, True, False]                      # step 2 leaves the embedding without a gradient
    for u in uncond:
        a.step(*args, unconditional=u)
    assert a.opt.t == 3 and a.opt.t_tail == 2
    folder = str(tmp_path / "step_3")
    a.save_state(folder)
    assert sorted(os.listdir(folder)) == ["custom_checkpoint_0.pkl", "custom_checkpoint_1.pkl", "optimizer.bin", "pytorch_model.bin",
                                          "pytorch_model_2.bin", "random_states_0.pkl", "scheduler.bin"]
    osd = torch.load(os.path.join(folder, "optimizer.bin"))
    n_unet = sum(1 for _ in a.model.parameters())
    assert len(osd["param_groups"][0]["params"]) == n_unet + 1 and float(osd["state"][n_unet]["step"]) == 2.0
    rest = [float(a.step(*args, unconditional=u)) for u in (False, True, False)]
    b = fresh()
    with torch.no_grad():                               # a resumed run starts from different weights: the checkpoint must win
        b.class_embedding.inner_module.weight.data.add_(1.0)
    b.load_state(folder)
    assert b.opt.t == 3 and b.opt.t_tail == 2
    resumed = [float(b.step(*args, unconditional=u)) for u in (False, True, False)]
    assert resumed == rest, (rest, resumed)
    assert torch.equal(b.opt.ema, a.opt.ema) and torch.equal(b.opt.flat, a.opt.flat)
    assert torch.equal(b.opt.exp_avg, a.opt.exp_avg) and torch.equal(b.opt.exp_avg_sq, a.opt.exp_avg_sq)


# ---- round 5: fp16 fine-tuning behind --mixed_precision fp16 (args_parser.py:381-390), the latent-diffusion trainer ----
@pytest.mark.parametrize("cfg,size", [(TINY, 16), (SMALL, 32)])
def test_sd_fp16_backward_matches_autograd_under_the_loss_scale(cfg, size):
    """fp16 activations / activation gradients through the transformer blocks' backward set (pd_attn_d64_bwd, pd_layernorm_bwd,
    pd_geglu_bwd, pd_token_wgrad, pd_token_embedding_grad in their f16 forms), fp32 parameter gradients carrying the GradScaler's
    scale: gradients / scale against torch.autograd over the fp32 oracle at fp16's tolerance."""
    import phendiff_amd as P
    r, emb, m, e2 = make_pair(cfg, "fp16")
    sched, clean, noise, ts, labels, noisy, target = batch(3, size)
    loss_ref, ref = oracle_grads(r, emb, noisy, ts, target, labels)
    tr = P.SDUNetTrainer(m, e2, sched, lr=1e-4, use_ema=False)
    assert tr.opt.scaler is not None and tr.opt.scaler.scale == 65536.0
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * float(loss_ref)
    got = {n: g / tr.opt.scaler.scale for n, g in tr.grads.items()}
    assert all(torch.isfinite(g).all() for g in got.values())
    compare(ref, got, 4e-2, 8e-3)


def test_sd_fp16_training_steps_reduce_the_loss_and_skip_on_overflow():
    import phendiff_amd as P
    _, _, m, e2 = make_pair(TINY, "fp16")
    sched, clean, noise, ts, labels, noisy, target = batch(4, 16)
    tr = P.SDUNetTrainer(m, e2, sched, lr=5e-4, use_ema=True)
    args = (noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), labels.cuda())
    tr.opt.scaler.scale = 2.0 ** 40
    before = tr.opt.flat.clone()
    tr.step(*args)
    torch.cuda.synchronize()
    assert tr.opt.scaler.scale == 2.0 ** 39 and tr.opt.t == 0 and torch.equal(tr.opt.flat, before)
    tr.opt.scaler.scale = 65536.0
    losses = [float(tr.step(*args, unconditional=(k == 3))) for k in range(8)]
    torch.cuda.synchronize()
    assert tr.opt.t == 8 and tr.opt.scaler.skipped == 1 and torch.isfinite(tr.opt.flat).all()
    assert min(losses[-3:]) < losses[0]
