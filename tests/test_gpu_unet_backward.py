"""UNet backward / training step on MI355X against torch.autograd over the CPU oracle (same seeded weights and batch):
what ``accelerator.backward(loss)`` + ``clip_grad_norm_`` + ``optimizer.step()`` produce in the reference
(utils_training.py:415-454)."""
import pytest
import torch

from test_gpu_unet_ddib import make_pair, rel

pytestmark = pytest.mark.gpu


def batch(B, size, seed=5):
    import phendiff_amd as P
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    g = torch.Generator().manual_seed(seed)
    clean = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    noise = torch.randn(B, 3, size, size, generator=g)
    ts = torch.tensor([2500, 700, 40, 1500, 2999, 0][:B])
    labels = torch.arange(B) % 2
    acp = sched.alphas_cumprod[ts]
    sa, sb = (acp ** 0.5).view(-1, 1, 1, 1), ((1 - acp) ** 0.5).view(-1, 1, 1, 1)
    return sched, clean, noise, ts, labels, sa * clean + sb * noise, sa * noise - sb * clean


def oracle_grads(r, noisy, ts, target, labels=None, class_emb=None):
    for p in r.parameters():
        p.requires_grad_(True)
        p.grad = None
    out = r(noisy, ts, class_labels=labels, class_emb=class_emb).sample
    loss = torch.nn.functional.mse_loss(out, target)     # v_prediction: utils_training.py:428-431
    loss.backward()
    return loss.detach(), {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in r.named_parameters()}


def compare(ref, got, per_param_tol, global_tol):
    gnorm = sum(float(g.double().pow(2).sum()) for g in ref.values()) ** 0.5
    num = 0.0
    for n, gr in ref.items():
        d = got[n].cpu() - gr
        num += float(d.double().pow(2).sum())
        # gradients that are mathematically zero (softmax is invariant to the key bias) hold only round-off
        if float(gr.norm()) > 1e-6 * gnorm:
            assert float(d.norm() / gr.norm()) < per_param_tol, (n, float(d.norm() / gr.norm()))
        else:
            assert float(d.norm()) < 1e-5 * gnorm, n
    assert num ** 0.5 / gnorm < global_tol


@pytest.mark.parametrize("mode,per_tol,glob_tol", [("f32", 2e-4, 2e-5), ("bf16", 8e-2, 2e-2)])
@pytest.mark.parametrize("size", [32, 64, 128])          # 128 = BASELINE configs[1]'s image size
def test_unet_backward_matches_autograd(mode, per_tol, glob_tol, size):
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", size, mode)
    B = 3
    sched, clean, noise, ts, labels, noisy, target = batch(B, size)
    loss_ref, ref = oracle_grads(r, noisy, ts, target, labels=labels)
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < (1e-5 if mode == "f32" else 5e-3) * float(loss_ref)
    compare(ref, tr.grads, per_tol, glob_tol)
    # gradients ACCUMULATE across calls (gradient accumulation; the optimizer zeroes them)
    tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    compare({n: 2 * g for n, g in ref.items()}, tr.grads, per_tol, glob_tol)


@pytest.mark.parametrize("mode,per_tol,glob_tol", [("f32", 2e-4, 2e-5), ("bf16", 8e-2, 2e-2)])
def test_centered_input_and_identity_class_rows_train(mode, per_tol, glob_tol):
    """Two config switches no shipped JSON sets but `CustomCondUNet2DModel` accepts (cond_unet_2d.py:146-153,272-273), in the training
    plan: center_input_sample (conv_in multiplies 2 x - 1: its weight gradient reads the centred tensor, the input gradient is doubled)
    and class_embed_type = "identity" (the "labels" are embedding rows: an input, no class parameter).  Parameter gradients and
    d loss / d sample against torch.autograd over the oracle."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef
    from phendiff_amd.unet_train import UNetTrainer
    torch.manual_seed(0)
    cfg = dict(P.UNET_CONFIGS["super_small"], sample_size=32, center_input_sample=True, class_embed_type="identity", num_class_embeds=None)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in cfg.items() if k in keys}).eval()
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **cfg)
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    sched, clean, noise, ts, _, noisy, target = batch(3, 32)
    rows = torch.randn(3, m.time_embed_dim, generator=torch.Generator().manual_seed(9))
    x = noisy.clone().requires_grad_(True)
    for p in r.parameters():
        p.requires_grad_(True)
        p.grad = None
    out = r(x, ts, class_labels=rows).sample
    loss_ref = torch.nn.functional.mse_loss(out, target)
    loss_ref.backward()
    ref = {n: p.grad.clone() for n, p in r.named_parameters()}
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=rows.cuda())
    torch.cuda.synchronize()
    assert set(ref) == set(tr.grads) and abs(float(loss) - float(loss_ref.detach())) < (1e-5 if mode == "f32" else 5e-3) * float(loss_ref.detach())
    compare(ref, tr.grads, per_tol, glob_tol)
    # the input gradient (what the gradient-guided transfer differentiates): d loss / d sample = 2 x d loss / d (2 x - 1)
    plan = m.input_grad_plan(3, 32, 32, torch.device("cuda:0"))
    st = torch.cuda.current_stream().cuda_stream
    o = torch.empty(3, 3, 32, 32, device="cuda:0")
    plan.forward(noisy.cuda().contiguous(), ts.cuda().float(), None, rows.cuda().contiguous(), o, st)
    dout = (2.0 / o.numel()) * (o - target.cuda())
    plan.backward(dout.contiguous(), st)
    torch.cuda.synchronize()
    assert rel(plan.dsample, x.grad) < (2e-5 if mode == "f32" else 4e-2)


BWD_VARIANTS = [dict(resnet_time_scale_shift="scale_shift"), dict(class_embed_type="timestep"),
                dict(center_input_sample=True, resnet_time_scale_shift="scale_shift", class_embed_type="timestep")]


@pytest.mark.parametrize("mode,per_tol,glob_tol", [("f32", 2e-4, 2e-5), ("bf16", 8e-2, 2e-2), ("fp16", 3e-2, 6e-3)])
@pytest.mark.parametrize("variant", BWD_VARIANTS, ids=lambda v: "+".join(sorted(v)))
def test_scale_shift_resnets_and_the_timestep_class_mlp_train(mode, per_tol, glob_tol, variant):
    """Round 6: the two constructor switches whose backward was refused until now (cond_unet_2d.py:146-153,180,191,225).
    ``resnet_time_scale_shift = "scale_shift"``: h = norm2(h1) (1 + scale) + shift -- the GroupNorm backward works on the per-sample
    affine and returns d [scale | shift] (pd_gn_bwd_args.mod / dmod), which feeds the stacked time_emb_proj gradient.
    ``class_embed_type = "timestep"``: emb += class_embedding(time_proj(labels)) -- the class MLP gets the time-embedding MLP's three
    gradient launches on d emb; a step whose rows bypass it (class_emb given: the unconditional steps) leaves its gradients at zero.
    Parameter gradients, d loss / d sample, and three optimisation steps against torch on the oracle."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef
    from phendiff_amd.unet_train import UNetTrainer
    torch.manual_seed(3)
    cfg = dict(P.UNET_CONFIGS["super_small"], sample_size=32, **variant)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in cfg.items() if k in keys}).eval()
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **cfg)
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    sched, clean, noise, ts, labels, noisy, target = batch(3, 32)
    x = noisy.clone().requires_grad_(True)
    for p in r.parameters():
        p.requires_grad_(True)
        p.grad = None
    out = r(x, ts, class_labels=labels).sample
    loss_ref = torch.nn.functional.mse_loss(out, target)
    loss_ref.backward()
    ref = {n: p.grad.clone() for n, p in r.named_parameters()}
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    scale = tr.opt.scaler.scale if tr.opt.scaler is not None else 1.0
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert set(ref) == set(tr.grads)
    assert abs(float(loss) - float(loss_ref.detach())) < {"f32": 1e-5, "bf16": 5e-3, "fp16": 2e-3}[mode] * float(loss_ref.detach())
    compare(ref, {n: g / scale for n, g in tr.grads.items()}, per_tol, glob_tol)
    if variant.get("class_embed_type") == "timestep":
        # rows given directly (an unconditional step's zeros, utils_training.py:507-515): the class MLP is bypassed and gets no gradient.
        # (The reference itself cannot take this path with class_embed_type = "timestep" -- cond_unet_2d.py:302 calls
        # time_proj(class_labels) on None -- so there is no oracle value: the engine's superset is checked for what it must not do.)
        tr.opt.grad.zero_()
        zeros = torch.zeros(3, m.time_embed_dim)
        tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_emb=zeros.cuda())
        torch.cuda.synchronize()
        assert all(float(g.abs().max()) == 0.0 for n, g in tr.grads.items() if n.startswith("class_embedding."))
        assert all(torch.isfinite(g).all() for g in tr.grads.values()) and float(tr.grads["conv_in.weight"].abs().max()) > 0
        assert tr.opt.tail_names == tuple(n for n in tr.grads if n.startswith("class_embedding.")) and len(tr.opt.tail_names) == 4
    # d loss / d sample through the same plan (the gradient-guided transfer's gradient)
    if mode != "fp16":
        plan = m.input_grad_plan(3, 32, 32, torch.device("cuda:0"))
        st = torch.cuda.current_stream().cuda_stream
        o = torch.empty(3, 3, 32, 32, device="cuda:0")
        plan.forward(noisy.cuda().contiguous(), ts.cuda().float(), labels.cuda(), None, o, st)
        dout = (2.0 / o.numel()) * (o - target.cuda())
        plan.backward(dout.contiguous(), st)
        torch.cuda.synchronize()
        assert rel(plan.dsample, x.grad) < (2e-5 if mode == "f32" else 4e-2)
    if mode == "f32":
        # three optimisation steps track torch.optim.AdamW on the oracle (the re-pack refreshes the class MLP's transposed weights)
        tr2 = UNetTrainer(m, sched, lr=2e-4, use_ema=False)
        opt = torch.optim.AdamW(r.parameters(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
        for _ in range(3):
            lr_, _ = oracle_grads(r, noisy, ts, target, labels=labels)
            torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
            opt.step()
            lg = float(tr2.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda()))
            assert abs(lg - float(lr_)) < 2e-4 * abs(float(lr_))
        torch.cuda.synchronize()
        sd = r.state_dict()
        num = sum(float((p.detach().cpu() - sd[n]).double().pow(2).sum()) for n, p in m.named_parameters())
        den = sum(float(sd[n].double().pow(2).sum()) for n, _ in m.named_parameters())
        assert (num / den) ** 0.5 < 1e-5


def test_inference_after_training_steps_sees_the_updated_upsampler_phase_kernels():
    """The sub-pixel upsamplers multiply with PRE-SUMMED copies of the 3x3 weights (four 2x2 phase kernels): the re-pack after every
    optimizer step has to refresh them too -- an inference forward of the trained model (train.py's evaluation generation runs in the same
    process) against the oracle holding the trained state_dict, at a size where the sub-pixel form is in use (64 x 64: 32 -> 64)."""
    from phendiff_amd.unet_train import UNetTrainer
    from test_gpu_unet_ddib import synth_batch
    r, m = make_pair("super_small", 64, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 64)
    tr = UNetTrainer(m, sched, lr=2e-3, use_ema=False)
    before = {n: p.detach().clone() for n, p in m.named_parameters() if ".upsamplers." in n and n.endswith("conv.weight")}
    for _ in range(2):
        tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert all(float((p.detach() - before[n]).abs().max()) > 1e-4 for n, p in m.named_parameters() if n in before)      # the step moved them
    r.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    x, lb = synth_batch(2, 64)
    with torch.no_grad():
        ref = r(x, 700, class_labels=lb).sample
    out = m(x.cuda(), 700, lb.cuda()).sample
    assert rel(out, ref) < 2e-5


def test_input_gradient_after_training_steps_sees_the_updated_gradient_layout_weights():
    """`input_grad_plan` (the gradient-guided transfer) caches the transposed / flipped input-gradient weights on the model;
    a trainer's re-pack after every optimizer step has to refresh THAT set too -- d out / d sample of the trained model, through a plan
    built BEFORE the training steps and through one built after, against autograd over the oracle holding the trained state_dict."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 32)
    dev = torch.device("cuda:0")
    early = m.input_grad_plan(2, 32, 32, dev)            # built on the untrained weights
    tr = UNetTrainer(m, sched, lr=2e-3, use_ema=False)
    for _ in range(2):
        tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    r.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    x = noisy.clone().requires_grad_(True)
    out_ref = r(x, ts, class_labels=labels).sample
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(out_ref.shape, generator=g)
    (gx,) = torch.autograd.grad((out_ref * dout).sum(), x)
    plan = m.input_grad_plan(2, 32, 32, dev)
    assert plan is not early                              # the stale plan was dropped when the trainer bound its gradient-layout set
    st = torch.cuda.current_stream().cuda_stream
    o = torch.empty(2, 3, 32, 32, device=dev)
    plan.forward(noisy.cuda().contiguous(), ts.cuda().float(), labels.cuda(), None, o, st)
    plan.backward(dout.cuda().contiguous(), st)
    torch.cuda.synchronize()
    assert rel(o, out_ref.detach()) < 2e-5 and rel(plan.dsample, gx) < 2e-5


# (bf16 per-parameter bound: the worst parameters are attention to_q / to_k weights of the 4x4 / 2x2 levels whose gradients carry 6e-5 of the
#  global norm -- measured 0.079 with the 3x3 upsampler form, 0.082 with the sub-pixel form of round 4; global error 5.1e-3 either way)
@pytest.mark.parametrize("mode,per_tol,glob_tol", [("f32", 2e-4, 2e-5), ("bf16", 1.2e-1, 2e-2)])
def test_orig_google_ddpm_backward_matches_autograd(mode, per_tol, glob_tol):
    """models_configs/denoiser/orig_google_ddpm_model_denoiser.json trains too (VERDICT r2 missing 3: the reference trains whatever
    config it loads, utils_models.py:158-182, train.py:180-182): gradients of all its parameters at 64x64 against torch.autograd --
    one 512-channel attention head (pd_attn_wide_bwd), Downsample2D(padding=0) (odd-phase zero-stuffed input gradient), six
    levels down to 2x2, no class table."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef, UNET_CONFIGS as REF_CONFIGS
    from phendiff_amd.unet_train import UNetTrainer
    torch.manual_seed(0)
    r = CondUNet2DRef(**dict(REF_CONFIGS["orig_google_ddpm"], sample_size=64)).eval()
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **dict(P.UNET_CONFIGS["orig_google_ddpm_model_denoiser"], sample_size=64))
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 64)
    loss_ref, ref = oracle_grads(r, noisy, ts, target)
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda())
    torch.cuda.synchronize()
    assert len(ref) == len(tr.grads) and abs(float(loss) - float(loss_ref)) < (1e-5 if mode == "f32" else 5e-3) * float(loss_ref)
    compare(ref, tr.grads, per_tol, glob_tol)
    if mode == "f32":       # and it optimises: three AdamW steps track torch's on the oracle
        opt = torch.optim.AdamW(r.parameters(), lr=1e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
        tr.opt.grad.zero_()
        for _ in range(3):
            l_ref, _ = oracle_grads(r, noisy, ts, target)
            torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
            opt.step()
            l = tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda())
            assert abs(float(l) - float(l_ref)) < 2e-4 * abs(float(l_ref))


def test_sd21_denoiser_config_backward_matches_autograd_f32():
    """models_configs/denoiser/SD_2-1_config.json (641.9 M parameters) trains as well: d = 8 attention with 40 / 80 / 160 heads on
    three levels, GroupNorm groups of 10 / 20 / 40 channels, 2 560-channel concatenations -- gradients of all parameters at 32x32
    from the exact-fp32 engine against torch.autograd over the CPU oracle."""
    import phendiff_amd as P
    from oracle import CondUNet2DRef, UNET_CONFIGS as REF_CONFIGS
    from phendiff_amd.unet_train import UNetTrainer
    torch.manual_seed(0)
    r = CondUNet2DRef(**dict(REF_CONFIGS["SD_2-1_config"], sample_size=32)).eval()
    m = P.CustomCondUNet2DModel(compute_dtype="f32", **dict(P.UNET_CONFIGS["SD_2-1_config"], sample_size=32))
    m.load_state_dict(r.state_dict())
    m = m.to("cuda:0")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 32)
    loss_ref, ref = oracle_grads(r, noisy, ts, target, labels=labels)
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 1e-5 * float(loss_ref)
    compare(ref, tr.grads, 2e-4, 2e-5)
    del tr, m
    torch.cuda.empty_cache()


def test_unet_backward_unconditional_step_f32():
    """class_emb = zeros (the reference's unconditional training step, utils_training.py:398-407): the class table gets
    no gradient, everything else does."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(2, 32)
    zeros = torch.zeros(2, 256)
    _, ref = oracle_grads(r, noisy, ts, target, class_emb=zeros)
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_emb=zeros.cuda())
    torch.cuda.synchronize()
    assert float(tr.grads["class_embedding.weight"].abs().max()) == 0.0
    compare(ref, tr.grads, 2e-4, 2e-5)


def test_training_steps_follow_torch_adamw_f32():
    """Three optimisation steps (clip_grad_norm_ 1.0 -> AdamW(betas .95/.999, wd 1e-6) -> re-packed kernel weights) track
    the same steps done by torch on the oracle."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    tr = UNetTrainer(m, sched, lr=2e-4, use_ema=True)
    opt = torch.optim.AdamW(r.parameters(), lr=2e-4, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    losses_ref, losses = [], []
    for _ in range(3):
        loss_ref, _ = oracle_grads(r, noisy, ts, target, labels=labels)
        torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
        opt.step()
        losses_ref.append(float(loss_ref))
        losses.append(float(tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())))
    torch.cuda.synchronize()
    assert losses_ref[-1] < losses_ref[0]
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) < 2e-4 * abs(b), (losses, losses_ref)
    sd = r.state_dict()
    num = den = 0.0
    for n, p in m.named_parameters():
        num += float((p.detach().cpu() - sd[n]).double().pow(2).sum())
        den += float((sd[n] - 0).double().pow(2).sum())
    assert (num / den) ** 0.5 < 1e-5
    # the inference entry point sees the updated weights (shared, re-packed in place)
    with torch.no_grad():
        ref_out = r(noisy, ts, class_labels=labels).sample
    got = m(noisy.cuda(), ts.cuda(), class_labels=labels.cuda()).sample
    assert rel(got, ref_out) < 1e-4


def test_training_step_bf16_reduces_loss():
    from phendiff_amd.unet_train import UNetTrainer
    _, m = make_pair("super_small", 32, "bf16")
    sched, clean, noise, ts, labels, noisy, _ = batch(4, 32)
    tr = UNetTrainer(m, sched, lr=5e-4)
    losses = [float(tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < 0.8 * losses[0], losses


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_device_repack_equals_host_packing(mode):
    """After an optimizer step the pd_pack_weight path must leave exactly what packing.py builds from the new parameters."""
    from phendiff_amd.unet import _PackedWeights
    from phendiff_amd.unet_train import TrainWeights, UNetTrainer
    _, m = make_pair("super_small", 32, mode)
    sched, clean, noise, ts, labels, noisy, _ = batch(2, 32)
    tr = UNetTrainer(m, sched, lr=1e-3)
    tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    fresh_w, fresh_t = _PackedWeights(m, "cuda:0"), TrainWeights(m, "cuda:0", m._weights.tdt)

    def same(a, b, path):
        items = b.items() if isinstance(b, dict) else vars(b).items()
        for k, v in items:
            d = a[k] if isinstance(a, dict) else getattr(a, k)
            if torch.is_tensor(v):
                assert torch.equal(d, v), f"{path}.{k}"
            elif isinstance(v, (dict,)) or hasattr(v, "__dict__") and not isinstance(v, (str, torch.dtype, torch.device)):
                same(d, v, f"{path}.{k}")
            elif isinstance(v, tuple):
                for i, (dd, vv) in enumerate(zip(d, v)):
                    if torch.is_tensor(vv):
                        assert torch.equal(dd, vv), f"{path}.{k}[{i}]"
    same(m._weights, fresh_w, "w")
    same(tr._tw, fresh_t, "tw")


def test_overlapped_gradient_allreduce_path_single_rank():
    """The bucketed all-reduce issued from inside the backward (second stream, event-ordered) must leave the same gradients
    as the plain path.  One rank is all a 1-GPU box offers: the collective is an identity, the stream/event/bucket logic is
    what runs (multi-rank averaging itself: tests/test_training_blocks.py, 2-rank gloo)."""
    import os
    import torch.distributed as dist
    from phendiff_amd.unet_train import UNetTrainer
    _, m = make_pair("super_small", 32, "f32")
    sched, clean, noise, ts, labels, noisy, _ = batch(2, 32)
    args = [t.cuda() for t in (noisy, ts, clean, noise)]
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    tr.forward_backward(*args, class_labels=labels.cuda())
    torch.cuda.synchronize()
    want = tr.opt.grad.clone()
    tr.opt.grad.zero_()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        plan = tr.plan_for(2, 32, 32)
        assert set(plan.grad_ready) == set(tr.grads)                      # every parameter has a completion point
        assert plan.grad_ready["conv_out.weight"] < plan.grad_ready["mid_block.resnets.0.conv1.weight"] \
            < plan.grad_ready["conv_in.weight"] <= plan.grad_ready["time_embedding.linear_1.weight"]
        # ... and that point IS a launch that writes into the flat gradient buffer (a helper launch emitted between _G() and the writer --
        # the GroupNorm pre-apply of the q/k/v weight gradient -- once made a bucket's hand-over one launch early: invisible at one rank,
        # stale values reduced over the real gradient at two: tests/test_gpu_two_rank_overlap.py)
        lo, hi = tr.opt.grad.data_ptr(), tr.opt.grad.data_ptr() + tr.opt.grad.numel() * 4
        for name, idx in plan.grad_ready.items():
            op = plan.bwd_ops[idx]
            ptrs = [getattr(op.args, f) for f, _ in op.args._fields_]
            assert any(isinstance(v, int) and lo <= v < hi for v in ptrs), (name, idx, op.what)
        loss = tr._forward_backward_overlapped(*args, labels.cuda(), None, None, 1, 4 << 20)
        torch.cuda.synchronize()
        assert len(tr._buckets) >= 4 and float(loss) > 0
        assert torch.equal(tr.opt.grad, want)
        # the same exchange through the C ABI (pd_allreduce_bucket: RCCL reduce-scatter + all-gather, a one-rank communicator)
        from phendiff_amd.comm import NativeComm
        tr.opt.grad.zero_()
        tr.use_native_comm(NativeComm(0, 1))
        tr._forward_backward_overlapped(*args, labels.cuda(), None, None, 1, 4 << 20)
        torch.cuda.synchronize()
        assert torch.equal(tr.opt.grad, want)
        tr.native_comm.close()
    finally:
        dist.destroy_process_group()


def _pipes(mode, size=32):
    import phendiff_amd as P
    from oracle import ConditionalDDIMPipelineRef, DDIMSchedulerRef
    r, m = make_pair("super_small", size, mode)
    cfg = P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"]
    return (ConditionalDDIMPipelineRef(r, DDIMSchedulerRef(**cfg)), P.ConditionalDDIMPipeline(m, P.DDIMScheduler(**cfg)))


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 4e-2), ("fp16", 6e-3)])      # measured 4.2e-6 / 1.7e-2 (profiles/r4_parity_errors.json)
@pytest.mark.parametrize("p", [2, 1.5])
def test_guidance_gradient_through_unet_matches_autograd(mode, tol, p):
    """d Lp(x0_pred, target) / d image through the UNet and the scheduler's x0 formula (clipped), one step
    (torch.autograd.grad(losses_seq, images), utils_Img2Img.py:744-745)."""
    from oracle import lp_loss_ref
    from test_gpu_unet_ddib import synth_batch
    rp, pp = _pipes(mode)
    x, labels = synth_batch(2, 32)
    g = torch.Generator().manual_seed(3)
    images = (x + 0.3 * torch.randn(x.shape, generator=g)).requires_grad_(True)
    target = x.clone()
    rp.scheduler.set_timesteps(4)
    t = rp.scheduler.timesteps[1]
    mo = rp.unet(images, t, labels).sample
    x0 = rp.scheduler.step(mo, t, images).pred_original_sample
    losses = lp_loss_ref(x0, target, p)
    (ref,) = torch.autograd.grad([losses[0], losses[1]], images)

    import ctypes as C
    import phendiff_amd._lib as L
    dev = "cuda:0"
    plan = pp.unet.input_grad_plan(2, 32, 32, torch.device(dev))
    st = torch.cuda.current_stream().cuda_stream
    im, tg, lb = images.detach().to(dev).contiguous(), target.to(dev), labels.to(dev)
    out, d_out, d_dir = (torch.empty_like(im) for _ in range(3))
    pp.scheduler.set_timesteps(4)
    plan.forward(im, torch.full((2,), float(t), device=dev), lb, None, out, st)
    sa, sb, _, _, _ = pp.scheduler.step_coefficients(t)
    partial = torch.empty(2 * 4, dtype=torch.float64, device=dev)
    ls = torch.empty(2, device=dev)
    a = L.LpGuidanceArgs(numel=im.numel(), per_sample=im[0].numel(), pred_type=2, clip=1, clip_range=1.0, sqrt_a=sa, sqrt_b=sb,
                         p=float(p), sample=im.data_ptr(), model_out=out.data_ptr(), target=tg.data_ptr(),
                         partial=partial.data_ptr(), splits=4, d_model_out=d_out.data_ptr(), d_sample_direct=d_dir.data_ptr(),
                         losses=ls.data_ptr())
    L.check(L.lib().pd_lp_guidance(C.byref(a), st), "pd_lp_guidance")
    # fp16 engine (round 6): the loss gradient goes through the fp16 backward times the static scale of the guided transfer
    S = P_GUIDANCE_SCALE() if mode == "fp16" else 1.0
    plan.backward(d_out * S, st)
    torch.cuda.synchronize()
    assert torch.isfinite(plan.dsample).all()
    assert rel(ls, losses.detach()) < {"f32": 1e-5, "bf16": 2e-2, "fp16": 3e-3}[mode]
    assert rel(d_dir + plan.dsample / S, ref) < tol


def P_GUIDANCE_SCALE():
    from phendiff_amd.img2img import GUIDANCE_GRAD_SCALE
    return float(GUIDANCE_GRAD_SCALE)


@pytest.mark.parametrize("mode,tol", [("bf16", 6e-2), ("fp16", 8e-3)])
def test_gradient_guided_transfer_16_bit_engines_vs_golden(mode, tol, monkeypatch):
    """_linear_interp_custom_guidance_inverted_start end to end in the 16-bit engines against the committed oracle vectors (S = 3, p = 2,
    loss scale 0.5).  fp16 is the precision the reference runs this method in (general_config.yaml:46): the UNet input gradient is
    taken under the static scale, and a scale that overflows is halved and the step redone (forced here by starting at 2^30)."""
    import os
    import numpy as np
    import phendiff_amd as P
    import phendiff_amd.img2img as I
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "guided_super_small_32_s3.npz"))
    _, pp = _pipes(mode)
    x, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
    run = lambda: P.linear_interp_custom_guidance_inverted_start(pp, x.cuda(), labels.cuda(), (1 - labels).cuda(), float(d["p"]),
                                                                 float(d["guidance_loss_scale"]), 3, output_type="pt")
    got = run()
    assert rel(got, d["out"]) < tol
    if mode == "fp16":
        monkeypatch.setattr(I, "GUIDANCE_GRAD_SCALE", 2.0 ** 30)      # overflows fp16 on the first cast: halved until finite
        again = run()
        assert torch.isfinite(again).all() and rel(again, d["out"]) < tol


def test_gradient_guided_transfer_matches_oracle_f32():
    """_linear_interp_custom_guidance_inverted_start end to end (inversion + guided generation), S = 3.  The loss scale is
    raised from the reference default 1e-3 so that a wrong gradient could not hide inside the tolerance."""
    import phendiff_amd as P
    from oracle import linear_interp_custom_guidance_inverted_start_ref
    from test_gpu_unet_ddib import synth_batch
    rp, pp = _pipes("f32")
    x, labels = synth_batch(2, 32)
    target = 1 - labels
    ref = linear_interp_custom_guidance_inverted_start_ref(rp, x, labels, target, 2, 0.5, 3)
    got = P.linear_interp_custom_guidance_inverted_start(pp, x.cuda(), labels.cuda(), target.cuda(), 2, 0.5, 3, output_type="pt")
    assert rel(got, ref) < 2e-3
    plain = P.ddib(pp, x.cuda(), labels.cuda(), target.cuda(), 3, output_type="numpy")      # guidance must have had an effect
    moved = (got / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu().numpy()
    assert float(abs(moved - plain).max()) > 1e-2
    with pytest.raises(NotImplementedError):
        P.custom_guided_generation(pp, x.cuda(), target.cuda(), "inf", 0.001, 3)


def test_golden_fixture_guided_transfer_f32():
    """Committed oracle vectors (tests/golden/make_golden.py): weights seed 0, super_small @32, S = 3, p = 2, scale 0.5."""
    import os
    import numpy as np
    import phendiff_amd as P
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "guided_super_small_32_s3.npz"))
    _, pp = _pipes("f32")
    x, labels = torch.from_numpy(d["images"]), torch.from_numpy(d["labels"])
    got = P.linear_interp_custom_guidance_inverted_start(pp, x.cuda(), labels.cuda(), (1 - labels).cuda(), float(d["p"]),
                                                         float(d["guidance_loss_scale"]), 3, output_type="pt")
    assert rel(got, d["out"]) < 2e-3


def test_unet_backward_small_denoiser_and_non_square_f32():
    """Wider model (128/256/512 channels, concat inputs up to 1024) and a non-square sample (32 x 64)."""
    from phendiff_amd.unet_train import UNetTrainer
    import phendiff_amd as P
    for name, hw in (("small_denoiser_config", (32, 32)), ("super_small", (32, 64))):
        r, m = make_pair(name, 32, "f32")
        sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
        g = torch.Generator().manual_seed(9)
        B = 2
        clean = torch.rand(B, 3, *hw, generator=g) * 2 - 1
        noise = torch.randn(B, 3, *hw, generator=g)
        ts = torch.tensor([2100, 333])
        labels = torch.tensor([1, 0])
        acp = sched.alphas_cumprod[ts]
        sa, sb = (acp ** 0.5).view(-1, 1, 1, 1), ((1 - acp) ** 0.5).view(-1, 1, 1, 1)
        noisy, target = sa * clean + sb * noise, sa * noise - sb * clean
        _, ref = oracle_grads(r, noisy, ts, target, labels=labels)
        tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
        tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
        torch.cuda.synchronize()
        compare(ref, tr.grads, 3e-4, 3e-5)


def test_save_state_resume_continues_bitwise(tmp_path):
    """accelerate-layout checkpoint (utils_misc.py:322-347, utils_training.py:56-94): a run resumed from step_2 on a fresh
    trainer reproduces the uninterrupted run's losses exactly (deterministic kernels, ordered reductions)."""
    import os
    import phendiff_amd as P
    from phendiff_amd import train_state as TS
    sched, clean, noise, ts, labels, noisy, _ = batch(4, 32)
    args = [t.cuda() for t in (noisy, ts, clean, noise)]

    def fresh():
        _, m = make_pair("super_small", 32, "bf16")
        return P.UNetTrainer(m, sched, lr=3e-4)
    a = fresh()
    first = [float(a.step(*args, class_labels=labels.cuda())) for _ in range(2)]
    folder = TS.save_checkpoint(a, str(tmp_path), 2, checkpoints_total_limit=1)
    assert sorted(os.listdir(folder)) == ["custom_checkpoint_0.pkl", "optimizer.bin", "pytorch_model.bin", "random_states_0.pkl",
                                          "scheduler.bin"]
    rest = [float(a.step(*args, class_labels=labels.cuda())) for _ in range(3)]
    TS.save_checkpoint(a, str(tmp_path), 5, checkpoints_total_limit=1)
    assert os.listdir(tmp_path) == ["step_5"]                                  # total_limit pruning
    TS.save_checkpoint(a, str(tmp_path), 2, checkpoints_total_limit=None)       # put a step_2 lookalike back for "which="
    b = fresh()
    # resume from the real step-2 state: re-create it by replaying (the pruned folder is gone) -> use a second trainer
    c = fresh()
    for _ in range(2):
        c.step(*args, class_labels=labels.cuda())
    c.save_state(str(tmp_path / "again" / "step_2"))
    first_epoch, resume_step, global_step, sched_state = TS.resume_from_checkpoint(b, str(tmp_path / "again"), "latest", 10)
    assert (first_epoch, resume_step, global_step) == (0, 2, 2) and sched_state["last_epoch"] == 2 and b.opt.t == 2
    resumed = [float(b.step(*args, class_labels=labels.cuda())) for _ in range(3)]
    assert resumed == rest, (first, rest, resumed)
    assert torch.equal(b.opt.ema, a.opt.ema) and torch.equal(b.opt.flat, a.opt.flat)


def test_trainer_rebinds_after_the_model_dropped_its_packed_weights():
    """``model.to()`` / ``pipeline.to()`` / ``load_state_dict()`` drop the model's kernel-layout weights (``invalidate``); an inference
    forward then packs a NEW set.  The trainer must notice: its plans, gradient-layout weights and re-packer were bound to the old
    object, and ``refresh_weights`` would otherwise leave inference (pipelines, EMA evaluation) on stale weights."""
    import phendiff_amd as P
    sched, clean, noise, ts, labels, noisy, _ = batch(2, 32)
    args = [t.cuda() for t in (noisy, ts, clean, noise)]

    def run(disturb):
        _, m = make_pair("super_small", 32, "bf16")
        tr = P.UNetTrainer(m, sched, lr=3e-3)
        losses = [float(tr.step(*args, class_labels=labels.cuda()))]
        if disturb:
            m.to("cuda")                                        # same device: parameters keep aliasing the flat buffer
            assert m._weights is None
            m(noisy.cuda(), ts.cuda(), class_labels=labels.cuda())     # inference packs a new set from the current parameters
        losses += [float(tr.step(*args, class_labels=labels.cuda())) for _ in range(2)]
        out = m(noisy.cuda(), ts.cuda(), class_labels=labels.cuda()).sample.clone()
        return losses, out
    l0, o0 = run(False)
    l1, o1 = run(True)
    assert l0 == l1 and torch.equal(o0, o1)


# ---- round 5: fp16 training behind the reference's own flag (--mixed_precision fp16: launch_script_DDIM.sh:56, args_parser.py:381-390) ----
@pytest.mark.parametrize("size", [32, 128])
def test_fp16_backward_matches_autograd_under_the_loss_scale(size):
    """fp16 activations and activation gradients (f16 MFMA), fp32 parameter gradients that carry the GradScaler's scale (2**16 at the
    start, as accelerate builds it): gradients / scale against torch.autograd over the fp32 oracle at fp16's tolerance (11 mantissa bits:
    between the f32 engine's 2e-4 / 2e-5 and the bf16 engine's 8e-2 / 2e-2)."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", size, "fp16")
    sched, clean, noise, ts, labels, noisy, target = batch(3, size)
    loss_ref, ref = oracle_grads(r, noisy, ts, target, labels=labels)
    tr = UNetTrainer(m, sched, lr=1e-4, use_ema=False)
    assert tr.opt.scaler is not None and tr.opt.scaler.scale == 65536.0
    loss, _ = tr.forward_backward(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * float(loss_ref)          # the reported loss is the unscaled one
    got = {n: g / tr.opt.scaler.scale for n, g in tr.grads.items()}
    assert all(torch.isfinite(g).all() for g in got.values())
    compare(ref, got, 3e-2, 6e-3)


def test_fp16_overflow_step_is_skipped_and_the_scale_halves():
    """GradScaler semantics: a step whose gradients are not finite changes no parameter and no Adam moment, does not advance the step
    count, zeroes the gradients and halves the scale; the next step trains.  After `growth_interval` good steps the scale doubles."""
    from phendiff_amd.unet_train import UNetTrainer
    _, m = make_pair("super_small", 32, "fp16")
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    tr = UNetTrainer(m, sched, lr=1e-3, use_ema=True)
    sc = tr.opt.scaler
    sc.scale, sc.growth_interval = 2.0 ** 40, 3          # 2^40 x the loss gradient overflows fp16 on its first cast
    before, mom = tr.opt.flat.clone(), tr.opt.exp_avg.clone()
    args = (noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda())
    tr.step(*args, class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert sc.scale == 2.0 ** 39 and sc.skipped == 1 and tr.opt.t == 0
    assert torch.equal(tr.opt.flat, before) and torch.equal(tr.opt.exp_avg, mom) and float(tr.opt.grad.abs().max()) == 0.0
    assert torch.equal(tr.opt.ema, before)                # EMAModel.step on unchanged parameters
    sc.scale = 65536.0
    losses = []
    for _ in range(3):
        losses.append(float(tr.step(*args, class_labels=labels.cuda())))
    torch.cuda.synchronize()
    assert tr.opt.t == 3 and sc.skipped == 1 and sc.scale == 131072.0 and sc.growth_tracker == 0      # three good steps: doubled
    assert not torch.equal(tr.opt.flat, before) and torch.isfinite(tr.opt.flat).all() and losses[-1] < losses[0]
    assert 0.0 < float(tr.opt.grad_norm) < 1e3            # the reported norm is the UNSCALED gradients'
    # EMAModel.step ran on all four sync steps, the skipped one included (utils_training.py:553-556): its optimization_step is
    # optimizer steps + skipped steps, it is what the checkpoint stores, and a resumed run continues the decay schedule from it
    assert tr.opt.t_ema == 4


def test_fp16_checkpoint_with_a_skipped_step_resumes_on_the_same_ema_schedule(tmp_path):
    """A `--mixed_precision fp16` checkpoint whose run skipped a step (usual at init_scale 2**16) carries EMA optimization_step =
    optimizer step + skipped steps: it loads, and the resumed trainer's next EMA update uses the decay of step t_ema + 1 -- the
    uninterrupted run's."""
    import os
    from phendiff_amd import train_state as TS
    from phendiff_amd.unet_train import UNetTrainer
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    args = (noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda())

    def fresh():
        _, m = make_pair("super_small", 32, "fp16")
        return UNetTrainer(m, sched, lr=1e-3, use_ema=True)
    a = fresh()
    a.opt.scaler.scale = 2.0 ** 40                         # first step overflows and is skipped
    a.step(*args, class_labels=labels.cuda())
    a.opt.scaler.scale = 65536.0
    for _ in range(2):
        a.step(*args, class_labels=labels.cuda())
    assert (a.opt.t, a.opt.t_ema) == (2, 3)
    a.save_state(str(tmp_path / "ck" / "step_2"))
    esd = torch.load(os.path.join(tmp_path, "ck", "step_2", "custom_checkpoint_0.pkl"), map_location="cpu")
    assert esd["optimization_step"] == 3
    b = fresh()
    TS.load_state(b, str(tmp_path / "ck" / "step_2"))
    assert (b.opt.t, b.opt.t_ema) == (2, 3) and torch.equal(b.opt.ema, a.opt.ema)
    la, lb = float(a.step(*args, class_labels=labels.cuda())), float(b.step(*args, class_labels=labels.cuda()))
    torch.cuda.synchronize()
    assert la == lb and torch.equal(b.opt.ema, a.opt.ema) and torch.equal(b.opt.flat, a.opt.flat)


def test_fp16_training_tracks_the_bf16_loss_curve_and_checkpoints_its_scaler(tmp_path):
    """20 optimisation steps from the same weights on the same batches: the fp16 engine under its loss scale follows the bf16
    engine's losses (two reduced-precision roundings of the same trajectory), and accelerate's `scaler.pt` round-trips."""
    import os
    from phendiff_amd.unet_train import UNetTrainer
    curves = {}
    for mode in ("bf16", "fp16"):
        _, m = make_pair("super_small", 32, mode)
        sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
        tr = UNetTrainer(m, sched, lr=2e-4, use_ema=False)
        g = torch.Generator().manual_seed(3)
        out = []
        for k in range(20):
            nz = torch.randn(clean.shape, generator=g)
            t_k = torch.randint(0, 3000, (4,), generator=g)
            acp = sched.alphas_cumprod[t_k]
            nsy = (acp ** 0.5).view(-1, 1, 1, 1) * clean + ((1 - acp) ** 0.5).view(-1, 1, 1, 1) * nz
            out.append(float(tr.step(nsy.cuda(), t_k.cuda(), clean.cuda(), nz.cuda(), class_labels=labels.cuda())))
        curves[mode] = out
        if mode == "fp16":
            assert tr.opt.scaler.skipped == 0 and tr.opt.t == 20
            tr.opt.scaler.growth_tracker = 7
            folder = str(tmp_path / "step_20")
            tr.save_state(folder)
            sd = torch.load(os.path.join(folder, "scaler.pt"))
            assert sd["scale"] == 65536.0 and sd["_growth_tracker"] == 7 and sd["growth_interval"] == 2000
            tr.opt.scaler.scale, tr.opt.scaler.growth_tracker = 1.0, 0
            tr.load_state(folder)
            assert tr.opt.scaler.scale == 65536.0 and tr.opt.scaler.growth_tracker == 7
    a, b = torch.tensor(curves["bf16"]), torch.tensor(curves["fp16"])
    assert float(((a - b).abs() / a).max()) < 5e-2, (curves)
    assert b[-5:].mean() < b[:5].mean()                   # and it trains
