"""Partial-parameter training (/root/reference/train.py:189-220): components frozen with ``requires_grad_(False)``,
``--attention_fine_tuning`` (``unet.requires_grad_(False)`` then ``module.attentions.requires_grad_(True)``), class-embedding-only
training -- against ``torch.autograd`` + ``clip_grad_norm_`` + ``torch.optim.AdamW`` on the CPU oracle with the same flags (torch
skips a parameter whose ``.grad`` is None: no decay, no moments, no step, not in the norm)."""
import os

import pytest
import torch

from test_gpu_unet_backward import batch
from test_gpu_unet_ddib import make_pair, rel

pytestmark = pytest.mark.gpu


def attention_fine_tuning(unet, verbatim=True):
    """train.py:201-220.  ``verbatim``: the reference's own loop (``hasattr`` then ``.requires_grad_``), which the product must
    survive as diffusers' blocks do (a block without attention has NO ``attentions`` attribute); the oracle keeps ``None`` there."""
    unet.requires_grad_(False)
    for module in unet.modules():
        if hasattr(module, "attentions") and (verbatim or module.attentions is not None):
            module.attentions.requires_grad_(True)


def oracle_step(r, opt, noisy, ts, target, **cond):
    for p in r.parameters():
        p.grad = None
    out = r(noisy, ts, **cond).sample
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(r.parameters(), 1.0)
    opt.step()
    return float(loss.detach())


def adamw(r, lr):
    return torch.optim.AdamW(r.parameters(), lr=lr, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)


def check_against_oracle(r, m, before, trainable_names, tol=5e-4):
    sd = r.state_dict()
    num = den = 0.0
    for n, p in m.named_parameters():
        got = p.detach().cpu()
        if n in trainable_names:
            assert not torch.equal(got, before[n]), f"{n} trains but did not move"
            num += float((got - sd[n]).double().pow(2).sum())
            den += float((sd[n] - before[n]).double().pow(2).sum())
        else:
            assert torch.equal(got, before[n]), f"frozen parameter {n} changed"
            assert torch.equal(sd[n], before[n])
    assert (num / den) ** 0.5 < tol, (num / den) ** 0.5      # relative to the UPDATE, not to the weights


def test_attention_fine_tuning_follows_torch():
    """unet.requires_grad_(False) + module.attentions.requires_grad_(True): only the 60 attention parameters move, exactly as
    torch's AdamW moves them; no convolution weight-gradient launch is emitted; the global norm covers the attention gradients only."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    attention_fine_tuning(r, verbatim=False)
    attention_fine_tuning(m)
    trainable = {n for n, p in r.named_parameters() if p.requires_grad}
    assert len(trainable) == 60 and all(".attentions." in n for n in trainable)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    tr = UNetTrainer(m, sched, lr=2e-4, use_ema=True)
    assert set(tr.grads) - tr.frozen == trainable
    opt = adamw(r, 2e-4)
    for _ in range(3):
        loss_ref = oracle_step(r, opt, noisy, ts, target, class_labels=labels)
        loss = float(tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda()))
        assert abs(loss - loss_ref) < 2e-4 * abs(loss_ref)
    torch.cuda.synchronize()
    plan = tr.plan_for(4, 32, 32)
    kinds = {op.what for op in plan.bwd_ops}
    assert not any(k.startswith("wgrad3x3") for k in kinds) and "linear_wgrad" not in kinds, kinds
    assert set(plan.grad_ready) == trainable
    # the norm that clipped: torch's over the attention gradients of the LAST step
    ref_norm = torch.norm(torch.stack([p.grad.norm() for p in r.parameters() if p.grad is not None]))
    assert all(p.grad is None for n, p in r.named_parameters() if n not in trainable)
    check_against_oracle(r, m, before, trainable)
    # EMA shadow of a frozen parameter == the parameter (diffusers EMAModel.step copies it)
    off = 0
    for (n, p) in [(n, tr.params[n]) for n in tr.grads]:
        k = p.numel()
        if n in tr.frozen:
            assert torch.equal(tr.opt.ema[off:off + k], tr.opt.flat[off:off + k]), n
        off += k
    # inference sees the fine-tuned attention weights
    with torch.no_grad():
        ref_out = r(noisy, ts, class_labels=labels).sample
    assert rel(m(noisy.cuda(), ts.cuda(), class_labels=labels.cuda()).sample, ref_out) < 1e-4
    assert float(ref_norm) > 0
    # optimizer.bin of the checkpoint == torch's AdamW.state_dict(): a state entry for the parameters that got gradients, NONE for the frozen ones
    # (an unfrozen-later parameter must not resume with a bias-correction step it never took)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr.save_state(d)
        osd = torch.load(os.path.join(d, "optimizer.bin"), map_location="cpu")
    ref_osd = opt.state_dict()
    assert set(osd["state"]) == set(ref_osd["state"]) and len(osd["state"]) == 60
    assert all(float(v["step"]) == 3.0 for v in osd["state"].values())


def test_clip_norm_excludes_frozen_gradients():
    """A launch that writes several parameters' gradients at once (the fused q/k/v projection) may write a frozen member's segment:
    it must not enter clip_grad_norm_ (torch: its .grad is None)."""
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    names = [n for n, _ in r.named_parameters() if n.endswith("attentions.0.to_q.weight") or n.endswith("attentions.0.to_q.bias")]
    for model in (r, m):
        model.requires_grad_(False)
        for n, p in model.named_parameters():
            if n in names:
                p.requires_grad_(True)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    tr = UNetTrainer(m, sched, lr=1e-3, use_ema=False, max_grad_norm=1e-3)          # a bound that always clips
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    opt = adamw(r, 1e-3)
    for p in r.parameters():
        p.grad = None
    loss = torch.nn.functional.mse_loss(r(noisy, ts, class_labels=labels).sample, target)
    loss.backward()
    ref_norm = float(torch.nn.utils.clip_grad_norm_(r.parameters(), 1e-3))
    opt.step()
    tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    assert abs(float(tr.opt.grad_norm) - ref_norm) < 1e-4 * ref_norm, (float(tr.opt.grad_norm), ref_norm)
    check_against_oracle(r, m, before, set(names))


def test_class_embedding_only_and_explicit_names():
    from phendiff_amd.unet_train import UNetTrainer
    r, m = make_pair("super_small", 32, "f32")
    r.requires_grad_(False)
    r.class_embedding.requires_grad_(True)
    sched, clean, noise, ts, labels, noisy, target = batch(4, 32)
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    tr = UNetTrainer(m, sched, lr=1e-3, use_ema=False, trainable=["class_embedding.weight"])
    opt = adamw(r, 1e-3)
    zeros = torch.zeros(4, 256)
    for uncond in (False, True, False):
        if uncond:
            # the table has no gradient on an unconditional step (nothing in the graph requires grad: torch has nothing to do at
            # all), so the engine must leave it -- and its moments and step count -- alone
            tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_emb=zeros.cuda())
        else:
            oracle_step(r, opt, noisy, ts, target, class_labels=labels)
            tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), class_labels=labels.cuda())
    torch.cuda.synchronize()
    kinds = {op.what for op in tr.plan_for(4, 32, 32).bwd_ops}
    assert not any(k.startswith("wgrad") for k in kinds), kinds
    check_against_oracle(r, m, before, {"class_embedding.weight"})


def test_fully_frozen_model_is_refused_and_fresh_model_trains_everything():
    import phendiff_amd as P
    from phendiff_amd.training import resolve_trainable
    from phendiff_amd.unet_train import UNetTrainer, training_param_order
    _, m = make_pair("super_small", 32, "f32")
    sched = P.DDIMScheduler(**P.SCHEDULER_CONFIGS["3k_steps_clipping_rescaling"])
    assert all(resolve_trainable(training_param_order(m), m))                # as built: everything trains (diffusers' default)
    m.requires_grad_(False)                                                  # the caller froze the component (train.py:192-193)
    with pytest.raises(ValueError, match="no trainable parameter"):
        UNetTrainer(m, sched, lr=1e-4)
    with pytest.raises(ValueError, match="unknown parameter names"):
        UNetTrainer(m, sched, lr=1e-4, trainable=["no.such.weight"])


def test_sd_frozen_unet_trains_the_custom_embedding_only():
    """components_to_train = class_embedding (train.py:189-199): the SD UNet frozen, only the CustomEmbedding table moves."""
    import phendiff_amd as P
    from oracle import class_emb_to_encoder_hidden_states as ehs_ref
    from test_gpu_sd_unet import TINY, make_pair as make_sd_pair
    from test_gpu_sd_unet_backward import batch as sd_batch
    r, emb, m, e2 = make_sd_pair(TINY, "f32")
    sched, clean, noise, ts, labels, noisy, target = sd_batch(4, 16)
    r.requires_grad_(False)
    m.requires_grad_(False)
    before = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    table_before = e2.inner_module.weight.detach().cpu().clone()
    tr = P.SDUNetTrainer(m, e2, sched, lr=1e-3, use_ema=False)
    assert set(tr.grads) - tr.frozen == {"class_embedding.inner_module.weight"}
    opt = torch.optim.AdamW(emb.parameters(), lr=1e-3, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
    for _ in range(2):
        for p in emb.parameters():
            p.grad = None
        ehs = ehs_ref(emb(labels))
        loss = torch.nn.functional.mse_loss(r(noisy, ts, ehs).sample, target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(emb.parameters(), 1.0)
        opt.step()
        got = float(tr.step(noisy.cuda(), ts.cuda(), clean.cuda(), noise.cuda(), labels.cuda()))
        assert abs(got - float(loss.detach())) < 5e-4 * abs(float(loss.detach()))
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert torch.equal(p.detach().cpu(), before[n]), n
    upd = emb.inner_module.weight.detach() - table_before
    assert float(upd.abs().max()) > 0
    assert float((e2.inner_module.weight.detach().cpu() - emb.inner_module.weight.detach()).norm() / upd.norm()) < 1e-4
    kinds = {op.what for op in tr.plan_for(4, 16, 16).bwd_ops}
    assert not any(k.startswith("wgrad") for k in kinds), kinds
