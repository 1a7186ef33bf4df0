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
@pytest.mark.parametrize("size", [32, 64])
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
        loss = tr._forward_backward_overlapped(*args, labels.cuda(), None, None, 1, 4 << 20)
        torch.cuda.synchronize()
        assert len(tr._buckets) >= 4 and float(loss) > 0
        assert torch.equal(tr.opt.grad, want)
    finally:
        dist.destroy_process_group()
