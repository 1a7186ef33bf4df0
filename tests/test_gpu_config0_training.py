"""BASELINE configs[0] as a parity case: the 10-step tiny DDIM training run of tests/test_oracle_training.py (CPU oracle =
the reference's loop over torch autograd / AdamW / LambdaLR / EMAModel) against the HIP trainer on identical draws:
per-step loss, learning-rate schedule, conditional and unconditional steps, final weights and EMA shadow."""
import pytest
import torch

from test_oracle_training import SCHED

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,loss_tol,w_tol", [("f32", 2e-4, 2e-5), ("bf16", 3e-2, 2e-3)])
def test_config0_ten_steps_match_the_cpu_loop(mode, loss_tol, w_tol):
    import phendiff_amd as P
    from oracle import CondUNet2DRef, DDIMSchedulerRef, TINY_CONFIG0_UNET, TrainingLoopRef, synthetic_two_class_batch
    from phendiff_amd.training import cosine_lr_factor
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    r = CondUNet2DRef(**{k: v for k, v in TINY_CONFIG0_UNET.items() if k in keys})
    m = P.CustomCondUNet2DModel(compute_dtype=mode, **TINY_CONFIG0_UNET)
    m.load_state_dict(r.state_dict())
    lr0, warmup, total = 2e-4, 2, 10
    loop = TrainingLoopRef(r, DDIMSchedulerRef(**SCHED), lr=lr0, warmup=warmup, total_steps=total, proba_uncond=0.3, seed=5)
    sched = P.DDIMScheduler(**SCHED)
    tr = P.UNetTrainer(m.to("cuda:0"), sched, lr=lr0, use_ema=True)
    clean, labels = synthetic_two_class_batch(8, 32, 1234)
    cg, lg = clean.cuda(), labels.cuda()
    for k in range(total):
        noise, ts, uncond = loop.sample(clean)                     # the oracle's draws feed both runs
        want = loop.step(clean, labels, noise, ts, uncond)
        noisy = sched.add_noise(cg, noise.cuda(), ts.cuda())
        kw = dict(class_emb=torch.zeros(8, m.time_embed_dim, device="cuda")) if uncond else dict(class_labels=lg)
        got = float(tr.step(noisy, ts.cuda(), cg, noise.cuda(), lr=lr0 * cosine_lr_factor(k, warmup, total), **kw))
        assert abs(got - want) < loss_tol * abs(want), (k, got, want)
    torch.cuda.synchronize()
    sd = dict(r.named_parameters())
    num = den = 0.0
    for n, p in m.named_parameters():
        num += float((p.detach().cpu() - sd[n].detach()).double().pow(2).sum())
        den += float(sd[n].detach().double().pow(2).sum())
    assert (num / den) ** 0.5 < w_tol
    # EMA shadow (diffusers EMAModel: decay_k = min(1 - (1 + k - 1)^-0.75, 0.9999), decay_1 = 0)
    ema_ref = torch.cat([s.reshape(-1) for s in loop.ema])
    order = [n for n, _ in P.training_param_order(m)]
    names_ref = [n for n, _ in r.named_parameters()]
    ema_by_name = dict(zip(names_ref, loop.ema))
    ema_got = tr.opt.ema.cpu()
    off, num, den = 0, 0.0, 0.0
    for n in order:
        e = ema_by_name[n].reshape(-1)
        num += float((ema_got[off:off + e.numel()] - e).double().pow(2).sum())
        den += float(e.double().pow(2).sum())
        off += e.numel()
    assert off == ema_ref.numel() and (num / den) ** 0.5 < w_tol
