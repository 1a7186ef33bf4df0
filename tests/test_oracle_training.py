"""BASELINE configs[0] on the CPU: 32x32 2-class synthetic images, tiny 64-channel cond_unet_2d, 10 DDIM training steps through
the oracle's restatement of the reference's training loop (plumbing check, no GPU), plus the schedule helpers of the product
against the oracle's."""
import torch

from oracle import (CondUNet2DRef, DDIMSchedulerRef, TINY_CONFIG0_UNET, TrainingLoopRef, cosine_lr_lambda, ema_decay_ref,
                    synthetic_two_class_batch)

SCHED = dict(num_train_timesteps=3000, beta_start=1e-4, beta_end=0.02, beta_schedule="scaled_linear", clip_sample=True,
             clip_sample_range=1.0, prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def test_config0_ten_training_steps_on_cpu():
    torch.manual_seed(0)
    keys = CondUNet2DRef.__init__.__code__.co_varnames
    unet = CondUNet2DRef(**{k: v for k, v in TINY_CONFIG0_UNET.items() if k in keys})
    assert sum(p.numel() for p in unet.parameters()) == 1_132_995           # tiny 64-channel model
    loop = TrainingLoopRef(unet, DDIMSchedulerRef(**SCHED), lr=2e-4, warmup=2, total_steps=10, proba_uncond=0.3, seed=5)
    clean, labels = synthetic_two_class_batch(8, 32, 1234)
    losses, unconds, lrs = [], [], []
    w0 = [p.detach().clone() for p in loop.params]
    for _ in range(10):
        noise, ts, uncond = loop.sample(clean)
        lrs.append(loop.lr)
        losses.append(loop.step(clean, labels, noise, ts, uncond))
        unconds.append(uncond)
    assert all(l == l and l < 10 for l in losses)
    assert any(unconds) and not all(unconds)                                # both kinds of step were exercised
    assert lrs[0] == 0.0 and abs(lrs[2] - 2e-4) < 1e-12 and lrs[-1] < lrs[2]  # warm-up then cosine decay
    assert any(float((a - b).abs().max()) > 0 for a, b in zip(w0, loop.params))
    # EMA lags the weights and the first EMA step copies them (decay_1 = 0)
    assert ema_decay_ref(1) == 0.0 and 0 < ema_decay_ref(2) < ema_decay_ref(10) < 0.9999
    assert any(float((s - p).abs().max()) > 0 for s, p in zip(loop.ema, loop.params))


def test_product_schedules_match_oracle():
    from phendiff_amd.training import cosine_lr_factor, ema_decay
    for k in range(0, 40):
        assert abs(cosine_lr_factor(k, 5, 30) - cosine_lr_lambda(k, 5, 30)) < 1e-12
    for k in range(1, 50):
        assert abs(ema_decay(k) - ema_decay_ref(k)) < 1e-12
