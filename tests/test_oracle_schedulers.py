"""Scheduler known answers and closed-form properties (CPU)."""
import numpy as np
import pytest
import torch

from oracle import DDIMInverseSchedulerRef, DDIMSchedulerRef

CFG_3K = dict(num_train_timesteps=3000, beta_start=1e-4, beta_end=0.02, beta_schedule="scaled_linear",
              clip_sample=True, clip_sample_range=1.0, prediction_type="v_prediction",
              rescale_betas_zero_snr=True, timestep_spacing="trailing")
CFG_1K = dict(CFG_3K, num_train_timesteps=1000, prediction_type="epsilon")
CFG_SD = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
              clip_sample=False, prediction_type="v_prediction", rescale_betas_zero_snr=False,
              set_alpha_to_one=False, steps_offset=1, timestep_spacing="leading")


def test_tables_3k():
    s = DDIMSchedulerRef(**CFG_3K)
    a = s.alphas_cumprod
    assert a.dtype == torch.float32 and a.shape == (3000,)
    assert float(a[0]) == pytest.approx(0.999899983, rel=1e-6)
    assert float(a[1]) == pytest.approx(0.999799132, rel=1e-6)
    assert float(a[1500]) == pytest.approx(3.6693e-02, rel=1e-4)
    assert float(a[2998]) == pytest.approx(4.0904e-14, rel=1e-3)
    assert float(a[2999]) == 0.0
    s.set_timesteps(50)
    assert s.timesteps.dtype == torch.int64
    assert s.timesteps.tolist() == list(range(2999, 0, -60))
    s.set_timesteps(100)
    assert s.timesteps.tolist() == list(range(2999, 0, -30))


def test_tables_other_configs():
    s = DDIMSchedulerRef(**CFG_1K)
    assert float(s.alphas_cumprod[500]) == pytest.approx(3.178154e-01, rel=1e-5)
    assert float(s.alphas_cumprod[999]) == 0.0
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(999, 0, -20))
    s = DDIMSchedulerRef(**CFG_SD)
    assert float(s.alphas_cumprod[0]) == pytest.approx(0.999149978, rel=1e-6)
    assert float(s.alphas_cumprod[999]) == pytest.approx(4.660095e-03, rel=1e-5)
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(981, 0, -20))
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])


def test_inverse_scheduler_0182_semantics():
    fwd = DDIMSchedulerRef(**CFG_3K)
    inv = DDIMInverseSchedulerRef.from_config(fwd.config)
    # un-rescaled table, ascending "leading" timesteps, final alpha 0 (A.8)
    assert float(inv.alphas_cumprod[2999]) > 0 and float(inv.final_alpha_cumprod) == 0.0
    inv.set_timesteps(50)
    assert inv.timesteps.tolist() == list(range(0, 3000, 60))
    x = torch.randn(1, 3, 4, 4)
    v = torch.randn(1, 3, 4, 4)
    out = inv.step(v, inv.timesteps[-1], x)
    a = inv.alphas_cumprod[2940]
    eps = a ** 0.5 * v + (1 - a) ** 0.5 * x
    assert torch.allclose(out.prev_sample, eps, atol=1e-6)  # a' = 0  =>  x_next = predicted noise
    inv_sd = DDIMInverseSchedulerRef.from_config(DDIMSchedulerRef(**CFG_SD).config)
    assert float(inv_sd.final_alpha_cumprod) == float(inv_sd.alphas_cumprod[-1])  # set_alpha_to_one=False forwarded


def test_inverse_variant_020():
    fwd = DDIMSchedulerRef(**CFG_3K)
    inv = DDIMInverseSchedulerRef.from_config(fwd.config, variant="0.20+")
    assert float(inv.alphas_cumprod[2999]) == 0.0
    inv.set_timesteps(50)
    assert inv.timesteps.tolist() == list(range(59, 3000, 60))


@pytest.mark.parametrize("pt", ["epsilon", "sample", "v_prediction"])
def test_step_identities(pt):
    """With a perfect model, one DDIM step from level t lands exactly on level t_prev."""
    cfg = dict(CFG_3K, prediction_type=pt, rescale_betas_zero_snr=False, clip_sample=False)
    s = DDIMSchedulerRef(**cfg)
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x0 = torch.rand(2, 3, 8, 8, generator=g) * 2 - 1
    noise = torch.randn(2, 3, 8, 8, generator=g)
    t = s.timesteps[10]
    xt = s.add_noise(x0, noise, t.repeat(2))
    target = {"epsilon": noise, "sample": x0, "v_prediction": s.get_velocity(x0, noise, t.repeat(2))}[pt]
    out = s.step(target, t, xt)
    assert torch.allclose(out.pred_original_sample, x0, atol=2e-4)
    t_prev = s.timesteps[11]
    assert torch.allclose(out.prev_sample, s.add_noise(x0, noise, t_prev.repeat(2)), atol=2e-4)


def test_zero_model_closed_form():
    """A zero-output v-model makes the denoising loop closed-form: x <- sqrt(a'a) x + sqrt((1-a')(1-a)) x."""
    s = DDIMSchedulerRef(**dict(CFG_3K, clip_sample=False))
    s.set_timesteps(10)
    x = torch.ones(1, 1, 2, 2)
    ref = 1.0
    for i, t in enumerate(s.timesteps):
        x = s.step(torch.zeros_like(x), t, x).prev_sample
        a = float(s.alphas_cumprod[t])
        tp = int(t) - 300
        ap = float(s.alphas_cumprod[tp]) if tp >= 0 else 1.0
        ref = np.sqrt(ap) * np.sqrt(a) * ref + np.sqrt(1 - ap) * np.sqrt(1 - a) * ref
    assert float(x.flatten()[0]) == pytest.approx(ref, rel=1e-4)
