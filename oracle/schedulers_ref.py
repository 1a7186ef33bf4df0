"""Oracle (test infrastructure): DDIM / inverse-DDIM schedulers, CPU fp32.

Parity unpinned (see ``oracle/__init__.py``): restates the published
``diffusers==0.18.2`` algorithms (``schedulers/scheduling_ddim.py`` and
``schedulers/scheduling_ddim_inverse.py``; pinned by the reference at
``environment.yaml:80``) as the reference consumes them:

* forward scheduler: ``src/pipeline_conditional_ddim/pipeline_conditionial_ddim.py:45,248-269,340-347``
  (``DDIMScheduler.from_config``, ``set_timesteps``, ``add_noise``, ``step``),
  ``src/utils_training.py:256,420,430`` (``add_noise``, ``get_velocity``);
* inverse scheduler: ``src/utils_Img2Img.py:776-779,794-798``
  (``DDIMInverseScheduler.from_config(pipe.scheduler.config)``, ``set_timesteps``, ``step``).

All tensor arithmetic mirrors the op order of the published code (separate
mul / sub / add roundings, 0-dim fp32 coefficient tensors).
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch


def _betas(beta_schedule: str, beta_start: float, beta_end: float, n: int) -> torch.Tensor:
    if beta_schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if beta_schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    if beta_schedule == "squaredcos_cap_v2":
        def alpha_bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        b = []
        for i in range(n):
            t1, t2 = i / n, (i + 1) / n
            b.append(min(1 - alpha_bar(t2) / alpha_bar(t1), 0.999))
        return torch.tensor(b, dtype=torch.float32)
    raise NotImplementedError(beta_schedule)


def rescale_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """diffusers 0.18.2 ``scheduling_ddim.rescale_zero_terminal_snr`` (arXiv 2305.08891 alg. 1)."""
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    alphas_bar_sqrt = alphas_cumprod.sqrt()
    a0 = alphas_bar_sqrt[0].clone()
    aT = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= aT
    alphas_bar_sqrt *= a0 / (a0 - aT)
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


class _Out(SimpleNamespace):
    pass


class DDIMSchedulerRef:
    """``diffusers.DDIMScheduler`` (0.18.2) restated; SURVEY.md Appendix A.7."""

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_one=True, steps_offset=0, prediction_type="epsilon",
                 thresholding=False, clip_sample_range=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **unused):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
            steps_offset=steps_offset, prediction_type=prediction_type, thresholding=thresholding,
            clip_sample_range=clip_sample_range, timestep_spacing=timestep_spacing,
            rescale_betas_zero_snr=rescale_betas_zero_snr)
        assert not thresholding, "dynamic thresholding is unused by the reference configs"
        self.betas = _betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d = {k: v for k, v in d.items() if not k.startswith("_")}
        return cls(**d)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        assert num_inference_steps <= n
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            step_ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            step_ratio = n / num_inference_steps
            ts = np.round(np.arange(n, 0, -step_ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts)

    def _get_variance(self, timestep, prev_timestep):
        a = self.alphas_cumprod[timestep]
        ap = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - ap) / (1 - a)) * (1 - a / ap)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        assert self.num_inference_steps is not None
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            pred_epsilon = model_output
        elif pt == "sample":
            pred_original_sample = model_output
            pred_epsilon = (sample - alpha_prod_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        elif pt == "v_prediction":
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        else:
            raise ValueError(pt)
        if self.config.clip_sample:
            r = self.config.clip_sample_range
            pred_original_sample = pred_original_sample.clamp(-r, r)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        if use_clipped_model_output:
            pred_epsilon = (sample - alpha_prod_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev_sample = prev_sample + std_dev_t * variance_noise
        if not return_dict:
            return (prev_sample,)
        return _Out(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def _coefs(self, original_samples, timesteps):
        acp = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = acp[timesteps] ** 0.5
        sa = sa.flatten()
        while len(sa.shape) < len(original_samples.shape):
            sa = sa.unsqueeze(-1)
        sb = (1 - acp[timesteps]) ** 0.5
        sb = sb.flatten()
        while len(sb.shape) < len(original_samples.shape):
            sb = sb.unsqueeze(-1)
        return sa, sb

    def add_noise(self, original_samples, noise, timesteps):
        sa, sb = self._coefs(original_samples, timesteps)
        return sa * original_samples + sb * noise

    def get_velocity(self, sample, noise, timesteps):
        sa, sb = self._coefs(sample, timesteps)
        return sa * noise - sb * sample

    def __len__(self):
        return self.config.num_train_timesteps


class DDIMInverseSchedulerRef:
    """``diffusers.DDIMInverseScheduler`` restated; SURVEY.md Appendix A.8.

    ``variant="0.18.2"`` (default, the version the reference pins): the class knows nothing
    about ``timestep_spacing`` / ``rescale_betas_zero_snr`` (they fall into ``**kwargs``), so
    its table is the UN-rescaled one and its timesteps are always "leading", ascending.
    The deprecated key ``set_alpha_to_one`` of the DDIM config is forwarded into
    ``set_alpha_to_zero``.  ``step`` looks at ``alphas_cumprod[t + N//S]``.

    ``variant="0.20+"``: the rewritten class (honours spacing + rescale, treats ``t`` as the
    destination level).  Kept behind the flag because the two differ materially and the
    version cannot be verified offline -- every fixture names its variant.
    """

    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 clip_sample=True, set_alpha_to_zero=True, steps_offset=0, prediction_type="epsilon",
                 clip_sample_range=1.0, timestep_spacing="leading", rescale_betas_zero_snr=False,
                 variant="0.18.2", **kwargs):
        if kwargs.get("set_alpha_to_one", None) is not None:
            set_alpha_to_zero = kwargs["set_alpha_to_one"]
        assert variant in ("0.18.2", "0.20+")
        self.variant = variant
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_zero=set_alpha_to_zero,
            steps_offset=steps_offset, prediction_type=prediction_type, clip_sample_range=clip_sample_range,
            timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self.betas = _betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        if variant == "0.20+" and rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        if variant == "0.18.2":
            self.final_alpha_cumprod = torch.tensor(0.0) if set_alpha_to_zero else self.alphas_cumprod[-1]
        else:
            self.initial_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps).copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **overrides):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d = {k: v for k, v in d.items() if not k.startswith("_")}
        d.pop("thresholding", None)
        d.update(overrides)
        return cls(**d)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        assert num_inference_steps <= n
        self.num_inference_steps = num_inference_steps
        if self.variant == "0.18.2":
            step_ratio = n // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round().copy().astype(np.int64)
            ts += self.config.steps_offset
        else:
            sp = self.config.timestep_spacing
            if sp == "leading":
                step_ratio = n // num_inference_steps
                ts = (np.arange(0, num_inference_steps) * step_ratio).round().copy().astype(np.int64)
                ts += self.config.steps_offset
            elif sp == "trailing":
                step_ratio = n / num_inference_steps
                ts = np.round(np.arange(n, 0, -step_ratio)[::-1]).astype(np.int64)
                ts -= 1
            else:
                raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False,
             variance_noise=None, return_dict=True):
        n = self.config.num_train_timesteps
        ratio = n // self.num_inference_steps
        if self.variant == "0.18.2":
            prev_timestep = timestep + ratio
            alpha_prod_t = self.alphas_cumprod[timestep]
            alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep < n else self.final_alpha_cumprod
        else:
            prev_timestep = timestep - ratio
            alpha_prod_t = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.initial_alpha_cumprod
            alpha_prod_t_prev = self.alphas_cumprod[timestep]
        beta_prod_t = 1 - alpha_prod_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            pred_epsilon = model_output
        elif pt == "sample":
            pred_original_sample = model_output
            pred_epsilon = (sample - alpha_prod_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        elif pt == "v_prediction":
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        else:
            raise ValueError(pt)
        if self.config.clip_sample:
            r = self.config.clip_sample_range
            pred_original_sample = pred_original_sample.clamp(-r, r)
        pred_sample_direction = (1 - alpha_prod_t_prev) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if not return_dict:
            return (prev_sample, pred_original_sample)
        return _Out(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def __len__(self):
        return self.config.num_train_timesteps
