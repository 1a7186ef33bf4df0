"""``DDIMScheduler`` / ``DDIMInverseScheduler`` with the diffusers-0.18.2 call surface the reference uses
(``src/pipeline_conditional_ddim/pipeline_conditionial_ddim.py:45,248-269,340-347``,
``src/utils_Img2Img.py:776-779,794-798``, ``src/utils_training.py:249,256,415-430``).

Host side (this file): the noise tables (``alphas_cumprod``), the int64 timestep grids (bit-exact w.r.t. the
published algorithm) and the four fp32 coefficients of one update, computed with 0-dim fp32 CPU tensors in the
same op order as the reference so that they round identically.  Device side: ``pd_ddim_step`` /
``pd_add_noise`` (one fused elementwise launch per update instead of ~10).
"""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace

import numpy as np
import torch

from . import _lib as L



def randn_tensor(shape, generator, device, dtype=torch.float32):
    """diffusers ``randn_tensor``: a CPU generator draws on the CPU (then moves), a device generator on the device; a LIST of
    generators draws one ``(1, ...)`` sample per generator (per-sample reproducibility) and concatenates."""
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"list of {len(generator)} generators for a batch of {shape[0]}")
        return torch.cat([randn_tensor((1,) + tuple(shape[1:]), g, device, dtype) for g in generator], 0)
    gdev = generator.device if generator is not None else device
    return torch.randn(tuple(shape), generator=generator, device=gdev, dtype=dtype).to(device)


class SchedulerOutput(SimpleNamespace):
    """``DDIMSchedulerOutput`` stand-in: ``.prev_sample``, ``.pred_original_sample``."""


def make_betas(schedule: str, beta_start: float, beta_end: float, n: int) -> torch.Tensor:
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":  # the schedule of every shipped config (models_configs/noise_scheduler/*.json)
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    if schedule == "squaredcos_cap_v2":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return torch.tensor([min(1 - f((i + 1) / n) / f(i / n), 0.999) for i in range(n)], dtype=torch.float32)
    raise NotImplementedError(f"beta_schedule {schedule}")


def enforce_zero_terminal_snr(betas: torch.Tensor) -> torch.Tensor:
    """Shift/scale sqrt(alpha_bar) so that the last level has exactly zero SNR (arXiv 2305.08891, alg. 1)."""
    root = torch.cumprod(1.0 - betas, dim=0).sqrt()
    first, last = root[0].clone(), root[-1].clone()
    root = (root - last) * (first / (first - last))
    bar = root ** 2
    alphas = torch.cat([bar[:1], bar[1:] / bar[:-1]])
    return 1 - alphas


_PRED = L.PD_PRED


class _SchedulerBase:
    order = 1
    init_noise_sigma = 1.0
    _config_keys = ()

    def _finish_init(self, cfg: dict, rescale: bool):
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.prediction_type not in _PRED:
            raise ValueError(f"prediction_type {c.prediction_type}")
        betas = make_betas(c.beta_schedule, c.beta_start, c.beta_end, c.num_train_timesteps)
        if rescale:
            betas = enforce_zero_terminal_snr(betas)
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_inference_steps = None
        self._coef_dev = {}

    @classmethod
    def from_config(cls, config, **overrides):
        d = dict(config) if isinstance(config, dict) else dict(vars(config))
        d = {k: v for k, v in d.items() if not k.startswith("_")}
        d.update(overrides)
        return cls(**d)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **overrides):
        import os
        from .checkpoint import load_scheduler
        return load_scheduler(cls, os.path.join(path, subfolder) if subfolder else path, **overrides)

    def save_pretrained(self, path):
        from .checkpoint import save_scheduler
        save_scheduler(self, path)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    @staticmethod
    def _t_int(timestep) -> int:
        return int(timestep.item()) if torch.is_tensor(timestep) else int(timestep)

    # subclasses: (alpha_prod_t, alpha_prod_t_prev, sigma) as 0-dim fp32 tensors
    def _levels(self, t: int, eta: float):
        raise NotImplementedError

    def step_coefficients(self, timestep, eta: float = 0.0):
        """(sqrt_a, sqrt_b, sqrt_a_prev, dir_coef, sigma) as python floats holding fp32 values."""
        t = self._t_int(timestep)
        a, ap, sigma = self._levels(t, eta)
        sa, sb = a ** 0.5, (1 - a) ** 0.5
        sap = ap ** 0.5
        dirc = (1 - ap - sigma ** 2) ** 0.5
        return float(sa), float(sb), float(sap), float(dirc), float(sigma)

    def _device_step(self, model_output, timestep, sample, eta, use_clipped_model_output, generator, variance_noise,
                     uncond_output=None, w=None, guidance_cfg=False, out=None, want_x0=True, stream=None):
        if not (sample.is_cuda and model_output.is_cuda):
            raise L.PhenDiffHipError("phendiff_amd schedulers step on MI355X tensors only (no CPU fallback)")
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        lib = L.lib()
        sa, sb, sap, dirc, sigma = self.step_coefficients(timestep, eta)
        x = sample.contiguous().float()
        mo = model_output.contiguous().float()
        prev = torch.empty_like(x) if out is None else out
        x0 = torch.empty_like(x) if want_x0 else None
        c = self.config
        wt = None
        if uncond_output is not None:
            wt = (w if torch.is_tensor(w) else torch.tensor([float(w)])).to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        a = L.DdimStepArgs(numel=x.numel(), per_sample=x[0].numel(), pred_type=_PRED[c.prediction_type],
                           clip=int(bool(c.clip_sample)), clip_range=float(c.clip_sample_range),
                           use_clipped_model_output=int(bool(use_clipped_model_output)),
                           sqrt_a=sa, sqrt_b=sb, sqrt_ap=sap, dir_coef=dirc, sample=x.data_ptr(), model_out=mo.data_ptr(),
                           uncond_out=L.ptr(uncond_output), w=L.ptr(wt), w_per_sample=int(wt is not None and wt.numel() > 1),
                           guidance_cfg=int(guidance_cfg), prev_sample=prev.data_ptr(), pred_x0=L.ptr(x0))
        st = stream if stream is not None else torch.cuda.current_stream(x.device).cuda_stream
        L.check(lib.pd_ddim_step(C.byref(a), st), "pd_ddim_step")
        if eta > 0:
            if variance_noise is None:
                variance_noise = randn_tensor(mo.shape, generator, mo.device, mo.dtype)
            prev = prev + sigma * variance_noise.to(device=mo.device, dtype=mo.dtype)
        return prev, x0

    def _per_sample_coefs(self, timesteps, device):
        """sqrt(alpha_bar_t), sqrt(1 - alpha_bar_t) per sample.  The tables are computed once on the host in fp32 (the
        reference's arithmetic, bit for bit) and gathered on the device: no host sync when ``timesteps`` live there."""
        tabs = getattr(self, "_coef_tables", None)
        if tabs is None or tabs[0].device != torch.device(device):
            acp = self.alphas_cumprod
            tabs = ((acp ** 0.5).to(device), ((1 - acp) ** 0.5).to(device))
            self._coef_tables = tabs
        t = timesteps.detach().to(device=device, dtype=torch.long).reshape(-1)
        return tabs[0][t].contiguous(), tabs[1][t].contiguous()

    def _mix(self, x, noise, timesteps, velocity):
        if not x.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd schedulers run on MI355X tensors only (no CPU fallback)")
        x = x.contiguous().float()
        noise = noise.contiguous().float()
        if timesteps.numel() == 1:
            timesteps = timesteps.reshape(1).expand(x.shape[0])
        sa, sb = self._per_sample_coefs(timesteps, x.device)
        out = torch.empty_like(x)
        a = L.AddNoiseArgs(numel=x.numel(), per_sample=x[0].numel(), velocity=int(velocity), x=x.data_ptr(),
                           noise=noise.data_ptr(), sa=sa.data_ptr(), sb=sb.data_ptr(), out=out.data_ptr())
        L.check(L.lib().pd_add_noise(C.byref(a), torch.cuda.current_stream(x.device).cuda_stream), "pd_add_noise")
        return out

    def add_noise(self, original_samples, noise, timesteps):
        return self._mix(original_samples, noise, timesteps, False)

    def get_velocity(self, sample, noise, timesteps):
        return self._mix(sample, noise, timesteps, True)


class DDIMScheduler(_SchedulerBase):
    """diffusers ``DDIMScheduler`` surface (constructor defaults of 0.18.2)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **ignored):
        if trained_betas is not None or thresholding:
            raise NotImplementedError("trained_betas / dynamic thresholding are not used by PhenDiff's configs")
        cfg = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                   beta_schedule=beta_schedule, trained_betas=None, clip_sample=clip_sample,
                   set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
                   thresholding=False, dynamic_thresholding_ratio=dynamic_thresholding_ratio,
                   clip_sample_range=clip_sample_range, sample_max_value=sample_max_value,
                   timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr)
        self._finish_init(cfg, rescale_betas_zero_snr)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = torch.from_numpy(np.arange(num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {n}")
        self.num_inference_steps = num_inference_steps
        mode = self.config.timestep_spacing
        if mode == "linspace":
            grid = np.linspace(0, n - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif mode == "leading":
            grid = (np.arange(num_inference_steps) * (n // num_inference_steps)).round()[::-1].copy().astype(np.int64)
            grid += self.config.steps_offset
        elif mode == "trailing":
            grid = np.round(np.arange(n, 0, -(n / num_inference_steps))).astype(np.int64) - 1
        else:
            raise ValueError(f"{mode} is not supported. Choose one of 'leading', 'trailing' or 'linspace'.")
        self.timesteps = torch.from_numpy(grid)  # kept on the host: indexing tables must not sync the device
        return self

    def _levels(self, t, eta):
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        sigma = torch.tensor(0.0)
        if eta:
            variance = ((1 - ap) / (1 - a)) * (1 - a / ap)
            sigma = eta * variance ** 0.5
        return a, ap, sigma

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        prev, x0 = self._device_step(model_output, timestep, sample, eta, use_clipped_model_output, generator, variance_noise)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDIMInverseScheduler(_SchedulerBase):
    """diffusers ``DDIMInverseScheduler`` surface.  ``variant="0.18.2"`` (default = the version the reference
    pins, ``environment.yaml:80``): ignores ``timestep_spacing`` / ``rescale_betas_zero_snr`` (table NOT
    rescaled, ascending "leading" grid), forwards the deprecated ``set_alpha_to_one`` into ``set_alpha_to_zero``,
    and steps from level ``t`` to ``t + N//S``.  ``variant="0.20+"`` is the later rewrite."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_zero=True, steps_offset=0,
                 prediction_type="epsilon", clip_sample_range=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, variant="0.18.2", **kwargs):
        if kwargs.get("set_alpha_to_one") is not None:
            set_alpha_to_zero = kwargs["set_alpha_to_one"]
        if variant not in ("0.18.2", "0.20+"):
            raise ValueError("variant must be '0.18.2' or '0.20+'")
        if trained_betas is not None:
            raise NotImplementedError("trained_betas")
        self.variant = variant
        cfg = dict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                   beta_schedule=beta_schedule, trained_betas=None, clip_sample=clip_sample,
                   set_alpha_to_zero=set_alpha_to_zero, steps_offset=steps_offset, prediction_type=prediction_type,
                   clip_sample_range=clip_sample_range, timestep_spacing=timestep_spacing,
                   rescale_betas_zero_snr=rescale_betas_zero_snr)
        self._finish_init(cfg, rescale_betas_zero_snr and variant == "0.20+")
        self.final_alpha_cumprod = torch.tensor(0.0) if set_alpha_to_zero else self.alphas_cumprod[-1]
        self.initial_alpha_cumprod = torch.tensor(1.0)
        self.timesteps = torch.from_numpy(np.arange(num_train_timesteps).copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than {n}")
        self.num_inference_steps = num_inference_steps
        mode = "leading" if self.variant == "0.18.2" else self.config.timestep_spacing
        if mode == "leading":
            grid = (np.arange(num_inference_steps) * (n // num_inference_steps)).round().copy().astype(np.int64)
            grid += self.config.steps_offset
        elif mode == "trailing":
            grid = np.round(np.arange(n, 0, -(n / num_inference_steps))[::-1]).astype(np.int64) - 1
        else:
            raise ValueError(f"{mode} is not supported. Choose one of 'leading' or 'trailing'.")
        self.timesteps = torch.from_numpy(grid)
        return self

    def _levels(self, t, eta):
        n = self.config.num_train_timesteps
        ratio = n // self.num_inference_steps
        if self.variant == "0.18.2":
            nxt = t + ratio
            a = self.alphas_cumprod[t]
            ap = self.alphas_cumprod[nxt] if nxt < n else self.final_alpha_cumprod
        else:
            src = t - ratio
            a = self.alphas_cumprod[src] if src >= 0 else self.initial_alpha_cumprod
            ap = self.alphas_cumprod[t]
        return a, ap, torch.tensor(0.0)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             variance_noise=None, return_dict: bool = True):
        prev, x0 = self._device_step(model_output, timestep, sample, 0.0, False, None, None)
        if not return_dict:
            return (prev, x0)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)
