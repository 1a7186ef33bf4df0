"""Training-step building blocks around the UNet forward/backward (``unet_train.py``) -- SURVEY.md 8a rows A13-A16.

Device side (``csrc/train_kernels.hip``): ``pd_diffusion_loss`` (loss + d loss/d out, ``utils_training.py:415-433``),
``pd_grad_norm`` (``clip_grad_norm_(params, 1.0)``, ``:438-440``) and ``pd_adamw_ema`` (clip scaling + AdamW +
``EMAModel.step`` in one pass over flat fp32 buffers, ``:452-454,553-556``).  Host side (this file): the schedules the
reference takes from diffusers (EMA warm-up decay, cosine LR with warm-up), ``lr * sqrt(world)`` (``train.py:277``), the
per-step sampling of ``perform_training_epoch`` (``:244-256``), the unconditional-step coin flip WITHOUT the per-step
barrier + broadcast (``:262-275``: every rank seeds the same host RNG, only rank 0's draw was ever used), and
data-parallel gradient averaging (what DDP's all-reduce computes, ``train.py:311-326``).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Iterable, List, Optional

import torch

from . import _lib as L


# ---- host-side schedules -------------------------------------------------------------------------------------------
def ema_decay(optimization_step: int, decay: float = 0.9999, use_ema_warmup: bool = True, inv_gamma: float = 1.0,
              power: float = 0.75, update_after_step: int = 0, min_decay: float = 0.0) -> float:
    """diffusers ``EMAModel.get_decay`` (0.18.2), called with the 1-based step count (``EMAModel.step`` increments first).
    Reference settings: train.py:229-237 (decay .9999, warm-up on, inv_gamma 1, power .75)."""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    cur = 1 - (1 + step / inv_gamma) ** -power if use_ema_warmup else (1 + step) / (10 + step)
    return max(min(cur, decay), min_decay)


def cosine_lr_factor(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    """diffusers ``get_scheduler("cosine")`` LambdaLR factor (train.py:298-303)."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    progress = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def scaled_lr(lr: float, world_size: int) -> float:
    """``lr *= sqrt(num_processes)`` (train.py:277)."""
    return lr * math.sqrt(world_size)


class UnconditionalStepFlags:
    """Replaces the per-step barrier + 8-byte broadcast of ``rand(1) < proba_uncond`` (utils_training.py:262-275): every
    rank draws from an identically seeded host generator, so all ranks agree with zero traffic."""

    def __init__(self, seed: int, proba_uncond: float):
        self.g = torch.Generator().manual_seed(int(seed))
        self.p = float(proba_uncond)

    def next(self) -> bool:
        return bool(torch.rand(1, generator=self.g).item() < self.p)


@torch.no_grad()
def sample_training_inputs(clean_images: torch.Tensor, scheduler, cpu_generator: Optional[torch.Generator] = None,
                           device_generator: Optional[torch.Generator] = None):
    """perform_training_epoch's per-step sampling: noise on the CPU RNG then H2D (utils_training.py:244), timesteps
    ``randint(0, N, (B,))`` on the device (:247-252), ``noisy = add_noise(clean, noise, t)`` (:256, ``pd_add_noise``)."""
    B = clean_images.shape[0]
    noise = torch.randn(clean_images.shape, generator=cpu_generator).to(clean_images.device)
    timesteps = torch.randint(0, scheduler.config.num_train_timesteps, (B,), device=clean_images.device,
                              generator=device_generator).long()
    return noise, timesteps, scheduler.add_noise(clean_images, noise, timesteps)


# ---- which parameters train (train.py:189-220) -----------------------------------------------------------------------
def mark_requires_grad_calls(cls):
    """Class decorator for the engine's model classes: remember that ``requires_grad_`` was CALLED on the model (the reference
    freezes components with ``pipeline.unet.requires_grad_(False)`` and re-enables sub-modules with
    ``module.attentions.requires_grad_(True)``, train.py:189-220).  The engine's parameters are built with ``requires_grad=False``
    (there is no autograd graph; gradients come from the HIP backward plan), so "all False" alone cannot tell a freshly built model
    -- everything trains, as with diffusers' default of True -- from one the caller froze."""
    orig = cls.requires_grad_

    def requires_grad_(self, requires_grad: bool = True):
        self._requires_grad_explicit = True
        return orig(self, requires_grad)

    cls.requires_grad_ = requires_grad_
    return cls


def resolve_trainable(named_params, module=None, trainable=None) -> List[bool]:
    """One flag per (name, parameter) pair of a trainer's flat buffers -- what ``torch.optim.AdamW`` would update in the reference:
    every parameter whose ``.grad`` is not None after ``accelerator.backward`` = every parameter with ``requires_grad``.

    ``trainable``: an explicit choice (iterable of names, or ``f(name, parameter) -> bool``).  Otherwise the parameters'
    ``requires_grad`` flags decide: if ANY is set, exactly those train (``--attention_fine_tuning``: ``unet.requires_grad_(False)``
    then ``module.attentions.requires_grad_(True)``); if none is set and ``requires_grad_`` was never called on ``module``, all of
    them train (a freshly built engine model); if it was called (the caller froze the whole component), none does."""
    named_params = list(named_params)
    if trainable is not None:
        if callable(trainable):
            return [bool(trainable(n, p)) for n, p in named_params]
        names = set(trainable)
        unknown = names - {n for n, _ in named_params}
        if unknown:
            raise ValueError(f"trainable: unknown parameter names {sorted(unknown)[:5]}")
        return [n in names for n, _ in named_params]
    flags = [bool(p.requires_grad) for _, p in named_params]
    if any(flags):
        return flags
    if module is not None and getattr(module, "_requires_grad_explicit", False):
        return flags
    return [True] * len(flags)


# ---- device-side wrappers ------------------------------------------------------------------------------------------
class DiffusionLoss:
    """``loss, dloss/dout = DiffusionLoss(scheduler)(model_out, clean, noise, timesteps)`` (utils_training.py:415-433)."""

    def __init__(self, scheduler, device):
        self.scheduler = scheduler
        self.partial = torch.empty(1024, dtype=torch.float64, device=device)
        acp = scheduler.alphas_cumprod.float()
        # host-computed fp32 tables (the reference's arithmetic), gathered on the device: no host sync per step
        self.tables = tuple(t.to(device) for t in (acp / (1 - acp), acp ** 0.5, (1 - acp) ** 0.5))

    def __call__(self, model_out, clean, noise, timesteps, want_grad=True, grad_scale=1.0):
        if not model_out.is_cuda:
            raise L.PhenDiffHipError("phendiff_amd runs on MI355X only (no CPU fallback)")
        pt = self.scheduler.config.prediction_type
        dev = model_out.device
        mo, cl, nz = model_out.contiguous().float(), clean.contiguous().float(), noise.contiguous().float()
        if self.tables[0].device != dev:
            self.tables = tuple(t.to(dev) for t in self.tables)
        idx = timesteps.detach().to(device=dev, dtype=torch.long)
        w = sa = sb = None
        if pt == "sample":
            w = self.tables[0][idx].contiguous()                # SNR weights (extract_into_tensor, utils_misc.py:33-48)
        elif pt == "v_prediction":
            sa, sb = self.tables[1][idx].contiguous(), self.tables[2][idx].contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = torch.empty_like(mo) if want_grad else None
        a = L.LossArgs(numel=mo.numel(), per_sample=mo[0].numel(), pred_type=L.PD_PRED[pt], model_out=mo.data_ptr(),
                       noise=nz.data_ptr(), clean=cl.data_ptr(), weight=L.ptr(w), sa=L.ptr(sa), sb=L.ptr(sb),
                       grad_scale=float(grad_scale), grad_out=L.ptr(grad), partial=self.partial.data_ptr(),
                       loss_out=loss.data_ptr())
        L.check(L.lib().pd_diffusion_loss(C.byref(a), torch.cuda.current_stream(dev).cuda_stream), "pd_diffusion_loss")
        return loss, grad


class LossScaler:
    """``torch.cuda.amp.GradScaler`` as accelerate builds it for ``--mixed_precision fp16`` (args_parser.py:381-390,
    launch_script_DDIM.sh:56; accelerate 0.23 passes no arguments: init_scale 2**16, growth 2, backoff 0.5, growth_interval 2000):
    the loss gradient is multiplied by ``scale`` before the fp16 backward (``pd_diffusion_loss.grad_scale``); the optimizer un-scales
    the fp32 parameter gradients, SKIPS the step when their norm is not finite and halves the scale, and doubles it after
    ``growth_interval`` consecutive good steps.  ``state_dict`` is the dict accelerate writes as ``scaler.pt``."""

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        self.scale, self.growth_factor, self.backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self.growth_interval, self.growth_tracker = int(growth_interval), 0
        self.skipped = 0                      # steps skipped so far (diagnostic)

    def update(self, found_inf: bool):
        if found_inf:
            self.scale *= self.backoff_factor
            self.growth_tracker = 0
            self.skipped += 1
        else:
            self.growth_tracker += 1
            if self.growth_tracker == self.growth_interval:
                self.scale *= self.growth_factor
                self.growth_tracker = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self.growth_tracker}

    def load_state_dict(self, sd):
        self.scale, self.growth_factor, self.backoff_factor = float(sd["scale"]), float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval, self.growth_tracker = int(sd["growth_interval"]), int(sd["_growth_tracker"])


class FlatAdamWEMA:
    """AdamW (torch defaults of the reference: betas (.95, .999), wd 1e-6, eps 1e-8; args_parser.py:299-321) + global
    grad-norm clipping to ``max_grad_norm`` + diffusers EMA, fused over ONE flat fp32 buffer.  ``params`` are re-pointed
    to views of the flat buffer (their ``.grad`` to views of the flat gradient), so modules see the updates in place."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.95, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-6, max_grad_norm: Optional[float] = 1.0, use_ema: bool = True,
                 ema_kwargs: Optional[dict] = None):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params or not self.params[0].is_cuda:
            raise L.PhenDiffHipError("FlatAdamWEMA needs parameters on an MI355X device (no CPU fallback)")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.grad[off:off + k].view_as(p)
            off += k
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.ema = self.flat.clone() if use_ema else None
        self.lr, self.betas, self.eps, self.wd, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.ema_kwargs = ema_kwargs or {}
        self.scaler: Optional[LossScaler] = None      # fp16 training: set by the trainer; the gradients in self.grad then carry scaler.scale
        self.t = 0
        self.runs = None            # [(offset, numel, trainable)] when some parameters are frozen (see set_trainable), else None
        self.tail = None            # (offset, numel): a trailing segment with its own AdamW step count (see set_tail)
        self.tail_names = ()
        self.t_tail = 0
        # diffusers' EMAModel.optimization_step: EMAModel.step runs on EVERY sync step of the reference's loop, skipped fp16 steps
        # included (utils_training.py:438-456, 553-556), so it is the optimizer's step count PLUS the skipped steps -- its own counter,
        # saved / loaded as `optimization_step`, and the decay schedule is derived from it
        self.t_ema = 0
        self.partial = torch.empty(1024, dtype=torch.float64, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.clip_coef = torch.ones(1, dtype=torch.float32, device=dev)

    def set_tail(self, numel: int, names=()):
        """The last ``numel`` elements are a parameter that does not receive a gradient on every step (the ``CustomEmbedding``
        on unconditional steps): torch's AdamW skips a parameter whose ``.grad`` is None and counts its steps separately, so
        that segment keeps its own step count and ``step(tail_active=False)`` leaves it (and its moments) untouched."""
        self.tail = (self.flat.numel() - numel, numel)
        self.tail_names = tuple(names)        # state_dict names of the tail parameters (checkpoints record their own step count)

    def set_trainable(self, flags):
        """``flags[i]``: parameter ``i`` trains.  Frozen parameters get what ``torch.optim.AdamW`` gives a parameter whose ``.grad`` is
        None (train.py:189-220 freezes with ``requires_grad_(False)``): no decay, no moment update, no step; their gradient segment is
        zeroed before the global norm, so they do not enter ``clip_grad_norm_`` either (a launch that writes several parameters'
        gradients at once -- the fused q/k/v projection, the stacked ``time_emb_proj`` -- may have written them); the EMA shadow of
        a frozen parameter is the parameter itself (diffusers ``EMAModel.step`` copies it).  Adjacent parameters of equal status
        are one run = one ``pd_adamw_ema`` launch."""
        flags = [bool(f) for f in flags]
        if len(flags) != len(self.params):
            raise ValueError("set_trainable: one flag per parameter")
        self.trainable = flags
        self._frozen_views = None
        if all(flags):
            self.runs = None
            return
        runs, off = [], 0
        for p, f in zip(self.params, flags):
            k = p.numel()
            if runs and runs[-1][2] == f:
                runs[-1][1] += k
            else:
                runs.append([off, k, f])
            off += k
        self.runs = [tuple(r) for r in runs]

    def trainable_ranges(self):
        """[(offset, numel)] of the flat buffers' trainable runs (the whole buffer when nothing is frozen)."""
        if self.runs is None:
            return [(0, self.flat.numel())]
        return [(o, k) for o, k, f in self.runs if f]

    def step(self, lr: Optional[float] = None, zero_grad: bool = True, tail_active: bool = True):
        """clip_grad_norm_ -> optimizer.step -> zero_grad -> EMA.step (utils_training.py:438-454, 553-556). No host sync:
        the gradient norm stays on the device (``self.grad_norm``).  With a :class:`LossScaler` (fp16 training) the gradients carry its
        scale: the norm is taken of the scaled gradients against ``max_grad_norm * scale`` (the same clip decision), ONE host read of
        that norm per step decides whether the step is skipped (``GradScaler.step``: not finite -> no parameter / moment update, the
        step count does not advance, EMA still steps on its own counter ``t_ema`` -- accelerate's loop calls ``ema.step`` regardless), and the factor the update
        multiplies the gradients with becomes clip_coef / scale."""
        lib = L.lib()
        st = torch.cuda.current_stream(self.flat.device).cuda_stream
        lr = self.lr if lr is None else lr
        b1, b2 = self.betas
        if self.runs is not None:
            # frozen runs: whatever a fused launch wrote into a frozen member's segment (the q/k/v projection, the stacked time_emb_proj)
            # must not count in the norm -- ONE multi-tensor launch for all of them (attention fine-tuning alternates: dozens of runs)
            if getattr(self, "_frozen_views", None) is None or self._frozen_views[0] is not self.grad:
                self._frozen_views = (self.grad, [self.grad[off:off + k] for off, k, f in self.runs if not f])
            if self._frozen_views[1]:
                torch._foreach_zero_(self._frozen_views[1])
        S = self.scaler.scale if self.scaler is not None else None
        if self.max_grad_norm is not None or S is not None:
            max_norm = (float(self.max_grad_norm) if self.max_grad_norm is not None else float("inf")) * (S or 1.0)
            L.check(lib.pd_grad_norm(self.grad.data_ptr(), self.grad.numel(), self.partial.data_ptr(), max_norm,
                                     self.grad_norm.data_ptr(), self.clip_coef.data_ptr(), st), "pd_grad_norm")
        if S is not None:
            found_inf = not math.isfinite(float(self.grad_norm))        # the one host read of an fp16 step (GradScaler's found_inf)
            self.scaler.update(found_inf)
            if found_inf:      # skipped step: EMA of the unchanged parameters, gradients zeroed, no step count
                self.t_ema += 1
                d = ema_decay(self.t_ema, **self.ema_kwargs) if self.ema is not None else 0.0
                a = L.AdamWEmaArgs(numel=self.flat.numel(), lr=0.0, beta1=b1, beta2=b2, eps=self.eps, weight_decay=0.0, step_size=0.0,
                                   bias_correction2_sqrt=1.0, one_minus_decay=1.0 - d, zero_grad=int(zero_grad), clip_coef=None,
                                   param=self.flat.data_ptr(), grad=self.grad.data_ptr(), exp_avg=self.exp_avg.data_ptr(),
                                   exp_avg_sq=self.exp_avg_sq.data_ptr(), ema=self.ema.data_ptr() if self.ema is not None else None, ema_only=1)
                L.check(lib.pd_adamw_ema(C.byref(a), st), "pd_adamw_ema")
                return
            if self.max_grad_norm is None:
                self.clip_coef.fill_(1.0 / S)
            else:
                self.clip_coef.mul_(1.0 / S)         # (one element: the factor the update applies to the SCALED gradients)
            self.grad_norm.mul_(1.0 / S)             # the reported norm is the unscaled gradients'
        self.t += 1
        self.t_ema += 1
        d = ema_decay(self.t_ema, **self.ema_kwargs) if self.ema is not None else 0.0
        clip = self.clip_coef.data_ptr() if (self.max_grad_norm is not None or S is not None) else None

        def launch(off, numel, t, ema_only=0):
            a = L.AdamWEmaArgs(numel=numel, lr=lr, beta1=b1, beta2=b2, eps=self.eps, weight_decay=self.wd,
                               step_size=lr / (1 - b1 ** t), bias_correction2_sqrt=math.sqrt(1 - b2 ** t),
                               one_minus_decay=1.0 - d, zero_grad=int(zero_grad), clip_coef=clip,
                               param=self.flat.data_ptr() + 4 * off, grad=self.grad.data_ptr() + 4 * off,
                               exp_avg=self.exp_avg.data_ptr() + 4 * off, exp_avg_sq=self.exp_avg_sq.data_ptr() + 4 * off,
                               ema=(self.ema.data_ptr() + 4 * off) if self.ema is not None else None, ema_only=ema_only)
            L.check(lib.pd_adamw_ema(C.byref(a), st), "pd_adamw_ema")

        body_end = self.flat.numel() if self.tail is None else self.tail[0]
        if self.runs is None:
            launch(0, body_end, self.t)
        else:
            for off, k, f in self.runs:
                k = min(off + k, body_end) - off         # (the tail segment, when it trains, is stepped below with its own count)
                if f and k > 0:
                    launch(off, k, self.t)
        if self.tail is None:
            return
        off, n = self.tail
        if self.runs is not None and not self.trainable[-1]:
            return                                        # a frozen tail parameter: nothing to do
        if tail_active:
            self.t_tail += 1
            launch(off, n, self.t_tail)
        else:
            launch(off, n, 1, ema_only=1)


def broadcast_from_rank0_(flat: torch.Tensor, group=None, src: int = 0) -> torch.Tensor:
    """What wrapping a model in ``DistributedDataParallel`` does first (``accelerator.prepare``, train.py:311-326): every rank
    takes rank 0's parameters, so ranks that were initialised or loaded differently cannot diverge silently.  One collective over
    the flat fp32 buffer (every parameter is a view of it).  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat
    dist.broadcast(flat, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return flat


def allreduce_mean_ranges_(flat_grad: torch.Tensor, ranges, group=None, bucket_bytes: int = 256 << 20):
    """:func:`allreduce_mean_` over the trainable runs ``[(offset, numel)]`` of one flat buffer (frozen parameters in between are not
    exchanged): every collective is issued asynchronously before the first wait, so a layout that alternates frozen and trainable
    parameters (``--attention_fine_tuning``: dozens of runs) costs one latency, not one per run."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    world = dist.get_world_size(group)
    if world == 1:
        return flat_grad
    per = max(1, bucket_bytes // flat_grad.element_size())
    views = [flat_grad[o + i:o + min(k, i + per)] for o, k in ranges for i in range(0, k, per)]
    works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in views]
    for wk in works:
        wk.wait()
    torch._foreach_div_(views, float(world))
    return flat_grad


def allreduce_mean_(flat_grad: torch.Tensor, group=None, bucket_bytes: int = 256 << 20):
    """Data-parallel gradient averaging over one flat buffer: what DDP's bucketed all-reduce computes for
    ``accelerator.backward`` (utils_training.py:436, train.py:311-326), as a few LARGE asynchronous all-reduces (RCCL over
    xGMI is per-link bound: big buckets, 288 GB HBM) followed by the 1/world scale.  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    world = dist.get_world_size(group)
    if world == 1:
        return flat_grad
    per = max(1, bucket_bytes // flat_grad.element_size())
    works = [dist.all_reduce(flat_grad[i:i + per], op=dist.ReduceOp.SUM, group=group, async_op=True)
             for i in range(0, flat_grad.numel(), per)]
    for wk in works:
        wk.wait()
    flat_grad.div_(world)
    return flat_grad
