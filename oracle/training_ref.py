"""Oracle (test infrastructure): the reference's training step on CPU -- BASELINE configs[0] ("32x32 2-class synthetic, tiny
cond_unet_2d (64-ch) DDIM training, 10 steps on CPU").  Restates ``perform_training_epoch`` / ``_diffusion_and_backward`` /
``_syn_training_state`` (``src/utils_training.py:212-336,371-456,541-572``) with plain torch autograd, ``torch.optim.AdamW``
(``train.py:277-290``: betas (.95, .999), weight decay 1e-6, eps 1e-8), ``clip_grad_norm_(…, 1.0)``, the cosine-with-warm-up
LambdaLR of diffusers' ``get_scheduler("cosine")`` (``train.py:298-303``) and diffusers' ``EMAModel`` (Appendix A.12).

Parity unpinned like the rest of the oracle (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

TINY_CONFIG0_UNET = dict(
    act_fn="silu", attention_head_dim=8, block_out_channels=(64, 64), center_input_sample=False, class_embed_type=None,
    down_block_types=("DownBlock2D", "AttnDownBlock2D"), downsample_padding=1, flip_sin_to_cos=True, freq_shift=0, in_channels=3,
    layers_per_block=1, mid_block_scale_factor=1, norm_eps=1e-05, norm_num_groups=32, num_class_embeds=2, out_channels=3,
    resnet_time_scale_shift="default", sample_size=32, time_embedding_type="positional", up_block_types=("AttnUpBlock2D", "UpBlock2D"))


def synthetic_two_class_batch(B: int, size: int, seed: int):
    """x ~ U[-1, 1] + a class-dependent offset, labels = arange(B) % 2."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(B) % 2
    x = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    return (x + 0.25 * (2 * labels.float() - 1).view(B, 1, 1, 1)).clamp(-1, 1), labels


def cosine_lr_lambda(step, warmup, total):
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def ema_decay_ref(k, decay=0.9999, inv_gamma=1.0, power=0.75):
    step = max(0, k - 1)
    return 0.0 if step <= 0 else min(1 - (1 + step / inv_gamma) ** -power, decay)


class TrainingLoopRef:
    """One process of ``train.py`` for ``model_type == "DDIM"``; ``step`` = the body of ``perform_training_epoch``'s loop."""

    def __init__(self, unet, scheduler, lr=1e-4, warmup=2, total_steps=10, proba_uncond=0.1, seed=0, use_ema=True):
        self.unet, self.sched = unet, scheduler
        for p in unet.parameters():
            p.requires_grad_(True)
        self.params = list(unet.parameters())
        self.opt = torch.optim.AdamW(self.params, lr=lr, betas=(0.95, 0.999), weight_decay=1e-6, eps=1e-8)
        self.lr_sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: cosine_lr_lambda(s, warmup, total_steps))
        self.ema = [p.detach().clone() for p in self.params] if use_ema else None
        self.k = 0
        self.gen = torch.Generator().manual_seed(seed)              # noise / timesteps (the reference draws the noise on the CPU)
        self.flag_gen = torch.Generator().manual_seed(seed + 1)     # rank-0 `rand(1) < proba_uncond`
        self.proba_uncond = proba_uncond

    def sample(self, clean):
        B = clean.shape[0]
        noise = torch.randn(clean.shape, generator=self.gen)
        ts = torch.randint(0, self.sched.config.num_train_timesteps, (B,), generator=self.gen).long()
        uncond = bool(torch.rand(1, generator=self.flag_gen).item() < self.proba_uncond)
        return noise, ts, uncond

    def step(self, clean, labels, noise, ts, uncond):
        noisy = self.sched.add_noise(clean, noise, ts)
        if uncond:      # utils_training.py:510-516: class_labels=None, class_emb=zeros(B, time_embed_dim)
            out = self.unet(noisy, ts, class_labels=None, class_emb=torch.zeros(clean.shape[0], self.unet.time_embed_dim)).sample
        else:
            out = self.unet(noisy, ts, class_labels=labels).sample
        pt = self.sched.config.prediction_type
        if pt == "epsilon":
            loss = F.mse_loss(out, noise)
        elif pt == "sample":
            a = self.sched.alphas_cumprod[ts].view(-1, 1, 1, 1)
            loss = ((a / (1 - a)) * F.mse_loss(out, clean, reduction="none")).mean()
        else:
            loss = F.mse_loss(out, self.sched.get_velocity(clean, noise, ts))
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.params, 1.0)
        self.opt.step()
        self.lr_sched.step()
        self.opt.zero_grad()
        self.k += 1
        if self.ema is not None:
            d = ema_decay_ref(self.k)
            with torch.no_grad():
                for s, p in zip(self.ema, self.params):
                    s.sub_((1 - d) * (s - p))
        return float(loss.detach())

    @property
    def lr(self):
        return self.opt.param_groups[0]["lr"]
