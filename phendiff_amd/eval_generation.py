"""Eval-time class-conditional generation during training, the device part of
``_generate_samples_and_compute_metrics`` (``src/utils_training.py:642-803``): swap the EMA weights in, sample
``nb_generated_images`` images *per class* from pure Gaussian noise with one fixed-seed generator, swap the training weights
back.  The Inception-feature metrics (torch-fidelity, ``:944-1001``), W&B logging and PNG writing are the reference's control
plane and stay with the caller: every batch is handed back as the uint8 array the reference saves, under the file names it uses.

Everything below runs through the product pipelines (``ConditionalDDIMPipeline`` / ``CustomStableDiffusionImg2ImgPipeline``),
i.e. on the HIP engine; there is no CPU path."""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass, field
from math import ceil
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

EVAL_SEED = 5742877512           # utils_training.py:698 ("observe the outputs produced from the same Gaussian noise")


def split(l: Sequence, n: int, idx: int) -> list:
    """``utils_misc.split`` (``:63-71``): n contiguous parts whose lengths differ by at most one, the longer ones first."""
    k, m = divmod(len(l), n)
    return list(l[idx * k + min(idx, m):(idx + 1) * k + min(idx + 1, m)])


def eval_batch_sizes(nb_generated_images: int, eval_batch_size: int, num_processes: int = 1, process_index: int = 0) -> List[int]:
    """Per-class generation batches of THIS process (``utils_training.py:128-139``): ``ceil(n / bs)`` global batches, the last
    one ragged, dealt out contiguously over the processes."""
    tot = ceil(nb_generated_images / eval_batch_size)
    glob = [eval_batch_size] * (tot - 1) + [nb_generated_images - eval_batch_size * (tot - 1)]
    return split(glob, num_processes, process_index)


def get_initial_best_metric() -> float:
    return float("inf")


def is_it_best_model(main_metric_values: Sequence[float], best_metric: float) -> Tuple[bool, float]:
    """``utils_misc.is_it_best_model`` (``:350-366``): lower mean of the per-class main metric wins."""
    cur = float(np.mean(main_metric_values))
    return (True, cur) if cur < best_metric else (False, best_metric)


def images_to_uint8(images: np.ndarray) -> np.ndarray:
    """``(images * 255).round().astype("uint8")`` (``utils_training.py:861,925``; also what ``numpy_to_pil`` saves)."""
    return (np.asarray(images) * 255).round().astype("uint8")


def latents_preview(latents: torch.Tensor) -> np.ndarray:
    """The logged latent thumbnails (``utils_training.py:873-881``): channel mean, per-sample min-max to [0, 1], uint8."""
    p = latents.detach().float().mean(dim=1, keepdim=True)
    p = p - p.amin(dim=(2, 3), keepdim=True)
    p = p / p.amax(dim=(2, 3), keepdim=True)
    return (p.cpu().numpy() * 255).round().astype("uint8")


@contextmanager
def ema_weights(trainer, use_ema: bool = True):
    """diffusers ``EMAModel.store`` -> ``copy_to`` on entry, ``restore`` on exit (``utils_training.py:667-681``,
    ``:1046-1051``) over the trainer's flat fp32 buffers, re-packing the kernel-layout weight copies both times, so pipelines
    built on ``trainer.model`` sample with the EMA weights inside the block and training resumes untouched after it."""
    opt = trainer.opt
    if not use_ema or opt.ema is None:
        yield trainer
        return
    if not opt.flat.is_cuda:
        raise L.PhenDiffHipError("ema_weights needs the trainer's parameters on an MI355X device (no CPU fallback)")
    stored = opt.flat.clone()
    opt.flat.copy_(opt.ema)
    trainer.refresh_weights()
    try:
        yield trainer
    finally:
        opt.flat.copy_(stored)
        trainer.refresh_weights()


@dataclass
class GeneratedBatch:
    class_label: int
    class_name: str
    batch_idx: int
    filenames: List[str]
    images: np.ndarray                      # (b, H, W, C) float in [0, 1] — the pipeline's "numpy" output
    latents: Optional[torch.Tensor] = None  # SD only ("np+latent")

    @property
    def uint8(self) -> np.ndarray:
        return images_to_uint8(self.images)


def _filenames(local_process_index: int, eval_batch_size: int, batch_idx: int, n: int) -> List[str]:
    return [f"process_{local_process_index}_sample_{eval_batch_size * batch_idx + i}.png" for i in range(n)]


def generate_images_for_this_class_DDIM(pipeline, class_label: int, batch_sizes: Sequence[int], *, guidance_factor,
                                        num_inference_steps: int, generator, proba_uncond: float = 0.0,
                                        eval_batch_size: Optional[int] = None, local_process_index: int = 0,
                                        class_name: str = "") -> Iterator[GeneratedBatch]:
    """``_generate_save_images_for_this_class_DDIM`` (``utils_training.py:884-941``) minus the disk / W&B writes."""
    ebs = eval_batch_size if eval_batch_size is not None else (max(batch_sizes) if batch_sizes else 0)
    for batch_idx, bs in enumerate(batch_sizes):
        if proba_uncond == 1:
            labels, emb = None, torch.zeros((bs, pipeline.unet.time_embed_dim), device=pipeline.device)
        else:
            labels, emb = torch.full((bs,), class_label, device=pipeline.device).long(), None
        images = pipeline(labels, emb, guidance_factor, generator=generator, num_inference_steps=num_inference_steps,
                          output_type="numpy").images
        yield GeneratedBatch(class_label, class_name, batch_idx, _filenames(local_process_index, ebs, batch_idx, bs), images)


def generate_images_for_this_class_SD(pipeline, class_label: int, batch_sizes: Sequence[int], *, guidance_factor,
                                      num_inference_steps: int, generator, latent_hw: Tuple[int, int] = (16, 16),
                                      eval_batch_size: Optional[int] = None, local_process_index: int = 0,
                                      class_name: str = "") -> Iterator[GeneratedBatch]:
    """``_generate_save_images_for_this_class_SD`` (``utils_training.py:806-881``): from-noise latents of the hard-coded
    ``(b, 4, 16, 16)`` shape (a keyword here), ``strength = 1``, ``output_type = "np+latent"``."""
    ebs = eval_batch_size if eval_batch_size is not None else (max(batch_sizes) if batch_sizes else 0)
    dev = pipeline._execution_device
    for batch_idx, bs in enumerate(batch_sizes):
        images, latents = pipeline(image=None, latent_shape=(bs, 4, *latent_hw),
                                   class_labels=torch.tensor([class_label] * bs, device=dev).long(), strength=1,
                                   num_inference_steps=num_inference_steps, guidance_scale=guidance_factor,
                                   generator=generator, output_type="np+latent", device=dev)
        yield GeneratedBatch(class_label, class_name, batch_idx, _filenames(local_process_index, ebs, batch_idx, bs), images,
                             latents)


@dataclass
class EvalGeneration:
    batches: List[GeneratedBatch] = field(default_factory=list)

    def images_of(self, class_label: int) -> np.ndarray:
        return np.concatenate([b.images for b in self.batches if b.class_label == class_label])

    def files_of(self, class_label: int) -> List[str]:
        return [f for b in self.batches if b.class_label == class_label for f in b.filenames]


def generate_samples(pipeline, *, nb_classes: int, nb_generated_images: int, eval_batch_size: int, num_inference_steps: int,
                     guidance_factor=None, proba_uncond: float = 0.0, model_type: str = "DDIM", trainer=None,
                     use_ema: bool = True, class_names: Optional[Sequence[str]] = None, num_processes: int = 1,
                     process_index: int = 0, local_process_index: Optional[int] = None, generator=None,
                     latent_hw: Tuple[int, int] = (16, 16),
                     on_class_done: Optional[Callable[[int, str, List[GeneratedBatch]], None]] = None) -> EvalGeneration:
    """Steps 2–5 of ``_generate_samples_and_compute_metrics`` (``utils_training.py:664-769``).  ``trainer`` (optional): swap
    its EMA weights in for the duration.  ``generator`` defaults to the reference's: a generator on the pipeline's device
    seeded with ``EVAL_SEED``, shared by all classes and batches (so class c's noise depends on the batches drawn before
    it).  ``on_class_done(label, name, batches)`` is where the reference computes that class's FID/IS/KID."""
    if model_type not in ("DDIM", "StableDiffusion"):
        raise ValueError(f"Unknown model type {model_type}")
    dev = pipeline.device if model_type == "DDIM" else pipeline._execution_device
    if generator is None:
        generator = torch.Generator(device=dev).manual_seed(EVAL_SEED)
    sizes = eval_batch_sizes(nb_generated_images, eval_batch_size, num_processes, process_index)
    lpi = process_index if local_process_index is None else local_process_index
    if proba_uncond == 1:
        nb_classes = 1                                     # one pass for an unconditional model (:706-708)
    out = EvalGeneration()

    @contextmanager
    def _noop():
        yield

    with (ema_weights(trainer, use_ema) if trainer is not None else _noop()):
        for c in range(nb_classes):
            name = "unconditional" if proba_uncond == 1 else (class_names[c] if class_names is not None else str(c))
            if model_type == "DDIM":
                it = generate_images_for_this_class_DDIM(pipeline, c, sizes, guidance_factor=guidance_factor,
                                                         num_inference_steps=num_inference_steps, generator=generator,
                                                         proba_uncond=proba_uncond, eval_batch_size=eval_batch_size,
                                                         local_process_index=lpi, class_name=name)
            else:
                it = generate_images_for_this_class_SD(pipeline, c, sizes, guidance_factor=guidance_factor,
                                                       num_inference_steps=num_inference_steps, generator=generator,
                                                       latent_hw=latent_hw, eval_batch_size=eval_batch_size,
                                                       local_process_index=lpi, class_name=name)
            got = list(it)
            out.batches += got
            if on_class_done is not None:
                on_class_done(c, name, got)
    return out
