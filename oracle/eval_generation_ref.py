"""TEST INFRASTRUCTURE (CPU oracle) — eval-time generation during training, restating
``_generate_samples_and_compute_metrics`` / ``_generate_save_images_for_this_class_{DDIM,SD}``
(``src/utils_training.py:642-941``) and the eval-batch split of ``:128-139`` + ``utils_misc.split`` (``:63-71``).
Parity unpinned (see oracle/__init__.py).  The metrics (torch-fidelity) are not restated."""
from math import ceil

import numpy as np
import torch


def split_ref(l, n, idx):
    k, m = divmod(len(l), n)
    return [l[i * k + min(i, m):(i + 1) * k + min(i + 1, m)] for i in range(n)][idx]


def eval_batch_sizes_ref(nb_generated_images, eval_batch_size, num_processes=1, process_index=0):
    tot = ceil(nb_generated_images / eval_batch_size)
    glob = [eval_batch_size] * (tot - 1)
    glob += [nb_generated_images - eval_batch_size * (tot - 1)]
    return split_ref(glob, num_processes, process_index)


class EMASwapRef:
    """diffusers ``EMAModel.store`` / ``copy_to`` / ``restore`` on lists of tensors (``utils_training.py:667-681,1046-1051``)."""

    def __init__(self, shadow_params, params):
        self.shadow, self.params, self.temp = shadow_params, list(params), None

    def __enter__(self):
        self.temp = [p.detach().clone() for p in self.params]
        with torch.no_grad():
            for s, p in zip(self.shadow, self.params):
                p.copy_(s)
        return self

    def __exit__(self, *exc):
        with torch.no_grad():
            for t, p in zip(self.temp, self.params):
                p.copy_(t)
        self.temp = None


def eval_generation_ddim_ref(pipe, nb_classes, batch_sizes, guidance_factor, num_inference_steps, generator, proba_uncond=0.0,
                             eval_batch_size=None, local_process_index=0):
    """-> {class_label: (filenames, float images NHWC in [0, 1])}; one generator shared across classes and batches."""
    if proba_uncond == 1:
        nb_classes = 1
    ebs = eval_batch_size if eval_batch_size is not None else max(batch_sizes)
    out = {}
    for c in range(nb_classes):
        names, imgs = [], []
        for batch_idx, bs in enumerate(batch_sizes):
            if proba_uncond == 1:
                labels, emb = None, torch.zeros((bs, pipe.unet.time_embed_dim))
            else:
                labels, emb = torch.full((bs,), c).long(), None
            images = pipe(labels, emb, guidance_factor, generator=generator, num_inference_steps=num_inference_steps,
                          output_type="numpy").images
            imgs.append(images)
            names += [f"process_{local_process_index}_sample_{ebs * batch_idx + i}.png" for i in range(bs)]
        out[c] = (names, np.concatenate(imgs))
    return out


def eval_generation_sd_ref(pipe, nb_classes, batch_sizes, guidance_factor, num_inference_steps, generator, latent_hw=(16, 16),
                           initial_latents=None):
    """The reference draws the starting latents from the GLOBAL device RNG (``custom_pipeline...:407-409``: ``torch.randn``
    without the generator), which a CPU run cannot reproduce; ``initial_latents`` (one tensor per (class, batch), in loop
    order) stands in for those draws — fed as 4-channel "images" without forward noise, which is the same arithmetic."""
    out = {}
    draws = iter(initial_latents) if initial_latents is not None else None
    for c in range(nb_classes):
        imgs, lats = [], []
        for bs in batch_sizes:
            start = dict(image=None, latent_shape=(bs, 4, *latent_hw)) if draws is None else \
                dict(image=next(draws), add_forward_noise_to_image=False)
            images, latents = pipe(class_labels=torch.tensor([c] * bs).long(),
                                   strength=1, num_inference_steps=num_inference_steps, guidance_scale=guidance_factor,
                                   generator=generator, output_type="np+latent", **start)
            imgs.append(images)
            lats.append(latents)
        out[c] = (np.concatenate(imgs), torch.cat(lats))
    return out


def latents_preview_ref(latents):
    p = latents.mean(dim=1, keepdim=True)
    p -= p.amin(dim=(2, 3), keepdim=True)
    p /= p.amax(dim=(2, 3), keepdim=True)
    return (p.cpu().numpy() * 255).round().astype("uint8")
