"""``CustomStableDiffusionImg2ImgPipeline`` with the reference's call surface
(``src/custom_pipeline_stable_diffusion_img2img/custom_pipeline_stable_diffusion_img2img.py:43-730``)
on the HIP engine: class labels -> ``CustomEmbedding`` -> one token + 76 zero tokens as ``encoder_hidden_states``; VAE encode
(``prepare_latents``) -> optional forward noising -> DDIM denoising with the SD UNet (classifier-free guidance as ONE 2B-batch
UNet evaluation, exactly as the reference concatenates it) -> VAE decode -> ``VaeImageProcessor.postprocess``.

What runs where: UNet, VAE, scheduler update (+ the guidance combine, fused into ``pd_ddim_step``), forward noising, latent
sampling and post-processing are HIP kernels; torch is used for the random draws, the class-embedding row gather, and two
small latent copies per step under guidance (plumbing).
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, List, Optional, Tuple, Union
from warnings import warn

import torch

from . import _lib as L
from .pipeline import numpy_to_pil
from .schedulers import DDIMScheduler, randn_tensor
from .vae import VaeImageProcessor

DEFAULT_NUM_INFERENCE_STEPS = 50


def hack_class_embedding(cl_embed: torch.Tensor) -> torch.Tensor:
    """``utils_Img2Img.py:179-187`` / ``custom_pipeline...:626-634``: (N, E) -> (N, 77, E) with 76 zero tokens."""
    assert cl_embed.ndim == 2, "Expecting a tensor of shape (N, E)"
    bs, ed = cl_embed.shape
    cl_embed = cl_embed.reshape(bs, 1, ed)
    padding = torch.zeros_like(cl_embed).repeat(1, 76, 1).to(cl_embed.device)
    return torch.cat([cl_embed, padding], dim=1)


class CustomStableDiffusionImg2ImgPipeline:
    def __init__(self, vae, unet, scheduler, class_embedding):
        cfg = dict(vars(scheduler.config)) if not isinstance(scheduler.config, dict) else dict(scheduler.config)
        patched = False
        if cfg.get("steps_offset", 1) != 1:                   # :74-91 (deprecation fix-up)
            cfg["steps_offset"], patched = 1, True
        if cfg.get("clip_sample", False) is True:             # :93-109
            cfg["clip_sample"], patched = False, True
        if patched:
            scheduler = type(scheduler).from_config(cfg)
        self._modules = {}
        self.register_modules(vae=vae, unet=unet, scheduler=scheduler, class_embedding=class_embedding)
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)       # :144
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)  # :145
        self._progress_bar_config = {}

    # ---- DiffusionPipeline protocol pieces the reference touches -----------------------------------------------------
    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            self._modules[k] = v
            setattr(self, k, v)

    @property
    def components(self):
        return dict(self._modules)

    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.device

    def to(self, *args, **kwargs):
        for m in (self.vae, self.unet, self.class_embedding):
            m.to(*args, **kwargs)
        return self

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable=None, total=None):
        if self._progress_bar_config.get("disable", True):
            return iterable
        from tqdm.auto import tqdm
        return tqdm(iterable, total=total, **{k: v for k, v in self._progress_bar_config.items() if k != "disable"})

    numpy_to_pil = staticmethod(numpy_to_pil)

    @classmethod
    def from_pretrained(cls, path, compute_dtype="bf16", **kwargs):
        """diffusers folder: ``vae/``, ``unet/``, ``scheduler/``, ``class_embedding/`` (``model_index.json`` records the
        custom classes by module path)."""
        import os
        from .checkpoint import load_weights_file
        from .sd_unet import CustomEmbedding, SDUNet2DConditionModel
        from .vae import AutoencoderKL
        vae = kwargs.pop("vae", None) or AutoencoderKL.from_pretrained(os.path.join(path, "vae"), compute_dtype=compute_dtype)
        unet = kwargs.pop("unet", None) or SDUNet2DConditionModel.from_pretrained(os.path.join(path, "unet"), compute_dtype=compute_dtype)
        scheduler = kwargs.pop("scheduler", None) or DDIMScheduler.from_pretrained(os.path.join(path, "scheduler"))
        emb = kwargs.pop("class_embedding", None)
        if emb is None:
            sd = load_weights_file(os.path.join(path, "class_embedding"))
            n, d = sd["inner_module.weight"].shape
            emb = CustomEmbedding(n, d)
            emb.load_state_dict(sd)
        return cls(vae=vae, unet=unet, scheduler=scheduler, class_embedding=emb)

    def save_pretrained(self, path, safe_serialization=True):
        import json
        import os
        from .checkpoint import save_scheduler, save_weights_file
        os.makedirs(path, exist_ok=True)
        index = {"_class_name": "CustomStableDiffusionImg2ImgPipeline", "_diffusers_version": "0.18.2",
                 "vae": ["diffusers", "AutoencoderKL"], "unet": ["diffusers", "UNet2DConditionModel"],
                 "scheduler": ["diffusers", type(self.scheduler).__name__],
                 "class_embedding": ["src.custom_embedding.custom_embedding", "CustomEmbedding"]}
        with open(os.path.join(path, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)
        self.vae.save_pretrained(os.path.join(path, "vae"), safe_serialization)
        self.unet.save_pretrained(os.path.join(path, "unet"), safe_serialization)
        save_scheduler(self.scheduler, os.path.join(path, "scheduler"))
        folder = os.path.join(path, "class_embedding")
        os.makedirs(folder, exist_ok=True)
        emb = self.class_embedding.inner_module
        with open(os.path.join(folder, "config.json"), "w") as f:
            json.dump({"_class_name": "CustomEmbedding", "num_classes": emb.num_embeddings,
                       "class_embedding_dim": emb.embedding_dim}, f, indent=2)
        save_weights_file(self.class_embedding.state_dict(), folder, safe_serialization)

    # ---- class conditioning rows (replaces `_encode_class`, custom_pipeline...:221-281) ------------------------------------
    @staticmethod
    def _label_rows(class_labels):
        """int | list[int] | 1-D tensor -> int64 tensor [B] (None stays None)."""
        if class_labels is None or isinstance(class_labels, torch.Tensor):
            return None if class_labels is None else class_labels.long()
        return torch.as_tensor([class_labels] if isinstance(class_labels, int) else class_labels).long()

    def _encode_class(self, class_labels, device, do_classifier_free_guidance, class_labels_embeds=None, lora_scale=None):
        rows = class_labels_embeds
        if rows is None:
            labels = self._label_rows(class_labels)
            rows = self.class_embedding(labels.to(self.class_embedding.inner_module.weight.device))
        rows = rows.to(dtype=torch.float32, device=device)
        if not do_classifier_free_guidance:
            return rows
        # the unconditional half of a guided batch is the all-zero embedding, stacked in front
        return torch.cat([rows.new_zeros((rows.shape[0], self.unet.config.cross_attention_dim)), rows])

    # ---- :283-303 --------------------------------------------------------------------------------------------------------
    def prepare_extra_step_kwargs(self, generator, eta):
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    # ---- argument contract (replaces `check_inputs`, custom_pipeline...:305-373): the messages are the interface --------------
    def check_inputs(self, class_labels, strength, callback_steps, class_labels_embeds, latent_shape, image, guidance_scale):
        have_labels, have_rows = class_labels is not None, class_labels_embeds is not None
        rules = (   # (violated?, message), checked in order
            (image is None and latent_shape is None, "Either `image` or `latent_shape` must be provided as input."),
            (not 0 <= strength <= 1, f"The value of strength should be in [0, 1] but is {strength}"),
            (not (isinstance(callback_steps, int) and callback_steps > 0),
             f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}."),
            (have_labels and have_rows,
             "Cannot forward both `class_labels` and `class_labels_embeds`. Please make sure to only forward one of the two."),
            (not have_labels and not have_rows,
             "Provide either `class_labels` or `class_labels_embeds`. Cannot leave both `class_labels` and "
             "`class_labels_embeds` undefined."),
            (have_labels and not isinstance(class_labels, (int, list, torch.Tensor)),
             f"`class_labels` has to be of type `int` or `list` or `torch.Tensor` but is {type(class_labels)}"),
            (isinstance(class_labels, torch.Tensor) and class_labels.ndim != 1, "If a Tensor `class_labels` should be 1D"),
            (guidance_scale is not None and not isinstance(guidance_scale, (float, int, torch.Tensor)),
             f"`guidance_scale` has to be of type `int` or `float` or `Tensor` or `None` but is {type(guidance_scale)}"),
        )
        for violated, message in rules:
            if violated:
                raise ValueError(message)
        assert not isinstance(guidance_scale, torch.Tensor) or guidance_scale.ndim == 1, "If a Tensor `guidance_scale` should be 1D"
        if image is None and strength != 1:
            warn("`image` is None so the generation will start from pure Gaussian noise, but `strength` is not set to 1 "
                 "so the denoising process will not run for the full denoising trajectory. This will produce images "
                 "that are not fully denoised.")

    # ---- the last `strength` fraction of the schedule (replaces `get_timesteps`, custom_pipeline...:375-383) ---------------
    def get_timesteps(self, num_inference_steps, strength, device=None):
        kept = min(int(num_inference_steps * strength), num_inference_steps)
        first = max(num_inference_steps - kept, 0)
        return self.scheduler.timesteps[first * self.scheduler.order:], num_inference_steps - first

    @staticmethod
    def _randn(shape, generator, device):
        return randn_tensor(shape, generator, device)

    def prepare_latents(self, image, timestep, batch_size, dtype, device, latent_shape, generator, add_forward_noise_to_image):
        if image is not None and not isinstance(image, (torch.Tensor, list)):
            raise ValueError(f"`image` has to be of type `torch.Tensor`, `PIL.Image.Image`, list, or `None`, but is {type(image)}")
        if image is None:
            return torch.randn(tuple(latent_shape), device=device, dtype=torch.float32)
        image = image.to(device=device, dtype=torch.float32)
        if image.shape[1] == 4:
            init_latents = image                              # already latents ("ugly hardcoded test", :414-417)
        else:
            sf = float(self.vae.config.scaling_factor)
            if isinstance(generator, list):
                n = image.shape[0]
                if len(generator) != n:
                    raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective "
                                     f"batch size of {n}. Make sure the batch size matches the length of the generators.")
                dist = self.vae.encode(image).latent_dist     # one encode; per-image draws like the reference's loop
                shape = (1,) + tuple(dist.mean.shape[1:])
                noise = torch.cat([self._randn(shape, g, device) for g in generator], 0)
                init_latents = dist.sample(noise=noise, scale=sf)
            else:
                init_latents = self.vae.encode(image).latent_dist.sample(generator, scale=sf)    # scaling_factor folded in
        if add_forward_noise_to_image:
            noise = self._randn(init_latents.shape, generator, device)
            init_latents = self.scheduler.add_noise(init_latents, noise, timestep)
        return init_latents

    # ---- :447-730 --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image: Optional[torch.Tensor] = None, latent_shape: Optional[Tuple[int, ...]] = None,
                 class_labels: Optional[Union[int, List[int], torch.Tensor]] = None, strength: float = 0.8,
                 add_forward_noise_to_image: bool = True, num_inference_steps: Optional[int] = DEFAULT_NUM_INFERENCE_STEPS,
                 guidance_scale: Optional[Union[float, torch.Tensor]] = None, eta: Optional[float] = 0.0,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 class_labels_embeds: Optional[torch.Tensor] = None, output_type: str = "pil",
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, device=None):
        self.check_inputs(class_labels=class_labels, strength=strength, callback_steps=callback_steps,
                          class_labels_embeds=class_labels_embeds, latent_shape=latent_shape, image=image,
                          guidance_scale=guidance_scale)
        if class_labels is not None and isinstance(class_labels, int):
            batch_size = 1
        elif class_labels is not None and isinstance(class_labels, list):
            batch_size = len(class_labels)
        elif class_labels is not None and isinstance(class_labels, torch.Tensor):
            batch_size = class_labels.shape[0]                # the reference keeps the torch.Size; `.repeat(Size)` == `.repeat(n)`
        else:
            batch_size = class_labels_embeds.shape[0]
        device = self._execution_device
        if isinstance(guidance_scale, torch.Tensor):
            do_cfg = True
        else:
            do_cfg = guidance_scale is not None and guidance_scale > 1.0
        lora_scale = cross_attention_kwargs.get("scale", None) if cross_attention_kwargs is not None else None
        embeds = self._encode_class(class_labels=class_labels, device=device, do_classifier_free_guidance=do_cfg,
                                    class_labels_embeds=class_labels_embeds, lora_scale=lora_scale)
        ehs = hack_class_embedding(embeds)
        if image is not None:
            image = self.image_processor.preprocess(image)
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength, device)
        latent_timestep = timesteps[:1].repeat(batch_size)
        latents = self.prepare_latents(image=image, timestep=latent_timestep, batch_size=batch_size, dtype=torch.float32,
                                       device=device, latent_shape=latent_shape, generator=generator,
                                       add_forward_noise_to_image=add_forward_noise_to_image)
        latents = latents.contiguous()
        extra = self.prepare_extra_step_kwargs(generator, eta)
        step_gen = generator
        B = latents.shape[0]
        two = torch.empty((2 * B,) + tuple(latents.shape[1:]), dtype=torch.float32, device=device) if do_cfg else None
        num_warmup_steps = len(timesteps) - num_inference_steps * self.scheduler.order
        for i, t in enumerate(timesteps):
            if do_cfg:
                two[:B].copy_(latents)
                two[B:].copy_(latents)
                pred = self.unet(two, t, encoder_hidden_states=ehs, cross_attention_kwargs=cross_attention_kwargs, return_dict=False)[0]
                # noise_pred_uncond + guidance_scale * (noise_pred_cond - noise_pred_uncond), fused into the scheduler update
                latents, _ = self.scheduler._device_step(pred[B:], t, latents, extra.get("eta", 0.0) or 0.0, False, step_gen, None,
                                                         uncond_output=pred[:B], w=guidance_scale, guidance_cfg=False, want_x0=False)
            else:
                pred = self.unet(latents, t, encoder_hidden_states=ehs, cross_attention_kwargs=cross_attention_kwargs, return_dict=False)[0]
                latents = self.scheduler.step(pred, t, latents, **extra, return_dict=False)[0]
            if i == len(timesteps) - 1 or ((i + 1) > num_warmup_steps and (i + 1) % self.scheduler.order == 0):
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)
        if output_type != "latent":
            out = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]
        else:
            out = latents
        out = self.image_processor.postprocess(out, output_type=output_type.removesuffix("+latent"),
                                               do_denormalize=[True] * out.shape[0])
        if "+latent" in output_type:
            return out, latents
        return out
