"""Oracle (test infrastructure): ``CustomStableDiffusionImg2ImgPipeline`` and the latent-diffusion branches of the class
transfer loops, CPU fp32 over the oracle's own UNet2DConditionRef / AutoencoderKLRef / DDIM schedulers.

Parity unpinned (see ``oracle/__init__.py``).  Follows
``src/custom_pipeline_stable_diffusion_img2img/custom_pipeline_stable_diffusion_img2img.py:60-145`` (constructor: scheduler
config fix-ups, ``vae_scale_factor``), ``:221-281`` (``_encode_class``), ``:375-383`` (``get_timesteps``), ``:385-445``
(``prepare_latents``), ``:571-730`` (``__call__``) and ``src/utils_Img2Img.py:179-187`` (``hack_class_embedding``),
``:566-612`` (``_ddib``), ``:615-648`` (``_classifier_free_guidance_forward_start``), ``:763-800`` (``_inversion``),
``:803-847`` (``_LDM_preprocess`` / ``_encode_to_latents`` / ``_decode_to_images``).
"""
from __future__ import annotations

import torch

from .schedulers_ref import DDIMInverseSchedulerRef, DDIMSchedulerRef
from .vae_ref import vae_postprocess_ref, vae_preprocess_ref


def hack_class_embedding_ref(cl_embed: torch.Tensor) -> torch.Tensor:
    """``utils_Img2Img.py:179-187``: (N, E) -> (N, 77, E), 76 zero tokens behind the class token."""
    assert cl_embed.ndim == 2
    bs, ed = cl_embed.shape
    cl_embed = cl_embed.reshape(bs, 1, ed)
    return torch.cat([cl_embed, torch.zeros_like(cl_embed).repeat(1, 76, 1)], dim=1)


class SDImg2ImgPipelineRef:
    def __init__(self, vae, unet, scheduler, class_embedding):
        cfg = dict(vars(scheduler.config))
        if cfg.get("steps_offset", 1) != 1:          # :74-91
            cfg["steps_offset"] = 1
        if cfg.get("clip_sample", False) is True:    # :93-109
            cfg["clip_sample"] = False
        self.scheduler = DDIMSchedulerRef.from_config(cfg)
        self.vae, self.unet, self.class_embedding = vae, unet, class_embedding
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)

    def _encode_class(self, class_labels, do_classifier_free_guidance, class_labels_embeds=None):
        if class_labels is not None:
            if isinstance(class_labels, int):
                class_labels = torch.tensor([class_labels]).long()
            elif isinstance(class_labels, list):
                class_labels = torch.tensor(class_labels).long()
            else:
                class_labels = class_labels.long()
            batch_size = class_labels.shape[0]
        else:
            batch_size = class_labels_embeds.shape[0]
        if class_labels_embeds is None:
            class_labels_embeds = self.class_embedding(class_labels)
        if do_classifier_free_guidance:
            uncond = torch.zeros((batch_size, self.unet.config.cross_attention_dim))
            class_labels_embeds = torch.cat([uncond, class_labels_embeds])
        return class_labels_embeds

    def get_timesteps(self, num_inference_steps, strength):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start:], num_inference_steps - t_start

    def prepare_latents(self, image, timestep, latent_shape, generator, add_forward_noise_to_image):
        if image is None:
            return torch.randn(latent_shape, dtype=torch.float32)
        if image.shape[1] == 4:
            init_latents = image
        else:
            init_latents = self.vae.encode(image).latent_dist.sample(generator)
            init_latents = self.vae.config.scaling_factor * init_latents
        if add_forward_noise_to_image:
            noise = torch.randn(init_latents.shape, generator=generator, dtype=torch.float32)
            init_latents = self.scheduler.add_noise(init_latents, noise, timestep)
        return init_latents

    @torch.no_grad()
    def __call__(self, image=None, latent_shape=None, class_labels=None, strength=0.8, add_forward_noise_to_image=True,
                 num_inference_steps=50, guidance_scale=None, eta=0.0, generator=None, class_labels_embeds=None,
                 output_type="np"):
        if class_labels is not None and isinstance(class_labels, int):
            batch_size = 1
        elif class_labels is not None and isinstance(class_labels, list):
            batch_size = len(class_labels)
        elif class_labels is not None:
            batch_size = class_labels.shape[0]
        else:
            batch_size = class_labels_embeds.shape[0]
        if isinstance(guidance_scale, torch.Tensor):
            do_cfg = True
        else:
            do_cfg = guidance_scale is not None and guidance_scale > 1.0
        ehs = hack_class_embedding_ref(self._encode_class(class_labels, do_cfg, class_labels_embeds))
        if image is not None:
            image = vae_preprocess_ref(image)
        self.scheduler.set_timesteps(num_inference_steps)
        timesteps, num_inference_steps = self.get_timesteps(num_inference_steps, strength)
        latent_timestep = timesteps[:1].repeat(batch_size)
        latents = self.prepare_latents(image, latent_timestep, latent_shape, generator, add_forward_noise_to_image)
        if isinstance(guidance_scale, torch.Tensor):
            guidance_scale = guidance_scale.view(guidance_scale.shape[0], 1, 1, 1)
        for t in timesteps:
            inp = torch.cat([latents] * 2) if do_cfg else latents
            noise_pred = self.unet(inp, t, encoder_hidden_states=ehs, return_dict=False)[0]
            if do_cfg:
                uncond, cond = noise_pred.chunk(2)
                noise_pred = uncond + guidance_scale * (cond - uncond)
            latents = self.scheduler.step(noise_pred, t, latents, eta=eta, generator=generator, return_dict=False)[0]
        if output_type != "latent":
            out = self.vae.decode(latents / self.vae.config.scaling_factor, return_dict=False)[0]
        else:
            out = latents
        out = vae_postprocess_ref(out, output_type.removesuffix("+latent"))
        return (out, latents) if "+latent" in output_type else out


@torch.no_grad()
def encode_to_latents_ref(pipe, images, generator=None):
    return pipe.vae.encode(images).latent_dist.sample(generator) * pipe.vae.config.scaling_factor


@torch.no_grad()
def sd_inversion_ref(pipe, latents, ehs, num_inference_steps, variant="0.18.2"):
    """``_inversion`` with an SD pipeline: ``pipe.unet(gauss, t, class_labels)`` passes the 77-token embedding positionally
    as ``encoder_hidden_states``."""
    gauss = latents.clone()
    inv = DDIMInverseSchedulerRef.from_config(pipe.scheduler.config, variant=variant)
    inv.set_timesteps(num_inference_steps)
    for t in inv.timesteps:
        out = pipe.unet(gauss, t, ehs).sample
        gauss = inv.step(out, t, gauss).prev_sample
    return gauss


@torch.no_grad()
def sd_ddib_ref(pipe, clean_images, orig_class_labels, target_class_labels, num_inference_steps, generator=None,
                variant="0.18.2", output_type="np"):
    """``_ddib``, ``CustomStableDiffusionImg2ImgPipeline`` branch: VAE-encode, invert under the original class embedding,
    denoise under the target class (``strength=1``, no forward noise, ``guidance_scale=0``), VAE-decode."""
    latents = encode_to_latents_ref(pipe, clean_images, generator)
    ehs = hack_class_embedding_ref(pipe._encode_class(orig_class_labels, False))
    inverted = sd_inversion_ref(pipe, latents, ehs, num_inference_steps, variant)
    out = pipe(image=inverted, class_labels=target_class_labels, strength=1, add_forward_noise_to_image=False,
               num_inference_steps=num_inference_steps, guidance_scale=0, output_type=output_type)
    return out, inverted, latents


@torch.no_grad()
def sd_cfg_forward_start_ref(pipe, clean_images, target_class_labels, guidance_scale, frac_diffusion_skipped,
                             num_inference_steps, generator=None, output_type="np"):
    """``_classifier_free_guidance_forward_start``, SD branch: ``strength = frac_diffusion_skipped``."""
    return pipe(image=clean_images, class_labels=target_class_labels, strength=frac_diffusion_skipped,
                num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, generator=generator,
                output_type=output_type)


def sd_linear_interp_custom_guidance_inverted_start_ref(pipe, clean_images, orig_class_labels, target_class_labels, p,
                                                        guidance_loss_scale, num_inference_steps, generator=None, variant="0.18.2"):
    """``_linear_interp_custom_guidance_inverted_start``, ``CustomStableDiffusionImg2ImgPipeline`` branch
    (utils_Img2Img.py:651-696): ``_LDM_preprocess`` (VAE-encode, both label tensors -> 77-token embeddings), inversion under the
    original class embedding, gradient-guided generation IN LATENT SPACE under the target class embedding
    (``pipe.unet(images, t, target_class_embeds)``, :718-726; the Lp target is the inverted latent), ``_decode_to_images`` and the
    min-max renormalisation to [-1, 1] (:689-695).  Returns (images in [-1, 1], guided latents, inverted latents)."""
    from .pipeline_ref import custom_guided_generation_ref
    with torch.no_grad():
        latents = encode_to_latents_ref(pipe, clean_images, generator)
        ehs_orig = hack_class_embedding_ref(pipe._encode_class(orig_class_labels, False))
        ehs_target = hack_class_embedding_ref(pipe._encode_class(target_class_labels, False))
        inverted = sd_inversion_ref(pipe, latents, ehs_orig, num_inference_steps, variant)
    guided = custom_guided_generation_ref(pipe, inverted, ehs_target, p, guidance_loss_scale, num_inference_steps)
    with torch.no_grad():
        image = pipe.vae.decode(guided / pipe.vae.config.scaling_factor, return_dict=False)[0]
        image = image - image.min()
        image = image / image.max()
        image = image * 2 - 1
    return image, guided, inverted
