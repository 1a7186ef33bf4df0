"""Oracle (test infrastructure): conditional DDIM pipeline + DDIB class transfer, CPU fp32.

Parity unpinned (see ``oracle/__init__.py``).  Follows
``src/pipeline_conditional_ddim/pipeline_conditionial_ddim.py:139-361`` (``__call__``),
``src/utils_Img2Img.py:763-800`` (``_inversion``) and ``:566-612`` (``_ddib``).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from .schedulers_ref import DDIMInverseSchedulerRef, DDIMSchedulerRef


class ConditionalDDIMPipelineRef:
    def __init__(self, unet, scheduler):
        # pipeline_conditionial_ddim.py:45 -- always re-made as a DDIM scheduler from the config
        self.unet = unet
        self.scheduler = DDIMSchedulerRef.from_config(scheduler.config)

    @torch.no_grad()
    def __call__(self, class_labels, class_emb=None, w=None, generator=None, eta=0.0, num_inference_steps=50,
                 use_clipped_model_output=None, output_type="numpy", start_image=None,
                 add_forward_noise_to_image=True, frac_diffusion_skipped=None, guidance_eqn="imagen"):
        assert (frac_diffusion_skipped is None) == (start_image is None)  # :125-127
        if num_inference_steps is None:
            num_inference_steps = 50
        bs = class_labels.shape[0] if class_labels is not None else class_emb.shape[0]
        ss = self.unet.config.sample_size
        shape = (bs, self.unet.config.in_channels, ss, ss) if isinstance(ss, int) else (bs, self.unet.config.in_channels, *ss)
        if start_image is not None:  # :237-245
            image = start_image
        else:
            image = torch.randn(shape, generator=generator, dtype=torch.float32)
        self.scheduler.set_timesteps(num_inference_steps)  # :248
        if frac_diffusion_skipped is not None:  # :250-258
            init_t = self.scheduler.config.num_train_timesteps * (1 - frac_diffusion_skipped)
            timesteps = self.scheduler.timesteps[self.scheduler.timesteps <= init_t]
        else:
            timesteps = self.scheduler.timesteps
        if add_forward_noise_to_image:  # :263-269
            noise = torch.randn(image.shape, generator=generator, dtype=image.dtype)
            image = self.scheduler.add_noise(image, noise, timesteps[0].repeat(bs))
        do_cfg = (isinstance(w, torch.Tensor)  # :272-284
                  or (guidance_eqn == "imagen" and isinstance(w, (int, float)) and w > 1)
                  or (guidance_eqn == "CFG" and isinstance(w, (int, float)) and w > 0))
        for t in timesteps:  # :286-347
            cond = self.unet(sample=image, timestep=t, class_labels=class_labels, class_emb=class_emb).sample
            if do_cfg:
                uncond = self.unet(sample=image, timestep=t, class_labels=None,
                                   class_emb=torch.zeros((bs, self.unet.time_embed_dim))).sample
                if isinstance(w, torch.Tensor):
                    w = w.view(-1, 1, 1, 1)
                if guidance_eqn == "imagen":
                    guided = uncond + w * (cond - uncond)
                elif guidance_eqn == "CFG":
                    guided = cond + w * (cond - uncond)
                else:
                    raise ValueError(guidance_eqn)
            else:
                guided = cond
            image = self.scheduler.step(guided, t, image, eta=eta, use_clipped_model_output=use_clipped_model_output,
                                        generator=generator).prev_sample
        image = (image / 2 + 0.5).clamp(0, 1)  # :349
        image = image.cpu().permute(0, 2, 3, 1).numpy()  # :350
        return SimpleNamespace(images=image)


@torch.no_grad()
def inversion_ref(pipe, input_images, class_labels, num_inference_steps, variant="0.18.2"):
    """``_inversion`` (utils_Img2Img.py:763-800)."""
    gauss = input_images.clone().detach()
    inv = DDIMInverseSchedulerRef.from_config(pipe.scheduler.config, variant=variant)
    inv.set_timesteps(num_inference_steps)
    for t in inv.timesteps:
        out = pipe.unet(gauss, t, class_labels).sample
        gauss = inv.step(out, t, gauss).prev_sample
    return gauss


@torch.no_grad()
def ddib_ref(pipe, clean_images, orig_class_labels, target_class_labels, num_inference_steps, variant="0.18.2"):
    """``_ddib`` for the ConditionalDDIMPipeline branch (utils_Img2Img.py:566-599)."""
    inverted = inversion_ref(pipe, clean_images, orig_class_labels, num_inference_steps, variant)
    images = pipe(class_labels=target_class_labels, w=0, num_inference_steps=num_inference_steps,
                  start_image=inverted, add_forward_noise_to_image=False, frac_diffusion_skipped=0).images
    return images, inverted


def lp_loss_ref(x, y, p=2):
    """``Lp_loss`` (utils_Img2Img.py:245-270)."""
    return torch.linalg.vector_norm(x - y, dim=(1, 2, 3), ord=p)


def custom_guided_generation_ref(pipe, input_images, target_class_labels, p, guidance_loss_scale, num_inference_steps):
    """``_custom_guided_generation`` (utils_Img2Img.py:699-760), ConditionalDDIMPipeline branch: every step the image is
    pushed down the gradient -- taken THROUGH the UNet -- of ``Lp(x0_pred, input_images)``, then the scheduler steps with the
    model output computed before the push.  (``input_images`` is what the caller passes: the inverted Gaussian.)"""
    images = input_images.clone().detach()
    pipe.scheduler.set_timesteps(num_inference_steps)
    for t in pipe.scheduler.timesteps:
        images = images.detach().requires_grad_()
        with torch.enable_grad():
            model_output = pipe.unet(images, t, target_class_labels).sample
            x0 = pipe.scheduler.step(model_output, t, images).pred_original_sample
            losses = lp_loss_ref(x0, input_images, p)
            guidance_grad = torch.autograd.grad([losses[i] for i in range(len(input_images))], images)[0]
        images = images.detach() - guidance_loss_scale * guidance_grad
        images = pipe.scheduler.step(model_output.detach(), t, images).prev_sample
    return images.detach()


def linear_interp_custom_guidance_inverted_start_ref(pipe, clean_images, orig_class_labels, target_class_labels, p,
                                                     guidance_loss_scale, num_inference_steps, variant="0.18.2"):
    """``_linear_interp_custom_guidance_inverted_start`` (utils_Img2Img.py:651-696): inversion under the original class, then
    guided generation under the target class.  Returns the [-1, 1] image tensor (before ``tensor_to_PIL``)."""
    with torch.no_grad():
        inverted = inversion_ref(pipe, clean_images, orig_class_labels, num_inference_steps, variant)
    return custom_guided_generation_ref(pipe, inverted, target_class_labels, p, guidance_loss_scale, num_inference_steps)


def numpy_to_uint8(images: np.ndarray) -> np.ndarray:
    """``DiffusionPipeline.numpy_to_pil`` quantisation: ``(images * 255).round().astype("uint8")``."""
    return (images * 255).round().astype("uint8")


def tensor_to_uint8_ref(tensor: torch.Tensor, channel="mean") -> np.ndarray:
    """``tensor_to_PIL`` (utils_Img2Img.py:96-150) up to the uint8 NHWC array it hands to ``Image.fromarray``."""
    img = tensor.clone().detach()
    if tensor.shape[1] == 4:
        img -= img.min()
        img /= img.max()
        img = img.clamp(0, 1)
        img = img[:, channel].view(tensor.shape[0], 1, tensor.shape[2], tensor.shape[3]) if isinstance(channel, int) \
            else img.mean(dim=1, keepdim=True)
    elif tensor.shape[1] == 3:
        img = (img / 2 + 0.5).clamp(0, 1)
    return (img.cpu().permute(0, 2, 3, 1).numpy() * 255).round().astype("uint8")
