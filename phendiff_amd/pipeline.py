"""``ConditionalDDIMPipeline`` with the reference's call surface
(``src/pipeline_conditional_ddim/pipeline_conditionial_ddim.py:27-361``), running on the HIP engine.

Differences that do not change results: the conditional and unconditional UNet passes of classifier-free
guidance feed one fused ``pd_ddim_step`` (guidance combine + scheduler update in one launch); the final
``(x/2+.5).clamp(0,1)`` + NCHW->NHWC is one ``pd_postproc`` launch before the single D2H copy.
"""
from __future__ import annotations

import ctypes as C
from inspect import signature
from types import SimpleNamespace
from typing import List, Literal, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib as L
from .schedulers import DDIMScheduler, randn_tensor

DEFAULT_NUM_INFERENCE_STEPS = 50


class ImagePipelineOutput(SimpleNamespace):
    """``diffusers.ImagePipelineOutput`` stand-in: ``.images``."""


def numpy_to_pil(images: np.ndarray):
    """``DiffusionPipeline.numpy_to_pil``: float [0,1] NHWC -> list of PIL images (uint8 = round(255 x))."""
    from PIL import Image
    if images.ndim == 3:
        images = images[None, ...]
    images = (images * 255).round().astype("uint8")
    if images.shape[-1] == 1:
        return [Image.fromarray(im.squeeze(), mode="L") for im in images]
    return [Image.fromarray(im) for im in images]


class ConditionalDDIMPipeline:
    def __init__(self, unet, scheduler):
        # pipeline_conditionial_ddim.py:44-47: always converted to a DDIM scheduler
        scheduler = DDIMScheduler.from_config(scheduler.config)
        self._modules = {}
        self.register_modules(unet=unet, scheduler=scheduler)
        self._progress_bar_config = {}

    # ---- DiffusionPipeline protocol pieces the reference touches (train.py:227,269; utils_training.py) ----
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

    def to(self, *args, **kwargs):
        self.unet.to(*args, **kwargs)
        return self

    @classmethod
    def from_pretrained(cls, path, compute_dtype="bf16", **kwargs):
        """``ConditionalDDIMPipeline.from_pretrained(<folder>)`` (utils_models.py:144, DDIM.yaml:3-5)."""
        import os
        from .unet import CustomCondUNet2DModel
        unet = kwargs.pop("unet", None) or CustomCondUNet2DModel.from_pretrained(os.path.join(path, "unet"), compute_dtype=compute_dtype)
        scheduler = kwargs.pop("scheduler", None) or DDIMScheduler.from_pretrained(os.path.join(path, "scheduler"))
        return cls(unet=unet, scheduler=scheduler)

    def save_pretrained(self, path, safe_serialization=True):
        from .checkpoint import save_pipeline
        save_pipeline(self, path, safe_serialization)

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, iterable):
        if self._progress_bar_config.get("disable", True):
            return iterable
        from tqdm.auto import tqdm
        return tqdm(iterable, **{k: v for k, v in self._progress_bar_config.items() if k != "disable"})

    numpy_to_pil = staticmethod(numpy_to_pil)

    # ---- argument contract (replaces `check_inputs`, pipeline_conditionial_ddim.py:91-137): the messages are the interface ----
    def check_inputs(self, class_labels=None, class_emb=None, w=None, generator=None, frac_diffusion_skipped=None,
                     start_image=None) -> None:
        def tensor_of_rank(x, rank):
            return isinstance(x, torch.Tensor) and x.ndim == rank
        asserts = (   # (holds?, message)
            (class_labels is None or tensor_of_rank(class_labels, 1), "class_labels must be a 1D tensor of shape (batch_size,) if not None."),
            (class_emb is None or tensor_of_rank(class_emb, 2), "class_emb must be a 2D tensor of shape (batch_size, emb_dim) if not None."),
            (class_labels is None or class_emb is None, "Cannot pass both class_labels and class_emb."),
        )
        for holds, message in asserts:
            assert holds, message
        batch_size = (class_labels if class_labels is not None else class_emb).shape[0]
        assert w is None or isinstance(w, (float, int)) or (w.ndim == 1 and w.shape[0] == batch_size), \
            "w must be a 1D tensor of shape (batch_size,) if not None and not a single int/float."
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size} through class conditioning. Make sure the batch size matches the length of the generators.")
        assert (frac_diffusion_skipped is None) == (start_image is None), \
            "Either pass both frac_diffusion_skipped and start_image or none of them."
        assert frac_diffusion_skipped is None or (isinstance(frac_diffusion_skipped, (float, int)) and 0 <= frac_diffusion_skipped <= 1), \
            f"frac_diffusion_skipped must be a float (or int) between 0 and 1; got {frac_diffusion_skipped}."

    def _randn(self, shape, generator, device):
        return randn_tensor(shape, generator, device)

    @torch.no_grad()
    def __call__(self, class_labels: Optional[torch.Tensor], class_emb: Optional[torch.Tensor] = None,
                 w: Union[int, float, torch.Tensor, None] = None,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None, eta: float = 0.0,
                 num_inference_steps: int = DEFAULT_NUM_INFERENCE_STEPS, use_clipped_model_output: Optional[bool] = None,
                 output_type: Optional[str] = "pil", return_dict: bool = True, start_image: Optional[torch.Tensor] = None,
                 add_forward_noise_to_image: bool = True, frac_diffusion_skipped: Optional[float] = None,
                 guidance_eqn: Literal["imagen", "CFG"] = "imagen") -> Union[ImagePipelineOutput, Tuple]:
        self.check_inputs(class_labels, class_emb, w, generator, frac_diffusion_skipped, start_image)
        if num_inference_steps is None:
            num_inference_steps = DEFAULT_NUM_INFERENCE_STEPS
        if guidance_eqn not in ("imagen", "CFG"):
            raise ValueError(f"Unknown guidance equation '{guidance_eqn}'; should be 'imagen' or 'CFG'")
        batch_size = class_labels.shape[0] if class_labels is not None else class_emb.shape[0]
        device = self.device
        ss = self.unet.config.sample_size
        cin = self.unet.config.in_channels
        image_shape = (batch_size, cin, ss, ss) if isinstance(ss, int) else (batch_size, cin, *ss)
        if start_image is not None:  # :237-245
            image = start_image.to(device=device, dtype=torch.float32)
        else:
            image = self._randn(image_shape, generator, device)
        self.scheduler.set_timesteps(num_inference_steps)  # :248
        if frac_diffusion_skipped is not None:  # :250-258
            init_timestep = self.scheduler.config.num_train_timesteps * (1 - frac_diffusion_skipped)
            timesteps = self.scheduler.timesteps[self.scheduler.timesteps <= init_timestep]
        else:
            timesteps = self.scheduler.timesteps
        if add_forward_noise_to_image:  # :263-269
            noise = self._randn(image.shape, generator, device)
            image = self.scheduler.add_noise(image, noise, timesteps[0].repeat(batch_size))
        do_cfg = (isinstance(w, torch.Tensor)  # :272-284
                  or (guidance_eqn == "imagen" and isinstance(w, (float, int)) and w > 1)
                  or (guidance_eqn == "CFG" and isinstance(w, (float, int)) and w > 0))
        has_class_emb = "class_emb" in signature(self.unet.forward).parameters  # :288-290
        zeros_emb = None
        for t in self.progress_bar(timesteps):  # :286-347
            if has_class_emb:
                cond = self.unet(sample=image, timestep=t, class_labels=class_labels, class_emb=class_emb).sample
            else:
                assert not do_cfg, "'do_classifier_free_guidance' is True but the denoiser model does not take a 'class_emb' argument"
                cond = self.unet(sample=image, timestep=t, class_labels=class_labels).sample
            uncond = None
            if do_cfg:
                if zeros_emb is None:
                    zeros_emb = torch.zeros((batch_size, self.unet.time_embed_dim), device=device)
                uncond = self.unet(sample=image, timestep=t, class_labels=None, class_emb=zeros_emb).sample
            image, _ = self.scheduler._device_step(
                cond, t, image, eta, bool(use_clipped_model_output), generator,
                None, uncond_output=uncond, w=w, guidance_cfg=(guidance_eqn == "CFG"), want_x0=False)
        # :349-350 -- (image/2+.5).clamp(0,1), NCHW -> NHWC, one D2H copy
        B, Cc, H, W = image.shape
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=image.device)
        img = image.contiguous()
        a = L.PostprocArgs(B=B, C=Cc, H=H, W=W, x=img.data_ptr(), out_f32=out.data_ptr(), out_u8=None)
        L.check(L.lib().pd_postproc(C.byref(a), torch.cuda.current_stream(image.device).cuda_stream), "pd_postproc")
        images = out.cpu().numpy()
        if output_type == "pil":
            images = self.numpy_to_pil(images)
        if not return_dict:
            return (images,)
        return ImagePipelineOutput(images=images)
