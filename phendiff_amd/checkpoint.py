"""diffusers ``save_pretrained`` / ``from_pretrained`` folder layout for the drop-in objects, so that
PhenDiff checkpoints (``utils_training.py:1004-1061`` saves the inference pipeline with ``save_pretrained``;
``utils_models.py:144`` / ``img2img_comparison.py:100`` reload it with ``from_pretrained``) run on this engine:

    <pipeline>/model_index.json
    <pipeline>/unet/config.json + diffusion_pytorch_model.{safetensors,bin}
    <pipeline>/scheduler/scheduler_config.json

diffusers 0.18 writes the UNet2D-style attention weights under their deprecated names (``query/key/value/proj_attn``);
they are mapped to the in-memory names (``to_q/to_k/to_v/to_out.0``) on load, as diffusers itself does.
"""
from __future__ import annotations

import json
import os

import torch

_ATTN_ALIASES = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


def remap_deprecated_attention_keys(sd: dict) -> dict:
    out = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for old, new in _ATTN_ALIASES.items():
                if old in k:
                    k = k.replace(old, new)
        out[k] = v
    return out


def _load_weights(folder: str) -> dict:
    st = os.path.join(folder, "diffusion_pytorch_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    return torch.load(os.path.join(folder, "diffusion_pytorch_model.bin"), map_location="cpu", weights_only=True)


def load_weights_file(folder: str) -> dict:
    """``diffusion_pytorch_model.{safetensors,bin}`` of a diffusers model folder, attention keys in their in-memory names."""
    return remap_deprecated_attention_keys(_load_weights(folder))


def save_weights_file(state_dict: dict, folder: str, safe_serialization: bool = True) -> None:
    sd = {k: v.detach().cpu().contiguous() for k, v in state_dict.items()}
    if safe_serialization:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(folder, "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, os.path.join(folder, "diffusion_pytorch_model.bin"))


def save_unet(unet, folder: str, safe_serialization: bool = True) -> None:
    os.makedirs(folder, exist_ok=True)
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(unet.config).items()}
    cfg.update(_class_name="CustomCondUNet2DModel", _diffusers_version="0.18.2")
    with open(os.path.join(folder, "config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in unet.state_dict().items()}
    if safe_serialization:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(folder, "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, os.path.join(folder, "diffusion_pytorch_model.bin"))


def load_unet(cls, folder: str, compute_dtype: str = "bf16", **overrides):
    with open(os.path.join(folder, "config.json")) as f:
        cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    unet = cls.from_config(cfg, compute_dtype=compute_dtype, **overrides)
    unet.load_state_dict(remap_deprecated_attention_keys(_load_weights(folder)))
    return unet


def save_scheduler(scheduler, folder: str) -> None:
    os.makedirs(folder, exist_ok=True)
    cfg = dict(vars(scheduler.config))
    cfg.update(_class_name=type(scheduler).__name__, _diffusers_version="0.18.2")
    with open(os.path.join(folder, "scheduler_config.json"), "w") as f:
        json.dump(cfg, f, indent=2, sort_keys=True)


def load_scheduler(cls, folder: str, **overrides):
    with open(os.path.join(folder, "scheduler_config.json")) as f:
        cfg = json.load(f)
    return cls.from_config(cfg, **overrides)


def save_pipeline(pipe, folder: str, safe_serialization: bool = True) -> None:
    os.makedirs(folder, exist_ok=True)
    index = {"_class_name": "ConditionalDDIMPipeline", "_diffusers_version": "0.18.2",
             # the reference records its custom classes by module path
             "unet": ["src.cond_unet_2d.cond_unet_2d", "CustomCondUNet2DModel"],
             "scheduler": ["diffusers", "DDIMScheduler"]}
    with open(os.path.join(folder, "model_index.json"), "w") as f:
        json.dump(index, f, indent=2)
    save_unet(pipe.unet, os.path.join(folder, "unet"), safe_serialization)
    save_scheduler(pipe.scheduler, os.path.join(folder, "scheduler"))
