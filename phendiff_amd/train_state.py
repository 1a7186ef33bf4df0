"""Training-state checkpoints in the layout of ``accelerator.save_state`` (accelerate 0.23), which is what the reference
writes under ``<run>/checkpoints/step_<n>`` (``utils_misc.py:322-347``) and reads back in ``resume_from_checkpoint``
(``utils_training.py:56-94``):

    step_<n>/pytorch_model.bin        unet.state_dict()                     (diffusers parameter names)
    step_<n>/optimizer.bin            torch.optim.AdamW.state_dict()        (parameters indexed in unet.parameters() order)
    step_<n>/scheduler.bin            LambdaLR.state_dict()                 (cosine schedule with warm-up, train.py:298-303)
    step_<n>/random_states_<rank>.pkl python / numpy / torch / torch.cuda RNG states
    step_<n>/pytorch_model_2.bin      class_embedding.state_dict()          (StableDiffusion runs: accelerate numbers the prepared
                                      models unet, vae, class_embedding -- train.py:318-326; the frozen VAE is not the trainer's)
    step_<n>/custom_checkpoint_<k>.pkl diffusers EMAModel.state_dict() per trained module (accelerate's slot for registered objects; the
                                      reference does not register its EMA -- writing it costs nothing and makes a resumed run
                                      continue the same EMA trajectory; a folder without it loads fine)

The engine keeps parameters / Adam moments in ONE flat fp32 buffer ordered by ``training_param_order``; the functions here
translate between that layout and torch's per-parameter dictionaries, so a checkpoint written by the reference's own
``torch.optim.AdamW`` resumes on the engine and vice versa.  Pure host-side IO (no device work).
"""
from __future__ import annotations

import os
import pickle
import random
import shutil
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

ADAMW_GROUP_DEFAULTS = dict(amsgrad=False, foreach=None, maximize=False, capturable=False, differentiable=False, fused=None)


def optimizer_state_dict(param_names: Sequence[str], flat_names: Sequence[str], sizes: Dict[str, torch.Size],
                         exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int, lr: float, betas, eps: float,
                         weight_decay: float, steps_by_name: Optional[Dict[str, int]] = None) -> dict:
    """``torch.optim.AdamW.state_dict()`` built from flat moment buffers.  ``param_names``: ``unet.named_parameters()``
    order (torch's parameter indices); ``flat_names``: order of the flat buffers."""
    off, where = 0, {}
    for n in flat_names:
        k = int(np.prod(sizes[n])) if len(sizes[n]) else 1
        where[n] = (off, k)
        off += k
    state = {}
    if step > 0:
        for i, n in enumerate(param_names):
            o, k = where[n]
            st_n = (steps_by_name or {}).get(n, step)     # a parameter torch skipped on some steps (no gradient) counts its own
            if st_n <= 0:
                continue
            state[i] = {"step": torch.tensor(float(st_n)),
                        "exp_avg": exp_avg[o:o + k].detach().cpu().reshape(sizes[n]).clone(),
                        "exp_avg_sq": exp_avg_sq[o:o + k].detach().cpu().reshape(sizes[n]).clone()}
    group = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, **ADAMW_GROUP_DEFAULTS,
                 params=list(range(len(param_names))))
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(sd: dict, param_names: Sequence[str], flat_names: Sequence[str], sizes: Dict[str, torch.Size],
                              exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, steps_out: Optional[Dict[str, int]] = None) -> int:
    """Inverse of :func:`optimizer_state_dict`: fills the flat moment buffers in place, returns the step count (the largest
    per-parameter count; ``steps_out`` receives every parameter's own)."""
    off, where = 0, {}
    for n in flat_names:
        k = int(np.prod(sizes[n])) if len(sizes[n]) else 1
        where[n] = (off, k)
        off += k
    idx = sd["param_groups"][0]["params"]
    if len(idx) != len(param_names):
        raise ValueError(f"optimizer state has {len(idx)} parameters, the model has {len(param_names)}")
    step = 0
    exp_avg.zero_()
    exp_avg_sq.zero_()
    for pos, n in zip(idx, param_names):
        st = sd["state"].get(pos)
        if st is None:
            continue
        o, k = where[n]
        if tuple(st["exp_avg"].shape) != tuple(sizes[n]):
            raise ValueError(f"optimizer state of {n}: shape {tuple(st['exp_avg'].shape)} != {tuple(sizes[n])}")
        exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1).to(exp_avg.device, torch.float32))
        exp_avg_sq[o:o + k].copy_(st["exp_avg_sq"].reshape(-1).to(exp_avg.device, torch.float32))
        step = max(step, int(float(st["step"])))
        if steps_out is not None:
            steps_out[n] = int(float(st["step"]))
    return step


def lr_scheduler_state_dict(base_lr: float, last_epoch: int, current_lr: float) -> dict:
    """``LambdaLR.state_dict()`` of diffusers ``get_scheduler("cosine")`` after ``last_epoch`` ``step()`` calls."""
    return {"base_lrs": [base_lr], "last_epoch": int(last_epoch), "verbose": False, "_step_count": int(last_epoch) + 1,
            "_get_lr_called_within_step": False, "_last_lr": [current_lr], "lr_lambdas": [None]}


def ema_state_dict(shadow_flat: torch.Tensor, flat_names: Sequence[str], param_names: Sequence[str], sizes, optimization_step: int,
                   decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=True, inv_gamma=1.0, power=0.75) -> dict:
    """diffusers ``EMAModel.state_dict()`` (0.18.2): shadow parameters as a list in ``parameters()`` order."""
    off, where = 0, {}
    for n in flat_names:
        k = int(np.prod(sizes[n])) if len(sizes[n]) else 1
        where[n] = (off, k)
        off += k
    shadow = [shadow_flat[where[n][0]:where[n][0] + where[n][1]].detach().cpu().reshape(sizes[n]).clone() for n in param_names]
    return {"decay": decay, "min_decay": min_decay, "optimization_step": int(optimization_step),
            "update_after_step": update_after_step, "use_ema_warmup": use_ema_warmup, "inv_gamma": inv_gamma, "power": power,
            "shadow_params": shadow}


def random_states() -> dict:
    st = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(), "torch_manual_seed": torch.get_rng_state()}
    if torch.cuda.is_available():
        st["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    return st


def restore_random_states(st: dict):
    random.setstate(st["random_state"])
    np.random.set_state(st["numpy_random_seed"])
    torch.set_rng_state(st["torch_manual_seed"])
    if "torch_cuda_manual_seed" in st and torch.cuda.is_available():
        try:
            torch.cuda.set_rng_state_all(st["torch_cuda_manual_seed"])
        except (RuntimeError, IndexError):        # saved on a different number of devices
            pass


def _checkpoint_modules(trainer):
    """[(file index, module, flat-name prefix)] in the order accelerate numbers the prepared models (``pytorch_model.bin``,
    ``pytorch_model_1.bin``, ...; train.py:311-326).  A trainer with more than the UNet says so itself."""
    f = getattr(trainer, "checkpoint_modules", None)
    return f() if f is not None else [(0, trainer.model, "")]


def _model_file(i: int) -> str:
    return "pytorch_model.bin" if i == 0 else f"pytorch_model_{i}.bin"


def _optimizer_names(trainer, mods):
    """Parameter names in the order torch's optimizer indexes them (``params_to_optimize``, train.py:268-272: the trained
    modules' ``parameters()`` one after the other), restricted to what the trainer actually optimises."""
    names = []
    for _, mod, prefix in mods:
        names += [prefix + n for n, _ in mod.named_parameters() if prefix + n in trainer.params]
    missing = [n for n in trainer.params if n not in set(names)]
    if missing:
        raise KeyError(f"trained parameters without a checkpoint module: {missing[:5]}")
    return names


def save_state(trainer, output_dir: str, rank: int = 0, base_lr: Optional[float] = None, lr_step: Optional[int] = None,
               current_lr: Optional[float] = None):
    """``accelerator.save_state(output_dir)`` for a :class:`phendiff_amd.unet_train.UNetTrainer` / ``SDUNetTrainer``: every
    trained module's weights, the Adam moments of EVERY optimised parameter (the ``CustomEmbedding`` included, with its own step
    count) and one EMA file per trained module."""
    os.makedirs(output_dir, exist_ok=True)
    opt = trainer.opt
    mods = _checkpoint_modules(trainer)
    pnames = _optimizer_names(trainer, mods)
    fnames = list(trainer.params)
    sizes = {n: trainer.params[n].shape for n in fnames}
    for i, mod, _ in mods:
        torch.save({k: v.detach().cpu() for k, v in mod.state_dict().items()}, os.path.join(output_dir, _model_file(i)))
    torch.save(optimizer_state_dict(pnames, fnames, sizes, opt.exp_avg, opt.exp_avg_sq, opt.t, opt.lr, opt.betas, opt.eps, opt.wd,
                                    # a frozen parameter (requires_grad_(False), train.py:189-220) never gets a gradient: torch.optim.AdamW
                                    # keeps NO state entry for it, and neither does this file
                                    {**{n: opt.t_tail for n in getattr(opt, "tail_names", ())}, **{n: 0 for n in getattr(trainer, "frozen", ())}}),
               os.path.join(output_dir, "optimizer.bin"))
    torch.save(lr_scheduler_state_dict(base_lr if base_lr is not None else opt.lr, lr_step if lr_step is not None else opt.t,
                                       current_lr if current_lr is not None else opt.lr), os.path.join(output_dir, "scheduler.bin"))
    if getattr(opt, "scaler", None) is not None:          # --mixed_precision fp16: accelerate writes the GradScaler's state_dict as scaler.pt
        torch.save(opt.scaler.state_dict(), os.path.join(output_dir, "scaler.pt"))
    with open(os.path.join(output_dir, f"random_states_{rank}.pkl"), "wb") as f:
        pickle.dump(random_states(), f)
    if opt.ema is not None:
        for slot, (_, mod, prefix) in enumerate(mods):       # one EMAModel per trained module (train.py:224-241)
            mnames = [prefix + n for n, _ in mod.named_parameters() if prefix + n in trainer.params]
            torch.save(ema_state_dict(opt.ema, fnames, mnames, sizes, getattr(opt, "t_ema", opt.t), **{k: v for k, v in opt.ema_kwargs.items()}),
                       os.path.join(output_dir, f"custom_checkpoint_{slot}.pkl"))


def load_state(trainer, input_dir: str, rank: int = 0) -> dict:
    """``accelerator.load_state(input_dir)``: parameters, Adam moments + step counts, RNG states and (when present) the EMA
    shadows.  Returns the saved LR-scheduler state (the caller owns the schedule: ``training.cosine_lr_factor``)."""
    opt = trainer.opt
    mods = _checkpoint_modules(trainer)
    pnames = _optimizer_names(trainer, mods)
    fnames = list(trainer.params)
    sizes = {n: trainer.params[n].shape for n in fnames}
    for i, mod, _ in mods:
        path = os.path.join(input_dir, _model_file(i))
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: the checkpoint lacks the weights of {type(mod).__name__}")
        sd = torch.load(path, map_location="cpu")
        own = dict(mod.named_parameters())
        missing = [k for k in own if k not in sd]
        if missing:
            raise KeyError(f"{_model_file(i)} misses parameters: {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        with torch.no_grad():
            for k, p in own.items():
                p.data.copy_(sd[k].to(p.device, p.dtype))        # in place: the flat buffer and every plan stay valid
    osd = torch.load(os.path.join(input_dir, "optimizer.bin"), map_location="cpu")
    steps = {}
    opt.t = load_optimizer_state_dict(osd, pnames, fnames, sizes, opt.exp_avg, opt.exp_avg_sq, steps)
    if getattr(opt, "tail_names", ()):
        opt.t_tail = max([steps.get(n, 0) for n in opt.tail_names])
    g = osd["param_groups"][0]
    opt.lr, opt.betas, opt.eps, opt.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
    spath = os.path.join(input_dir, "scaler.pt")
    if getattr(opt, "scaler", None) is not None and os.path.exists(spath):      # (a bf16 / fp32 trainer ignores an fp16 run's scaler.pt)
        opt.scaler.load_state_dict(torch.load(spath, map_location="cpu"))
    if opt.ema is not None:
        off, o = {}, 0
        for n in fnames:
            off[n] = o
            o += trainer.params[n].numel()
        for slot, (_, mod, prefix) in enumerate(mods):
            mnames = [prefix + n for n, _ in mod.named_parameters() if prefix + n in trainer.params]
            ema_path = os.path.join(input_dir, f"custom_checkpoint_{slot}.pkl")
            if os.path.exists(ema_path):
                esd = torch.load(ema_path, map_location="cpu")
                # EMAModel.optimization_step = optimizer steps + skipped fp16 steps (EMAModel.step runs on every sync step,
                # utils_training.py:553-556): the decay schedule continues from the SAVED count.  It can never be behind the optimizer.
                step = int(esd.get("optimization_step", opt.t))
                if step < opt.t:
                    raise ValueError(f"{ema_path}: EMA optimization_step {step} < optimizer step {opt.t}")
                opt.t_ema = step
                for n, t in zip(mnames, esd["shadow_params"]):
                    opt.ema[off[n]:off[n] + t.numel()].copy_(t.reshape(-1).to(opt.ema.device, torch.float32))
            else:                                                # the reference's behaviour: EMA restarts from the weights
                opt.t_ema = opt.t
                for n in mnames:
                    k = trainer.params[n].numel()
                    opt.ema[off[n]:off[n] + k].copy_(opt.flat[off[n]:off[n] + k])
    rs = os.path.join(input_dir, f"random_states_{rank}.pkl")
    if os.path.exists(rs):
        with open(rs, "rb") as f:
            restore_random_states(pickle.load(f))
    trainer.refresh_weights()
    sched = os.path.join(input_dir, "scheduler.bin")
    return torch.load(sched, map_location="cpu") if os.path.exists(sched) else {}


def save_checkpoint(trainer, chckpt_save_path: str, global_step: int, checkpoints_total_limit: Optional[int] = None,
                    is_main_process: bool = True, **kw) -> str:
    """``save_checkpoint`` (utils_misc.py:322-347): ``<path>/step_<global_step>`` + pruning of the oldest folders."""
    folder = os.path.join(chckpt_save_path, f"step_{global_step}")
    if is_main_process:
        save_state(trainer, folder, **kw)
        if checkpoints_total_limit is not None:
            dirs = sorted((d for d in os.listdir(chckpt_save_path) if d.startswith("step_") and d[5:].isdigit()),
                          key=lambda x: int(x.split("_")[1]))
            for d in dirs[:-checkpoints_total_limit] if len(dirs) > checkpoints_total_limit else []:
                shutil.rmtree(os.path.join(chckpt_save_path, d))
    return folder


def latest_checkpoint(chckpt_save_path: str) -> Optional[str]:
    """``resume_from_checkpoint == "latest"`` (utils_training.py:66-77): the ``step_<n>`` folder with the largest n."""
    if not os.path.isdir(chckpt_save_path):
        return None
    dirs: List[str] = sorted((d for d in os.listdir(chckpt_save_path) if d.startswith("step_") and d[5:].isdigit()),
                             key=lambda x: int(x.split("_")[1]))
    return os.path.join(chckpt_save_path, dirs[-1]) if dirs else None


def resume_from_checkpoint(trainer, chckpt_save_path: str, which: str = "latest", num_update_steps_per_epoch: int = 1,
                           gradient_accumulation_steps: int = 1):
    """``resume_from_checkpoint`` (utils_training.py:56-94) -> (first_epoch, resume_step, global_step, lr_scheduler_state)."""
    path = latest_checkpoint(chckpt_save_path) if which == "latest" else os.path.join(chckpt_save_path, os.path.basename(which))
    if path is None or not os.path.isdir(path):
        return 0, 0, 0, {}
    sched = load_state(trainer, path)
    global_step = int(path.split("_")[-1])
    resume_global_step = global_step * gradient_accumulation_steps
    first_epoch = global_step // num_update_steps_per_epoch
    resume_step = resume_global_step % (num_update_steps_per_epoch * gradient_accumulation_steps)
    return first_epoch, resume_step, global_step, sched
