"""Diagnostics over a launch plan's activation buffers (every activation of a plan owns its buffer, so after one evaluation the
whole forward can be inspected): the per-block max |activation| that makes an fp16 overflow a NAMED failure instead of a NaN at
the output (fp16 is BASELINE configs[4]'s dtype: `--mixed_precision fp16`, img2img_comparison.py:56-59)."""
import math

import torch

FP16_MAX = 65504.0


def activation_absmax(plan) -> dict:
    """{"<index>.<kind>.<block name>.<tensor>": max |x|} over every floating-point buffer on the plan's tape (block inputs /
    intermediates / outputs as the block emitters recorded them), in forward order.  Synchronises the device."""
    out = {}
    for i, rec in enumerate(plan.tape):
        for k, v in vars(rec).items():
            if torch.is_tensor(v) and v.is_floating_point() and v.numel() > 0:
                out[f"{i:03d}.{rec.kind}.{getattr(rec, 'name', '')}.{k}"] = float(v.float().abs().max())
    return out


def assert_finite_activations(plans, limit: float = FP16_MAX, what: str = "") -> dict:
    """Raise naming the FIRST buffer (forward order) that holds a NaN / Inf or exceeds ``limit``; returns the merged report plus
    ``"__max__": (name, value)`` of the largest entry."""
    report = {}
    for pi, plan in enumerate(plans if isinstance(plans, (list, tuple)) else [plans]):
        for k, v in activation_absmax(plan).items():
            report[f"p{pi}.{k}"] = v
    for k, v in report.items():
        if not math.isfinite(v) or v > limit:
            top = sorted(((x, n) for n, x in report.items() if math.isfinite(x)), reverse=True)[:5]
            raise AssertionError(f"{what}: activation overflow at {k}: max |x| = {v} (limit {limit}); largest finite before it: {top}")
    name = max(report, key=report.get) if report else None
    report["__max__"] = (name, report.get(name))
    return report
