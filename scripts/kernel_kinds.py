"""Kernel name (rocprofv3) -> launch-plan kind (bench.py's `roofline.kernel`).  ONE place: the collectors below write
`by_kind` with it, bench.py only reads `by_kind` (and refuses a file measured on other kernel sources)."""
import re

RULES = [
    (r"attn_kernel<|attn_glds_kernel<", "attn_d8"),                # register-staged and DMA-staged forms of pd_attn_d8
    (r"conv_kernel<[^,]+, (3|2), ", "conv3x3"),              # every 3x3 instantiation (stride 1 / 2, with / without the fused shortcut tail) and, round 4, the
                                                             # 2x2 sub-pixel phases that replace the upsamplers' 3x3 (the one "sum of 3x3" definition)
    (r"conv_kernel<[^,]+, 1, |linear_kernel<|linear_dma_kernel<", "conv1x1"),     # 1x1 convs (incl. conv_in's im2col form) and pd_linear: the plan's "conv1x1"
    (r"linear_fold_gn_kernel<", "conv1x1_fold"),             # the per-sample weight fold in front of the pixel q/k/v projection (same plan op as its GEMM)
    (r"gn_finalize", "gn_finalize"),
    (r"temb_kernel", "temb"), (r"ddim_step_kernel", "ddim_step"), (r"postproc_kernel", "postproc"), (r"add_noise_kernel", "add_noise"),
]


def kind_of(kernel_name: str):
    for pat, kind in RULES:
        if re.search(pat, kernel_name):
            return kind
    return None


def sources_sha256():
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from phendiff_amd._lib import source_hash
    return source_hash()


# ---- round 6: the side workloads (bench.py --workload train | sd_train | sd_img2img).  A plan KIND of those rooflines is one launch-plan op,
# which may be several kernels (the GroupNorm backward is three, the attention backward two, a weight gradient + its ordered reduce), and one
# kernel template can serve two kinds (conv_kernel: forward convolutions AND input gradients).  So the collectors write the HBM bytes of ALL
# launches of a kernel GROUP over exactly one profiled step, and bench.py divides by the number of plan ops the group stands for.
GROUP_RULES = [
    (r"attn_d64_dq_kernel|attn_d64_dkv_kernel", "attn_d64_bwd"),
    (r"attn_d64_kernel<", "attn_d64"),
    (r"attn_bwd_dq_kernel|attn_bwd_dkv_kernel|attn_bwd_fused_kernel|attn_dq_reduce_kernel", "attn_d8_bwd"),
    (r"attn_kernel<|attn_glds_kernel<", "attn_d8"),
    (r"token_wgrad", "wgrad_linear"),
    (r"wgrad_kernel<[^,]+, 3, |wgrad_reduce_kernel<9>", "wgrad3x3"),
    (r"wgrad_kernel<[^,]+, 2, |wgrad_reduce_kernel<4>", "wgrad2x2"),
    (r"wgrad_kernel<[^,]+, 1, |wgrad_reduce_kernel<1>", "wgrad1x1"),
    (r"gn_bwd_reduce_kernel|gn_bwd_finalize_kernel|gn_bwd_apply_kernel", "gn_silu_bwd"),
    (r"layernorm_bwd", "layernorm_bwd"),
    (r"layernorm_kernel", "layernorm"),
    (r"geglu_bwd_kernel", "geglu_bwd"), (r"geglu_kernel", "geglu"),
    (r"gn_apply_kernel", "gn_apply"),
    (r"conv_kernel<[^,]+, (3|2), ", "conv3x3"),            # forward 3x3 / sub-pixel 2x2 convolutions AND their input gradients
    (r"conv_kernel<[^,]+, 1, ", "conv1x1"),
    (r"linear_p8_kernel<|linear_dma_kernel<|linear_kernel<", "linear"),   # forward Linear layers AND their input gradients
    (r"gn_finalize", "gn_finalize"),
    (r"channel_sum", "channel_sum"),
    (r"adamw_ema_kernel|sumsq_kernel|grad_norm", "optimizer"),
    (r"pack_weight", "pack_weight"),
]
# plan kinds (bench.py per_kernel_ms keys, "fwd." / "bwd." prefixes stripped where the plan has them) each group covers
GROUP_KINDS = {
    "attn_d64_bwd": ["bwd.attn_d64_bwd"], "attn_d64": ["attn_d64", "fwd.attn_d64"], "attn_d8_bwd": ["bwd.attn_d8_bwd"], "attn_d8": ["fwd.attn_d8", "attn_d8"],
    "wgrad_linear": ["bwd.wgrad_linear"], "wgrad3x3": ["bwd.wgrad3x3"], "wgrad2x2": ["bwd.wgrad2x2"], "wgrad1x1": ["bwd.wgrad1x1"],
    "gn_silu_bwd": ["bwd.gn_silu_bwd"], "layernorm_bwd": ["bwd.layernorm_bwd"], "layernorm": ["layernorm", "fwd.layernorm"],
    "geglu_bwd": ["bwd.geglu_bwd"], "geglu": ["fwd.geglu", "geglu"], "gn_apply": ["gn_apply", "fwd.gn_apply", "bwd.gn_apply_bwd"],
    "conv3x3": ["conv3x3", "fwd.conv3x3", "bwd.dgrad3x3"], "conv1x1": ["conv1x1", "fwd.conv1x1"],
    "linear": ["linear", "fwd.linear", "bwd.dgrad_linear"], "gn_finalize": ["gn_finalize", "fwd.gn_finalize"],
}


def group_of(kernel_name: str):
    for pat, grp in GROUP_RULES:
        if re.search(pat, kernel_name):
            return grp
    return None
