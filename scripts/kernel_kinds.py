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
