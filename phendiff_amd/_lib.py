"""ctypes binding of ``libphendiff_hip.so`` (the C ABI declared in ``include/phendiff_hip.h``).

The product path has no CPU fallback: if the shared library is missing or fails to load, importing
:func:`lib` raises.  Build it with ``phendiff_amd/csrc/build.sh`` (``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PD_LIB: diagnostic override (same-box A/B of two builds of the library, kept OUTSIDE the package under build_ab/); the default
# -- and the only thing the driver ever loads -- is the in-tree build
LIB_PATH = os.environ.get("PD_LIB") or os.path.join(_HERE, "libphendiff_hip.so")



def source_hash() -> str:
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/phendiff_hip.h, the build script): identifies the code
    a build / a committed profile belongs to.  The definition lives in csrc/source_hash.py (standalone: build.sh runs it too)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_pd_source_hash", os.path.join(_HERE, "csrc", "source_hash.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def library_is_current() -> bool:
    """True when the loaded library's manifest names the sources in the tree (an in-tree build of this checkout)."""
    import json
    try:
        with open(LIB_PATH + ".manifest.json") as f:
            return json.load(f).get("sources_sha256") == source_hash()
    except (OSError, ValueError):
        return False


PD_F32, PD_BF16, PD_F16 = 0, 1, 2
PD_PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}
PD_OUT_NHWC, PD_OUT_NCHW_F32, PD_OUT_QKV_HEADS = 0, 1, 2
ABI_VERSION = 8

vp = C.c_void_p


class TembArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("c0", C.c_int), ("tdim", C.c_int), ("proj_dim", C.c_int),
                ("flip_sin_to_cos", C.c_int), ("freq_shift", C.c_float), ("num_classes", C.c_int),
                ("timesteps", vp), ("labels", vp), ("class_emb", vp),
                ("w1", vp), ("b1", vp), ("w2", vp), ("b2", vp), ("class_table", vp), ("wp", vp), ("bp", vp),
                ("emb", vp), ("proj", vp), ("feat", vp), ("z1", vp)]


class ConvInArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("x", vp), ("w", vp), ("bias", vp), ("y", vp)]


class GnStatsArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("HW", C.c_int), ("C0", C.c_int), ("C1", C.c_int),
                ("groups", C.c_int), ("eps", C.c_float), ("x0", vp), ("x1", vp), ("gamma", vp), ("beta", vp),
                ("partial", vp), ("splits", C.c_int), ("scale", vp), ("shift", vp)]


class ConvArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
                ("C0", C.c_int), ("C1", C.c_int), ("Cout", C.c_int), ("Cout_pad", C.c_int), ("ksize", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int), ("upsample", C.c_int), ("silu", C.c_int), ("out_mode", C.c_int),
                ("heads", C.c_int), ("x0", vp), ("x1", vp), ("scale", vp), ("shift", vp), ("w_packed", vp), ("bias", vp),
                ("temb", vp), ("temb_stride", C.c_int), ("residual", vp), ("y", vp), ("stats_out", vp), ("tail_x0", vp), ("tail_x1", vp), ("tail_C0", C.c_int),
                ("tail_C1", C.c_int), ("im2col3", C.c_int), ("phase", C.c_int), ("phase_in", C.c_int)]


class GnFinalizeArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("HW", C.c_int), ("groups", C.c_int), ("eps", C.c_float),
                ("C0", C.c_int), ("T0", C.c_int), ("stats0", vp), ("C1", C.c_int), ("T1", C.c_int), ("stats1", vp),
                ("gamma", vp), ("beta", vp), ("scale", vp), ("shift", vp), ("mean", vp), ("rstd", vp),
                ("temb", vp), ("temb_stride", C.c_int)]


class GnBwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("HW", C.c_int), ("C0", C.c_int), ("C1", C.c_int), ("groups", C.c_int),
                ("silu", C.c_int), ("x0", vp), ("x1", vp), ("dz0", vp), ("dz1", vp), ("mean", vp), ("rstd", vp), ("gamma", vp),
                ("beta", vp), ("partial", vp), ("splits", C.c_int), ("coef", vp), ("dx0", vp), ("dx1", vp),
                ("accumulate0", C.c_int), ("accumulate1", C.c_int), ("dgamma", vp), ("dbeta", vp),
                ("dz_combined", C.c_int), ("res", vp), ("sum0", vp), ("sum1", vp),
                ("mod", vp), ("mod_stride", C.c_int), ("dmod", vp)]


class Pool2x2Args(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("du", vp), ("dx", vp),
                ("accumulate", C.c_int)]


class ChannelSumArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("HW", C.c_int), ("C", C.c_int), ("x", vp), ("out", vp),
                ("out_stride", C.c_int), ("accumulate", C.c_int), ("total", vp), ("total_valid", C.c_int),
                ("workspace", vp), ("splits", C.c_int)]


class NchwToNhwcArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("Cpad", C.c_int), ("x", vp), ("out", vp)]


class LinearWgradArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("in_dim", C.c_int), ("out_dim", C.c_int), ("x_silu", C.c_int), ("dy", vp), ("x", vp),
                ("dw", vp), ("db", vp)]


class LinearDgradArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("in_dim", C.c_int), ("out_dim", C.c_int), ("dy", vp), ("w", vp), ("pre", vp), ("dx", vp)]


class EmbeddingGradArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("dim", C.c_int), ("num_classes", C.c_int), ("labels", vp), ("d", vp), ("dtable", vp)]


class WgradArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Hout", C.c_int), ("Wout", C.c_int),
                ("C0", C.c_int), ("C1", C.c_int), ("Cout", C.c_int), ("ksize", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("upsample", C.c_int), ("silu", C.c_int), ("x0", vp), ("x1", vp), ("scale", vp), ("shift", vp), ("dy", vp),
                ("slab", vp), ("slab_bytes", C.c_size_t), ("dw", vp), ("Cout_valid", C.c_int), ("Cin_valid", C.c_int),
                ("accumulate", C.c_int), ("phase", C.c_int), ("stage", C.c_int)]


class PackWeightArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("cout", C.c_int), ("cin", C.c_int), ("cout_pad", C.c_int), ("cin_pad", C.c_int),
                ("ksize", C.c_int), ("src_in", C.c_int), ("dgrad", C.c_int), ("src", vp), ("dst", vp),
                ("dst_ct_stride", C.c_longlong), ("dst2", vp), ("dst2_ct_stride", C.c_longlong)]


class PackWeightBatchArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("n", C.c_int), ("jobs", vp), ("starts", vp), ("total_blocks", C.c_int), ("max_ksize", C.c_int)]


class Im2col3Args(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("x", vp), ("out", vp)]


class AttnArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("N", C.c_int),
                ("q", vp), ("k", vp), ("v", vp), ("out", vp), ("lse", vp), ("kmax2", vp)]


class AttnD64Args(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("Nq", C.c_int), ("Nkv", C.c_int), ("q", vp),
                ("q_stride", C.c_int), ("k", vp), ("v", vp), ("kv_stride", C.c_int), ("out", vp), ("out_stride", C.c_int), ("lse", vp)]


class TokenWgradArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_longlong), ("K", C.c_int), ("N", C.c_int), ("x", vp), ("x_stride", C.c_int), ("dy", vp),
                ("dy_stride", C.c_int), ("dw", vp), ("accumulate", C.c_int), ("slab", vp), ("slab_bytes", C.c_size_t), ("stage", C.c_int)]


class GnApplyArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("HW", C.c_int), ("C0", C.c_int), ("C1", C.c_int), ("silu", C.c_int),
                ("x0", vp), ("x1", vp), ("scale", vp), ("shift", vp), ("y", vp)]


class LinearArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_longlong), ("K", C.c_int), ("N", C.c_int), ("N_pad", C.c_int), ("x", vp),
                ("x_stride", C.c_int), ("w_packed", vp), ("bias", vp), ("residual", vp), ("y", vp), ("scale", vp), ("shift", vp),
                ("rows_per_sample", C.c_int), ("qkv_heads", C.c_int), ("stats_out", vp), ("glu", C.c_int), ("kmax2_out", vp),
                ("fold_ws", vp), ("fold_ws_bytes", C.c_size_t)]


class AttnWideArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("D", C.c_int), ("Nq", C.c_int), ("Nkv", C.c_int),
                ("scale", C.c_float), ("q", vp), ("q_stride", C.c_int), ("k", vp), ("v", vp), ("kv_stride", C.c_int), ("out", vp),
                ("out_stride", C.c_int), ("lse", vp)]


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class AttnWideBwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("D", C.c_int), ("Nq", C.c_int), ("Nkv", C.c_int),
                ("scale", C.c_float), ("q", vp), ("q_stride", C.c_int), ("k", vp), ("v", vp), ("kv_stride", C.c_int), ("o", vp),
                ("dout", vp), ("o_stride", C.c_int), ("lse", vp), ("delta", vp), ("dq", vp), ("dq_stride", C.c_int), ("dk", vp),
                ("dv", vp), ("dkv_stride", C.c_int)]


class AttnD64BwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("Nq", C.c_int), ("Nkv", C.c_int), ("q", vp),
                ("q_stride", C.c_int), ("k", vp), ("v", vp), ("kv_stride", C.c_int), ("o", vp), ("dout", vp), ("o_stride", C.c_int),
                ("lse", vp), ("delta", vp), ("dq", vp), ("dq_stride", C.c_int), ("dk", vp), ("dv", vp), ("dkv_stride", C.c_int)]


class LayerNormBwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_longlong), ("C", C.c_int), ("eps", C.c_float), ("x", vp), ("dy", vp), ("gamma", vp),
                ("res", vp), ("dx", vp), ("dgamma", vp), ("dbeta", vp), ("partial", vp), ("dxsum", vp)]


class TokenEmbeddingGradArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_int), ("dim", C.c_int), ("num_classes", C.c_int), ("row_stride", C.c_longlong),
                ("labels", vp), ("d", vp), ("dtable", vp)]


class GegluBwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_longlong), ("inner", C.c_int), ("x", vp), ("dy", vp), ("dx", vp),
                ("sums", vp), ("sum_splits", C.c_int), ("B", C.c_int)]


class UpsamplePhaseWeightsArgs(C.Structure):
    _fields_ = [("cout", C.c_int), ("cin", C.c_int), ("w", vp), ("out", vp)]


class LatentSampleArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("HW", C.c_int), ("scale", C.c_float), ("moments", vp), ("noise", vp), ("out", vp)]


class LayerNormArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_longlong), ("C", C.c_int), ("eps", C.c_float), ("x", vp), ("gamma", vp),
                ("beta", vp), ("y", vp)]


class GegluArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("rows", C.c_longlong), ("inner", C.c_int), ("x", vp), ("y", vp)]


class LpGuidanceArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("per_sample", C.c_int64), ("pred_type", C.c_int), ("clip", C.c_int),
                ("clip_range", C.c_float), ("sqrt_a", C.c_float), ("sqrt_b", C.c_float), ("p", C.c_float), ("sample", vp),
                ("model_out", vp), ("target", vp), ("partial", vp), ("splits", C.c_int), ("d_model_out", vp),
                ("d_sample_direct", vp), ("losses", vp)]


class GuidanceApplyArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("scale", C.c_float), ("x", vp), ("g_direct", vp), ("g_unet", vp), ("out", vp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("heads", C.c_int), ("N", C.c_int), ("q", vp), ("k", vp), ("v", vp),
                ("o", vp), ("dout", vp), ("lse", vp), ("delta", vp), ("dqkv", vp), ("slab", vp), ("slab_bytes", C.c_size_t)]


class DdimStepArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("per_sample", C.c_int64), ("pred_type", C.c_int), ("clip", C.c_int),
                ("clip_range", C.c_float), ("use_clipped_model_output", C.c_int),
                ("sqrt_a", C.c_float), ("sqrt_b", C.c_float), ("sqrt_ap", C.c_float), ("dir_coef", C.c_float),
                ("sample", vp), ("model_out", vp), ("uncond_out", vp), ("w", vp), ("w_per_sample", C.c_int),
                ("guidance_cfg", C.c_int), ("prev_sample", vp), ("pred_x0", vp)]


class AddNoiseArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("per_sample", C.c_int64), ("velocity", C.c_int),
                ("x", vp), ("noise", vp), ("sa", vp), ("sb", vp), ("out", vp)]


class LossArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("per_sample", C.c_int64), ("pred_type", C.c_int), ("model_out", vp), ("noise", vp),
                ("clean", vp), ("weight", vp), ("sa", vp), ("sb", vp), ("grad_scale", C.c_float), ("grad_out", vp),
                ("partial", vp), ("loss_out", vp)]


class AdamWEmaArgs(C.Structure):
    _fields_ = [("numel", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("step_size", C.c_float), ("bias_correction2_sqrt", C.c_float),
                ("one_minus_decay", C.c_float), ("zero_grad", C.c_int), ("clip_coef", vp), ("param", vp), ("grad", vp),
                ("exp_avg", vp), ("exp_avg_sq", vp), ("ema", vp), ("ema_only", C.c_int)]


class PostprocArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("C", C.c_int), ("H", C.c_int), ("W", C.c_int), ("x", vp), ("out_f32", vp), ("out_u8", vp)]


class ZeroArgs(C.Structure):
    _fields_ = [("ptr", vp), ("bytes", C.c_size_t)]


class ResizeTf1Args(C.Structure):
    _fields_ = [("dtype", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("OH", C.c_int), ("OW", C.c_int),
                ("scale_y", C.c_float), ("scale_x", C.c_float), ("sub", C.c_float), ("div", C.c_float), ("x", vp), ("y", vp)]


class ConvRectArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("Cin", C.c_int), ("Hout", C.c_int),
                ("Wout", C.c_int), ("Cout_pad", C.c_int), ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad_h", C.c_int),
                ("pad_w", C.c_int), ("relu", C.c_int), ("x", vp), ("x_cs", C.c_int), ("w_packed", vp), ("bias", vp), ("y", vp),
                ("y_cs", C.c_int), ("y_co", C.c_int)]


class Pool2dArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int), ("C", C.c_int), ("Hout", C.c_int),
                ("Wout", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("mode", C.c_int), ("x", vp),
                ("x_cs", C.c_int), ("y", vp), ("y_cs", C.c_int), ("y_co", C.c_int)]


class FcF32Args(C.Structure):
    _fields_ = [("rows", C.c_int), ("in_dim", C.c_int), ("out_dim", C.c_int), ("x", vp), ("wt", vp), ("bias", vp), ("y", vp)]


# every symbol include/phendiff_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pd_abi_version": (C.c_int, []),
    "pd_last_error": (C.c_char_p, []),
    "pd_temb": (C.c_int, [C.POINTER(TembArgs), vp]),
    "pd_conv_in": (C.c_int, [C.POINTER(ConvInArgs), vp]),
    "pd_gn_stats": (C.c_int, [C.POINTER(GnStatsArgs), vp]),
    "pd_conv": (C.c_int, [C.POINTER(ConvArgs), vp]),
    "pd_conv_stat_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pd_gn_finalize": (C.c_int, [C.POINTER(GnFinalizeArgs), vp]),
    "pd_attn_d8": (C.c_int, [C.POINTER(AttnArgs), vp]),
    "pd_ddim_step": (C.c_int, [C.POINTER(DdimStepArgs), vp]),
    "pd_add_noise": (C.c_int, [C.POINTER(AddNoiseArgs), vp]),
    "pd_zero": (C.c_int, [C.POINTER(ZeroArgs), vp]),
    "pd_postproc": (C.c_int, [C.POINTER(PostprocArgs), vp]),
    "pd_gn_silu_bwd": (C.c_int, [C.POINTER(GnBwdArgs), vp]),
    "pd_pool2x2_sum": (C.c_int, [C.POINTER(Pool2x2Args), vp]),
    "pd_channel_sum": (C.c_int, [C.POINTER(ChannelSumArgs), vp]),
    "pd_nchw_to_nhwc": (C.c_int, [C.POINTER(NchwToNhwcArgs), vp]),
    "pd_linear_wgrad": (C.c_int, [C.POINTER(LinearWgradArgs), vp]),
    "pd_linear_dgrad": (C.c_int, [C.POINTER(LinearDgradArgs), vp]),
    "pd_embedding_grad": (C.c_int, [C.POINTER(EmbeddingGradArgs), vp]),
    "pd_conv_wgrad_workspace": (C.c_size_t, [C.POINTER(WgradArgs)]),
    "pd_conv_wgrad": (C.c_int, [C.POINTER(WgradArgs), vp]),
    "pd_im2col3": (C.c_int, [C.POINTER(Im2col3Args), vp]),
    "pd_pack_weight": (C.c_int, [C.POINTER(PackWeightArgs), vp]),
    "pd_upsample_phase_weights": (C.c_int, [C.POINTER(UpsamplePhaseWeightsArgs), vp]),
    "pd_pack_weight_batch": (C.c_int, [C.POINTER(PackWeightBatchArgs), vp]),
    "pd_attn_d8_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), vp]),
    "pd_attn_d8_bwd_workspace": (C.c_size_t, [C.POINTER(AttnBwdArgs)]),
    "pd_attn_d64": (C.c_int, [C.POINTER(AttnD64Args), vp]),
    "pd_token_wgrad": (C.c_int, [C.POINTER(TokenWgradArgs), vp]),
    "pd_token_wgrad_workspace": (C.c_size_t, [C.POINTER(TokenWgradArgs)]),
    "pd_gn_apply": (C.c_int, [C.POINTER(GnApplyArgs), vp]),
    "pd_linear": (C.c_int, [C.POINTER(LinearArgs), vp]),
    "pd_linear_fold_workspace": (C.c_size_t, [C.POINTER(LinearArgs)]),
    "pd_attn_wide": (C.c_int, [C.POINTER(AttnWideArgs), vp]),
    "pd_attn_wide_bwd": (C.c_int, [C.POINTER(AttnWideBwdArgs), vp]),
    "pd_comm_unique_id": (C.c_int, [C.POINTER(CommId)]),
    "pd_comm_init": (C.c_int, [C.POINTER(CommId), C.c_int, C.c_int, C.POINTER(vp)]),
    "pd_allreduce_bucket": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int, vp]),
    "pd_comm_query": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pd_comm_destroy": (C.c_int, [vp]),
    "pd_latent_sample": (C.c_int, [C.POINTER(LatentSampleArgs), vp]),
    "pd_attn_d64_bwd": (C.c_int, [C.POINTER(AttnD64BwdArgs), vp]),
    "pd_layernorm_bwd": (C.c_int, [C.POINTER(LayerNormBwdArgs), vp]),
    "pd_layernorm_bwd_blocks": (C.c_int, [C.c_longlong]),
    "pd_geglu_bwd": (C.c_int, [C.POINTER(GegluBwdArgs), vp]),
    "pd_token_embedding_grad": (C.c_int, [C.POINTER(TokenEmbeddingGradArgs), vp]),
    "pd_layernorm": (C.c_int, [C.POINTER(LayerNormArgs), vp]),
    "pd_geglu": (C.c_int, [C.POINTER(GegluArgs), vp]),
    "pd_lp_guidance": (C.c_int, [C.POINTER(LpGuidanceArgs), vp]),
    "pd_guidance_apply": (C.c_int, [C.POINTER(GuidanceApplyArgs), vp]),
    "pd_diffusion_loss": (C.c_int, [C.POINTER(LossArgs), vp]),
    "pd_grad_norm": (C.c_int, [vp, C.c_int64, vp, C.c_float, vp, vp, vp]),
    "pd_adamw_ema": (C.c_int, [C.POINTER(AdamWEmaArgs), vp]),
    "pd_resize_tf1": (C.c_int, [C.POINTER(ResizeTf1Args), vp]),
    "pd_conv_rect": (C.c_int, [C.POINTER(ConvRectArgs), vp]),
    "pd_pool2d": (C.c_int, [C.POINTER(Pool2dArgs), vp]),
    "pd_fc_f32": (C.c_int, [C.POINTER(FcF32Args), vp]),
    "pd_graph_begin": (C.c_int, [vp]),
    "pd_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "pd_graph_launch": (C.c_int, [vp, vp]),
    "pd_graph_destroy": (C.c_int, [vp]),
    "pd_event_create": (C.c_int, [C.POINTER(vp)]),
    "pd_event_record": (C.c_int, [vp, vp]),
    "pd_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "pd_event_destroy": (C.c_int, [vp]),
}

_lib = None


class PhenDiffHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the HIP library; raises if it is absent -- there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PhenDiffHipError(
            f"{LIB_PATH} not found: build it with phendiff_amd/csrc/build.sh (hipcc --offload-arch=gfx950). "
            "phendiff_amd has no CPU fallback.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise PhenDiffHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if l.pd_abi_version() != ABI_VERSION:
        # same-box A/B against a library built from an older revision (scripts/build_rev.sh, scripts/ab_forward.py): the caller
        # vouches that the structs it exercises did not change between the two
        if not (os.environ.get("PD_ALLOW_ABI_MISMATCH") and LIB_PATH != os.path.join(_HERE, "libphendiff_hip.so")):
            raise PhenDiffHipError(f"ABI mismatch: library {l.pd_abi_version()} != binding {ABI_VERSION}")
    _lib = l
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().pd_last_error()
        raise PhenDiffHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
