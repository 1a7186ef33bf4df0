"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/phendiff_hip.h declares
(no compute calls here: there is no GPU).  Also pins the ctypes struct layouts against the header's field lists."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phendiff_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import phendiff_amd._lib as L
    assert os.path.exists(L.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = L.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/phendiff_hip.h but not exported"
        assert n in L.SYMBOLS, f"{n} has no ctypes prototype in phendiff_amd/_lib.py"
    assert sorted(L.SYMBOLS) == names
    # one version number in three places: the header's macro, the binding's constant, the library's answer (bumped whenever an
    # args struct grows or an entry point is added: r2 added trailing pointer fields, r3 = 3)
    import re
    macro = int(re.search(r"#define PD_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.pd_abi_version() == L.ABI_VERSION == macro
    assert lib.pd_last_error() is not None


def test_struct_field_order_matches_header():
    import phendiff_amd._lib as L
    src = open(HEADER).read()
    pairs = {"pd_temb_args": L.TembArgs, "pd_conv_in_args": L.ConvInArgs, "pd_gn_stats_args": L.GnStatsArgs,
             "pd_conv_args": L.ConvArgs, "pd_gn_finalize_args": L.GnFinalizeArgs, "pd_attn_args": L.AttnArgs,
             "pd_ddim_step_args": L.DdimStepArgs, "pd_add_noise_args": L.AddNoiseArgs, "pd_postproc_args": L.PostprocArgs,
             "pd_attn_d64_args": L.AttnD64Args, "pd_attn_wide_args": L.AttnWideArgs, "pd_attn_wide_bwd_args": L.AttnWideBwdArgs, "pd_latent_sample_args": L.LatentSampleArgs,
             "pd_attn_d64_bwd_args": L.AttnD64BwdArgs, "pd_layernorm_bwd_args": L.LayerNormBwdArgs,
             "pd_geglu_bwd_args": L.GegluBwdArgs, "pd_linear_args": L.LinearArgs, "pd_gn_apply_args": L.GnApplyArgs, "pd_token_wgrad_args": L.TokenWgradArgs, "pd_layernorm_args": L.LayerNormArgs, "pd_geglu_args": L.GegluArgs,
             "pd_pack_weight_args": L.PackWeightArgs, "pd_pack_weight_batch_args": L.PackWeightBatchArgs, "pd_zero_args": L.ZeroArgs}
    for cname, cls in pairs.items():
        body = re.search(r"typedef struct(?:\s+\w+)?\s*\{([^{}]*)\}\s*" + cname + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*$", part.strip())[0])
        assert fields == [f[0] for f in cls._fields_], cname


def test_argument_validation_without_gpu():
    """Entry points validate before launching: bad arguments return an error code and a message, no device needed."""
    import phendiff_amd._lib as L
    lib = L.lib()
    a = L.ConvArgs(dtype=7)
    assert lib.pd_conv(C.byref(a), None) == -1 and b"dtype" in lib.pd_last_error()
    a = L.ConvArgs(dtype=1, B=1, Hin=8, Win=8, Hout=8, Wout=8, C0=48, Cout=64, Cout_pad=64, ksize=3, stride=1, pad=1)
    assert lib.pd_conv(C.byref(a), None) == -2 and b"multiples of 32" in lib.pd_last_error()
    assert lib.pd_attn_d8(C.byref(L.AttnArgs(dtype=1, B=0)), None) == -2
    assert lib.pd_ddim_step(C.byref(L.DdimStepArgs(numel=0)), None) == -1
    assert lib.pd_conv_stat_tiles(256, 256, 3, 1) == 256 and lib.pd_conv_stat_tiles(64, 64, 3, 2) == 32
    assert lib.pd_attn_wide(C.byref(L.AttnWideArgs(dtype=1, B=1, heads=1, D=512, Nq=0, Nkv=4)), None) == -2
    assert lib.pd_attn_d64_bwd(C.byref(L.AttnD64BwdArgs(dtype=1, B=1, heads=1, Nq=4, Nkv=4)), None) == -1
    assert lib.pd_attn_wide_bwd(C.byref(L.AttnWideBwdArgs(dtype=1, B=1, heads=1, D=512, Nq=4, Nkv=4)), None) == -1
    assert lib.pd_allreduce_bucket(None, None, 16, 1, 1, None) == -1 and b"communicator" in lib.pd_last_error()
    assert lib.pd_comm_init(None, 0, 1, None) == -1 and lib.pd_comm_destroy(None) == 0
    assert lib.pd_layernorm_bwd(C.byref(L.LayerNormBwdArgs(dtype=1, rows=4, C=12)), None) == -1
    assert lib.pd_layernorm_bwd_blocks(10) == 3 and lib.pd_layernorm_bwd_blocks(1 << 20) == 2048
    with pytest.raises(L.PhenDiffHipError):
        L.check(-1, "x")
