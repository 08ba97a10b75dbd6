"""ctypes binding of libsbl_hip.so (include/sbl_hip.h).  The product path has NO
CPU fallback: if the library is missing or a call fails, it raises."""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbl_hip.so")

P, I, L, F, U64 = c_void_p, c_int, c_long, c_float, c_uint64

# name -> argtypes (all return int); mirrors include/sbl_hip.h one to one
SIGNATURES = {
    "sbl_gemm_f32": [I, I, I, I, I, P, L, P, L, P, L, P, I, P, L, I, P, P, L, P],
    "sbl_wgrad_seg_f32": [I, P, L, P, L, P, I, I, P, L, P, P],
    "sbl_wgrad_group_f32": [I, I, P, P, P, P, P, P, P, P, P, P, P, L, P],
    "sbl_colsum_f32": [P, L, P, I, I, I, P],
    "sbl_stem_conv_fwd": [P, P, P, P, I, I, I, I, P],
    "sbl_stem_bn_relu_pool_fwd": [P, P, P, P, P, P, P, I, I, I, P],
    "sbl_stem_bwd_reduce": [P, P, P, P, P, P, P, P, I, I, I, P],
    "sbl_stem_wgrad": [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P],
    "sbl_bn_finalize": [P, L, P, P, F, F, P, P, I, P, P],
    "sbl_bn_eval_stats": [P, P, F, P, P, I, P],
    "sbl_bn_apply_fwd": [P, P, P, P, P, P, P, L, I, I, P],
    "sbl_bn_apply_fwd_stats": [P, P, P, L, P, P, F, F, P, P, P, P, P, P, L, I, I, P],
    "sbl_bn_bwd_reduce": [P, P, P, P, P, P, L, I, I, P, L, P],
    "sbl_bn_bwd_apply": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, P],
    "sbl_conv_weight_pack": [P, P, P, I, I, I, I, P, I, P],
    "sbl_conv_wgrad_unpack": [P, P, I, I, I, I, I, P],
    "sbl_conv2d_fwd": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "sbl_conv2d_dgrad": [P, P, P, I, I, I, I, I, I, I, I, I, P, L, P],
    "sbl_conv2d_dgrad_bnstats": [P, P, P, I, I, I, I, I, I, I, I, I, P, L, P, P, P, P, P, I, P],
    "sbl_conv2d_dgrad_fused": [P, P, P, I, I, I, I, I, I, I, I, I, P, L, P, P, P, P, P, P, P, P, P, I, P],
    "sbl_conv1x1s2_dgrad_compact": [P, P, P, I, I, I, I, I, P, L, P],
    "sbl_conv2d_wgrad": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "sbl_avgpool_fwd": [P, P, I, I, I, P],
    "sbl_avgpool_bwd": [P, P, I, I, I, P],
    "sbl_dropout": [P, P, L, F, P, U64, P],
    "sbl_seed_bump": [P, P],
    "sbl_add_layernorm_fwd": [P, P, P, P, P, P, P, I, I, F, F, P, U64, P],
    "sbl_add_layernorm_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, F, P, U64, P],
    "sbl_add_pe": [P, P, P, I, I, I, P],
    "sbl_rowscale": [P, P, P, L, I, P],
    "sbl_attention_fwd": [P, L, P, L, P, L, P, L, P, I, P, I, I, I, I, F, F, P, U64, P],
    "sbl_attention_bwd": [P, L, P, L, P, L, P, L, P, P, L, P, L, P, L, I, I, I, I, F, F, P, U64, P],
    "sbl_attention_seg_fwd": [P, L, P, L, P, L, P, L, P, I, P, I, I, P, I, I, F, F, P, U64, P],
    "sbl_attention_seg2_fwd": [P, P, L, P, P, L, P, P, L, P, P, L, P, P, I, I, I, P, I, I, F, F, P, U64, U64, P],
    "sbl_gemm2_f32": [I, I, I, P, P, L, P, P, L, P, P, L, P, P, I, P, L, P],
    "sbl_add_layernorm2_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, F, F, P, U64, U64, P],
    "sbl_add_layernorm2_fusion_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, I, I, F, F, P, U64, U64, P],
    "sbl_attention_seg_bwd": [P, L, P, L, P, L, P, L, P, P, L, P, L, P, L, I, I, P, I, I, F, F, P, U64, P],
    "sbl_embed_pe_seg_fwd": [P, L, P, P, P, I, P, I, I, I, P],
    "sbl_embed_pe_drop2_fwd": [P, P, L, P, P, P, P, I, P, I, I, I, F, P, U64, U64, P],
    "sbl_decoder_tail_fwd": [P, P, P, P, P, P, P, P, L, P, P, L, I, I, I, P, I, I, I, P],
    "sbl_embed_seg_bwd": [P, L, P, P, I, P, I, I, I, P],
    "sbl_fusion_seg_fwd": [P, P, P, P, I, P, I, I, P],
    "sbl_fusion_seg_bwd": [P, P, P, P, I, P, I, I, P],
    "sbl_gather_last_fwd": [P, P, I, P, I, I, P],
    "sbl_gather_last_bwd": [P, P, I, P, I, I, P],
    "sbl_embed_pe_fwd": [P, L, P, P, P, I, I, I, I, P],
    "sbl_embed_bwd": [P, L, P, P, I, I, I, I, P],
    "sbl_fusion_fwd": [P, P, P, P, I, I, I, P],
    "sbl_fusion_bwd": [P, P, P, P, I, I, I, P],
    "sbl_argmax_select": [P, L, P, L, P, L, I, I, P, I, I, P],
    "sbl_decoder_preprocess": [P, P, P, P, P, P, I, I, I, L, L, L, P],
    "sbl_smoothed_ce_fwd": [P, P, P, I, I, F, I, P],
    "sbl_smoothed_ce_bwd": [P, P, P, P, P, I, I, F, I, P],
    "sbl_adam_step": [P, P, P, P, L, F, F, F, F, I, F, P],
    "sbl_preprocess_clips": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "sbl_set_matmul_precision": [I],
    "sbl_set_tuning": [I, I],
}

_lib = None


class SblHipError(RuntimeError):
    pass


def load():
    """Load libsbl_hip.so (once).  Raises SblHipError with a build hint if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SblHipError(
            "libsbl_hip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sbl_for_multilingual_lip_reading_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.sbl_last_error.restype = ctypes.c_char_p
    lib.sbl_last_error.argtypes = []
    lib.sbl_abi_version.restype = c_int
    lib.sbl_abi_version.argtypes = []
    for name in ("sbl_profile_end", "sbl_profile_last_slot", "sbl_profile_last_kernel", "sbl_profile_used", "sbl_get_matmul_precision"):
        getattr(lib, name).restype = c_int
        getattr(lib, name).argtypes = []
    lib.sbl_profile_begin.restype = c_int
    lib.sbl_profile_begin.argtypes = [c_void_p, c_int]
    lib.sbl_wgrad_group_table_bytes.restype = ctypes.c_long
    lib.sbl_wgrad_group_table_bytes.argtypes = [c_int]
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = c_int
        fn.argtypes = argtypes
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an entry point; non-zero status -> SblHipError(sbl_last_error())."""
    lib = _lib if _lib is not None else load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SblHipError("%s failed (%d): %s" % (name, rc, lib.sbl_last_error().decode("utf-8", "replace")))
