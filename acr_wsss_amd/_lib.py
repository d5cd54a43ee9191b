"""ctypes binding of libacr_hip.so (C ABI: include/acr_hip.h).

The HIP library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is
missing, or a call returns a negative status, this module raises -- loudly -- instead of degrading.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C acr_wsss_amd/csrc``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACR_LIB_PATH: a lab build of the same library (scripts/lab/_build/*.so, e.g. EXTRA=-DLAB_TL stamps) for A/B runs
LIB_PATH = os.environ.get("ACR_LIB_PATH") or os.path.join(_HERE, "libacr_hip.so")

ACR_F32, ACR_BF16, ACR_BF16_F32MATH, ACR_F32_BF16X3 = 0, 1, 2, 3
# acr_math (include/acr_hip.h): how an fp32 entry point multiplies -- a per-call argument.  "f32": exact-fp32 MFMA;
# "f32_split": six bf16-MFMA terms of a three-way operand split (fp32 tensors, fp32 accumulate, fp32-accurate)
MATH = {"f32": 0, "f32_split": 1}
BF16_F32MATH = False      # True: bf16 tensors take the exact-fp32 MFMA kernels (reference for the bf16-MFMA ones)
GETAM_FUNCS = {"grad": 0, "cam_grad": 1, "grad_s": 2, "cam_grad_s": 3}

c_void_p, c_int32, c_int64, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                  ctypes.c_float, ctypes.c_size_t)


class AttnDesc(ctypes.Structure):
    """struct acr_attn_desc (include/acr_hip.h)."""
    _fields_ = [("B", c_int32), ("H", c_int32), ("T", c_int32), ("head_dim", c_int32),
                ("dtype", c_int32), ("scale", c_float),
                ("qkv_sb", c_int64), ("qkv_st", c_int64), ("qkv_sh", c_int64),
                ("o_sb", c_int64), ("o_st", c_int64), ("o_sh", c_int64)]


_P = ctypes.POINTER(AttnDesc)
# name -> (restype, argtypes); must list every symbol include/acr_hip.h declares (tests check this)
SIGNATURES = {
    "acr_version": (c_int32, []),
    "acr_last_error": (ctypes.c_char_p, []),
    "acr_set_option": (c_int32, [c_int32, c_int32]),
    "acr_get_option": (c_int32, [c_int32]),
    "acr_attn_fwd": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                               c_void_p]),
    "acr_attn_bwd": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                               c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acr_attn_scores_floats": (c_int64, [_P]),
    "acr_attn_fwd_scores": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                      c_void_p]),
    "acr_attn_fwd_scores_oimg": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                           c_void_p, c_void_p]),
    "acr_attn_fwd_oimg_offered": (c_int32, [_P]),
    "acr_attn_bwd_scores": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acr_attn_bwd_ws_floats": (c_int64, [_P]),
    "acr_split3_bf16": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "acr_attn_probs": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acr_attn_dprobs": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acr_consistency_ws_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "acr_consistency_fwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                      c_void_p, c_void_p]),
    "acr_consistency_bwd": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                      c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "acr_linear_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                  c_int32, c_int32, c_int32, c_void_p]),
    "acr_gemm_f32_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_x3_image_floats": (c_size_t, [c_int32, c_int32]),
    "acr_x3_colsum_ws_floats": (c_size_t, [c_int32, c_int32]),
    "acr_x3_image": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "acr_x3_image_t": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_gemm_x3_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_gemm_x3": (c_int32, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int32,
                              c_int32, c_int32, c_void_p, c_void_p]),
    "acr_gemm_f32": (c_int32, [c_int32, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p,
                               c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_wgrad_ws_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "acr_wgrad_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                 c_void_p]),
    "acr_colsum_ws_floats": (c_size_t, [c_int32, c_int32]),
    "acr_colsum_bf16": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "acr_conv1x1_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_conv1x1_wgrad_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "acr_conv1x1_wgrad_bf16": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                         c_void_p]),
    "acr_conv1x1_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_conv1x1_x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_conv1x1_f32": (c_int32, [c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_conv1x1_wgrad_f32_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "acr_conv1x1_wgrad_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "acr_conv3x3_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_conv3x3_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_x3_image_many": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "acr_conv3x3_x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_conv3x3_wgrad_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_conv3x3_wgrad_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                        c_void_p]),
    "acr_space_to_depth2_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_depth_to_space2_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_conv_taps_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_conv_taps_x3": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_conv_taps_wgrad_ws_floats": (c_size_t, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    "acr_conv_taps_wgrad_f32": (c_int32, [c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "acr_maxpool3x3s2_fwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                            c_int32, c_void_p]),
    "acr_maxpool3x3s2_bwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                            c_int32, c_void_p]),
    "acr_maxpool3x3s2_fwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, c_void_p]),
    "acr_maxpool3x3s2_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, c_void_p]),
    "acr_sgd_chunk_elems": (c_int32, []),
    "acr_sgd_step_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_void_p]),
    "acr_sgd_step_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_void_p]),
    "acr_linear_gelu_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                       c_int32, c_int32, c_void_p]),
    "acr_linear_dgelu_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32,
                                        c_int32, c_int32, c_void_p]),
    "acr_wgrad_bias_ws_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "acr_wgrad_bias_bf16": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p]),
    "acr_transpose_many_bf16": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "acr_transpose_many_f32": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p]),
    "acr_layernorm_ws_floats": (c_size_t, [c_int32, c_int32]),
    "acr_layernorm_fwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float,
                                         c_void_p]),
    "acr_layernorm_bwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int32, c_int32, c_void_p]),
    "acr_groupnorm_fwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                         c_int32, c_float, c_int32, c_void_p]),
    "acr_groupnorm_bwd_bf16": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p]),
    "acr_groupnorm_fwd_ws_floats": (c_size_t, [c_int32, c_int32, c_int32]),
    "acr_groupnorm_fwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                        c_float, c_int32, c_void_p, c_void_p]),
    "acr_groupnorm_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_groupnorm_fwd_mask_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                             c_float, c_void_p, c_void_p]),
    "acr_groupnorm_bwd_mask_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "acr_weight_std_bf16": (c_int32, [c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p]),
    "acr_weight_std_f32": (c_int32, [c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p]),
    "acr_layernorm_fwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "acr_layernorm_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int32, c_int32, c_void_p]),
    "acr_layernorm_image_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p]),
    "acr_tokens_fwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_tokens_bwd_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "acr_mlsm_fwd_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_mlsm_bwd_f32": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_subsample2_fwd_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "acr_subsample2_bwd_f32": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    "acr_preprocess_batch": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                                       c_int32, c_void_p, c_void_p]),
    "acr_lattice_ws_bytes": (c_int64, [c_int32, c_int32]),
    "acr_lattice_build": (c_int32, [c_void_p, c_int32, c_int32, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "acr_lattice_info": (c_int32, [c_void_p, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32), c_void_p]),
    "acr_lattice_tables": (c_int32, [c_void_p, c_int32, c_int32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                     ctypes.POINTER(c_void_p)]),
    "acr_lattice_filter": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32,
                                     c_int32, c_void_p, c_void_p]),
    "acr_crf_unary": (c_int32, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    "acr_crf_norm": (c_int32, [c_void_p, c_int32, c_void_p]),
    "acr_crf_update": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "acr_getam_row_accum": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                      c_void_p, c_void_p]),
    "acr_aff_refine": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "acr_aff_refine_batch": (c_int32, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "acr_getam_rows_accum": (c_int32, [_P, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "acr_patch_cam": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                c_void_p, c_void_p]),
    "acr_bilinear_resize": (c_int32, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                      c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
}

# acr_option (include/acr_hip.h): kernel-variant selectors of the library's explicit option table, name -> code.  Set through
# set_option() by tests and lab scripts only; round 6 removed the environment variables that used to feed them at load time
# (every A/B they served is recorded and settled) -- the library itself never reads the environment.
OPTIONS = {"gemm_variant": 0, "gemm_nowide": 1, "gemm_regstage": 2, "wgrad_variant": 3, "wgrad_waves": 4, "dq_variant": 5,
           "gemm_f32_regstage": 6, "attn_delta_1head": 7, "gemm_f32_notail": 9, "attn_f32_nosplittail": 10, "gemm_x3_inkernel": 11, "gn_plan": 12}

_lib = None


def set_option(name, value):
    """Select a kernel variant (A/B measurement switch; acr_set_option in include/acr_hip.h)."""
    check(load().acr_set_option(OPTIONS[name], int(value)), "acr_set_option")


class AcrHipError(RuntimeError):
    pass


def load():
    """Load libacr_hip.so (once).  Raises AcrHipError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AcrHipError(
                "libacr_hip.so not found at %s -- the HIP extension is the product path and there is no "
                "fallback.  Build it: python -c 'import __graft_entry__ as g; g.build()'" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        if lib.acr_version() != 2:
            raise AcrHipError("libacr_hip.so ABI version %d != 2 (rebuild it: __graft_entry__.build())" % lib.acr_version())
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise AcrHipError("%s failed (%d): %s" % (what, rc, load().acr_last_error().decode()))


def stream_ptr():
    """Raw hipStream_t of torch's current stream (kernels are enqueued there; graph-capturable)."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def dtype_code(dt):
    if dt == torch.float32:
        return ACR_F32
    if dt == torch.bfloat16:
        return ACR_BF16_F32MATH if BF16_F32MATH else ACR_BF16
    raise AcrHipError("unsupported dtype %s (fp32 and bf16 are built)" % dt)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AcrHipError("acr_wsss_amd ops run on the GPU only (got a %s tensor); there is no CPU path"
                              % t.device)
