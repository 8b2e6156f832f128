"""ctypes binding of librald_hip.so (include/rald_hip.h).

The library is the product: if it is missing or fails to load, every entry point raises -
there is no eager/PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RALD_LIB_OVERRIDE") or os.path.join(_HERE, "librald_hip.so")
_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_i64, c_float, c_char_p = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_char_p
c_float_p = C.POINTER(C.c_float)


class DitConfig(C.Structure):
    _fields_ = [("n_latents", c_int), ("channels", c_int), ("depth", c_int), ("n_heads", c_int),
                ("d_head", c_int), ("t_channels", c_int), ("context_dim", c_int),
                ("n_cond_tokens", c_int), ("with_radar_enc", c_int), ("enc_hidden_ch", c_int),
                ("enc_radar_ch", c_int), ("radar_r", c_int), ("radar_a", c_int), ("radar_e", c_int),
                ("sigma_data", c_float), ("qkv_dtype", c_int)]


class AeConfig(C.Structure):
    _fields_ = [("dim", c_int), ("num_latents", c_int), ("latent_dim", c_int), ("depth", c_int),
                ("heads", c_int), ("dim_head", c_int), ("num_inputs", c_int), ("query_type", c_int)]


# name -> (restype, argtypes); everything include/rald_hip.h declares
SIGNATURES = {
    "rald_last_error": (c_char_p, []),
    "rald_version": (c_int, []),
    "rald_build_flags": (c_int, []),
    "rald_dit_default_config": (None, [C.POINTER(DitConfig)]),
    "rald_dit_create": (c_int, [C.POINTER(DitConfig), C.POINTER(c_void_p)]),
    "rald_dit_destroy": (None, [c_void_p]),
    "rald_dit_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "rald_dit_finalize": (c_int, [c_void_p]),
    "rald_debug_f16_saturation_count": (c_i64, [c_int]),
    "rald_debug_poison_lds": (c_int, [c_void_p]),
    "rald_dit_reserve": (c_int, [c_void_p, c_int]),
    "rald_dit_workspace_generation": (c_i64, [c_void_p]),
    "rald_dit_set_two_stream_min_batch": (c_int, [c_void_p, c_int]),
    "rald_dit_two_stream_min_batch": (c_int, [c_void_p]),
    "rald_ae_workspace_generation": (c_i64, [c_void_p]),
    "rald_dit_set_sigmas": (c_int, [c_void_p, c_float_p, c_int, c_void_p]),
    "rald_dit_cond_cache_bytes": (c_i64, [c_void_p, c_int]),
    "rald_dit_encode_cond_tokens": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rald_dit_encode_cond": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "rald_dit_denoise": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "rald_dit_sample": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_float, c_float, c_void_p, c_void_p]),
    "rald_dit_profile_begin": (c_int, [c_void_p]),
    "rald_dit_profile_end": (c_int, [c_void_p, C.POINTER(C.c_double), C.POINTER(c_int)]),
    "rald_dit_profile_end_kinds": (c_int, [c_void_p, C.POINTER(C.c_double), C.POINTER(c_int)]),
    "rald_dit_profile_set_kinds": (c_int, [c_void_p, C.c_uint32]),
    "rald_ae_create": (c_int, [C.POINTER(AeConfig), C.POINTER(c_void_p)]),
    "rald_ae_destroy": (None, [c_void_p]),
    "rald_ae_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "rald_ae_finalize": (c_int, [c_void_p]),
    "rald_ae_encode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rald_ae_ctx_bytes": (c_i64, [c_void_p, c_int]),
    "rald_ae_decode_latents": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rald_ae_decode_queries": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p]),
    "rald_radar_create": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, C.POINTER(c_void_p)]),
    "rald_radar_destroy": (None, [c_void_p]),
    "rald_radar_load_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_i64]),
    "rald_radar_finalize": (c_int, [c_void_p]),
    "rald_radar_encode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rald_radar_load_decoder_weight": (c_int, [c_void_p, C.c_char_p, c_void_p, c_i64]),
    "rald_radar_decode": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rald_post_scratch_bytes": (c_i64, [c_i64]),
    "rald_post_occupied_points": (c_int, [c_void_p, c_void_p, c_i64, C.POINTER(C.c_double), c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
    "rald_post_transform_points": (c_int, [c_void_p, c_i64, C.POINTER(C.c_double), c_int, c_int, c_int, c_void_p, c_void_p]),
    "rald_post_chamfer_sums": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p]),
    "rald_post_iou": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p]),
    "rald_query_uniform": (c_int, [c_void_p, c_i64, C.POINTER(C.c_double), c_int, c_int, c_void_p, c_void_p]),
    "rald_query_uniform_cart": (c_int, [c_void_p, c_i64, C.POINTER(C.c_double), C.POINTER(C.c_double), c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rald_query_norm_points": (c_int, [c_void_p, c_i64, C.POINTER(C.c_double), c_int, c_int, c_void_p, c_void_p]),
    "rald_query_refine": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), c_int, c_int, c_int,
                                  c_void_p, c_void_p]),
    "rald_optim_grad_sumsq": (c_int, [c_void_p, c_i64, c_void_p, c_void_p]),
    "rald_optim_clip_coef": (c_int, [c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "rald_optim_adamw_ema": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_double, c_i64, C.c_double, c_int, c_void_p]),
    "rald_optim_ema": (c_int, [c_void_p, c_void_p, c_i64, C.c_double, c_void_p]),
    "rald_radar_cube_prepare": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_float, c_void_p,
                                        c_void_p]),
    "rald_op_gemm_nt": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p,
                                c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "rald_op_gemm_nt2": (c_int, [c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64, c_i64, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "rald_op_transpose": (c_int, [c_void_p, c_int, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_ln_mod_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "rald_op_ln_mod_bwd_cast": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_i64, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "rald_op_geglu_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "rald_op_geglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "rald_op_colsum": (c_int, [c_void_p, c_int, c_i64, c_i64, c_int, c_void_p, c_void_p]),
    "rald_op_row_lse": (c_int, [c_void_p, c_i64, c_int, c_float, c_void_p, c_void_p]),
    "rald_op_rowdot_heads": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p, c_void_p]),
    "rald_op_attn_bwd_elem": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_int, c_float, c_int, c_void_p,
                                      c_void_p, c_void_p]),
    "rald_op_sgemm_acc": (c_int, [c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_void_p, c_i64, c_int, c_int, c_int, c_float, c_void_p]),
    "rald_op_silu_fwd": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
    "rald_op_silu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "rald_op_posemb": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rald_op_edm_loss_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rald_op_conv3d": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_conv_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_groupnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_groupnorm_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_void_p]),
    "rald_op_groupnorm_bwd_scratch_bytes": (c_i64, [c_int, c_int, c_int]),
    "rald_op_groupnorm_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_groupnorm_bwd_cast": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_conv3d_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_conv_in": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_conv_in_wgrad": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rald_op_pad_channels": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "rald_op_zero_insert2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_im2col_t": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_i64, c_int, c_void_p]),
    "rald_op_rowdot": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
    "rald_op_softmax_rows": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_int, c_int, c_void_p]),
    "rald_op_ae_decode_tables": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rald_op_gemm_tn": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rald_op_conv3d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_gemm_tn_workspace_bytes": (c_i64, [c_int, c_int, c_int]),
    "rald_op_gemm_tn_ws": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_int, c_int, c_int, c_void_p, c_i64, c_void_p]),
    "rald_op_conv3d_wgrad_workspace_bytes": (c_i64, [c_int] * 8),
    "rald_op_conv3d_wgrad_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_i64,
                                        c_void_p]),
    "rald_op_patches27": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rald_op_proj_in": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "rald_op_final_norm_proj": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "rald_op_ae_encode_tables": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rald_op_ae_enc_features": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rald_op_attention_f16kv": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                        c_void_p]),
    "rald_op_ae_decode_queries_nw": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_void_p, c_int, c_void_p]),
    "rald_op_attn_self_proj": (c_int, [c_void_p, c_i64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rald_op_xattn_q2_proj": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_float, c_void_p]),
    "rald_op_reduce_resid_ln": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_i64, c_int, c_float,
                                        c_float, c_void_p]),
    "rald_op_gemm_mx8": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_i64, c_i64,
                                 c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "rald_op_quantize_mx8": (c_int, [c_void_p, c_int, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p]),
    "rald_op_layernorm_mx8": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_i64, c_int, c_float, c_float,
                                      c_void_p]),
    "rald_op_layernorm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_void_p]),
    "rald_op_attention": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64,
                                  c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rald_op_attention_split_scratch_bytes": (c_i64, [c_int, c_int, c_int, c_int]),
    "rald_op_attention_split": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64,
                                        c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
    "rald_op_attention_vrow": (c_int, [c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_i64, c_i64,
                                       c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rald_op_attention_bwd": (c_int, [c_void_p, c_i64, c_i64] * 8 + [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rald_op_gemm_resid_ln": (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                                      c_float, c_float, c_int, c_int, c_void_p]),
    "rald_op_cast_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_void_p]),
}


def build_library(verbose: bool = False) -> str:
    """Compile rald_amd/csrc/*.hip for gfx950 into rald_amd/librald_hip.so (hipcc cross-compiles
    without a GPU).  Returns the library path."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-8000:])
    if r.returncode:
        raise RuntimeError("building librald_hip.so failed")
    return LIB_PATH


def lib():
    """The loaded library (cached).  Raises if it is absent - never falls back."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(or `make -C rald_amd/csrc`); rald_amd has no CPU/PyTorch fallback")
            L = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)          # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            _lib = L
        return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().rald_last_error()
        raise RuntimeError(f"librald_hip: {msg.decode() if msg else 'error'} (status {rc})")
