"""ctypes binding of libvitmi_hip.so (the C ABI declared in include/vitmi.h).

The library is the only compute path of this package: if it is missing or fails to load the
import of this module raises, there is no fallback.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_void_p

# torch first: it ships its own copy of the HIP runtime, and the library has to bind to the copy torch's streams and
# allocations live in.  Loaded the other way round the process holds two runtimes and every launch of this library
# fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# VITMI_LIB: another build of the same library, for A/B timing of kernel versions on one box (tools/)
LIB_PATH = os.environ.get("VITMI_LIB") or os.path.join(_HERE, "libvitmi_hip.so")

F32, BF16 = 0, 1
EPI_NONE, EPI_GELU, EPI_DGELU, EPI_GELU_D, EPI_MULAUX = 0, 1, 2, 3, 4
FAM_GEMM, FAM_ATTN_FWD, FAM_ATTN_BWD, FAM_LAYERNORM, FAM_TOKENS, FAM_OTHER = range(6)
FAM_GEMM_VARIANT0, FAM_COUNT = 8, 29

P, I, L, F = c_void_p, c_int, c_longlong, c_float
U64 = ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/vitmi.h one to one
SIGNATURES = {
    "vitmi_version": (I, []),
    "vitmi_last_error": (c_char_p, []),
    "vitmi_prof_enable": (I, [I]),
    "vitmi_prof_reset": (I, []),
    "vitmi_prof_select": (I, [I]),
    "vitmi_prof_stride": (I, [I]),
    "vitmi_prof_read": (I, [I, ctypes.POINTER(c_double), ctypes.POINTER(c_double), ctypes.POINTER(c_longlong)]),
    "vitmi_prof_name": (c_char_p, [I]),
    "vitmi_cast": (I, [P, I, P, I, L, P]),
    "vitmi_gemm": (I, [I, I, I, I, P, L, L, P, L, L, P, L, I, I, P, P, L, P, L, P]),
    "vitmi_gemm_colsum_workspace_bytes": (L, [I, I]),
    "vitmi_gemm_colsum": (I, [I, I, I, I, P, L, L, P, L, L, P, L, I, I, P, P, L, P, L, P, P, P]),
    "vitmi_gemm_wgrad_workspace_bytes": (L, [I, I, I, I]),
    "vitmi_gemm_wgrad": (I, [I, I, I, I, P, L, P, L, P, L, P, P]),
    "vitmi_gemm_wgrad_pair_workspace_bytes": (L, [I, I, I, I, I, I]),
    "vitmi_gemm_wgrad_pair": (I, [I, I, I, I, P, L, P, L, P, I, I, P, L, P, L, P, P, P]),
    "vitmi_debug_set_tn_batch": (I, [I]),
    "vitmi_row_stats": (I, [P, L, P, P, I, L, I, F, P]),
    "vitmi_gemm_ln": (I, [I, I, I, P, L, P, P, P, P, P, P, L, I, P]),
    "vitmi_fold_ln": (I, [P, P, P, P, P, P, P, I, I, P]),
    "vitmi_attn_block_infer_workspace_bytes": (L, [I, I, I, I, I]),
    "vitmi_attn_block_infer": (I, [I, I, I, I, I, P, F, P, P, P, P, P, P, P, P, P]),
    "vitmi_ff_block_infer_workspace_bytes": (L, [I, I, I]),
    "vitmi_ff_block_infer": (I, [I, I, I, P, F, P, P, P, P, P, P, P, P]),
    "vitmi_transpose_cast": (I, [P, I, P, I, I, P]),
    "vitmi_debug_set_nt_config": (I, [I]),
    "vitmi_debug_set_tn_config": (I, [I]),
    "vitmi_set_reserved_cus": (I, [I]),
    "vitmi_debug_set_nt_group": (I, [I]),
    "vitmi_debug_set_nt_tail_split": (I, [I]),
    "vitmi_debug_set_attn_head_groups": (I, [I]),
    "vitmi_debug_set_attn_bwd_fused": (I, [I]),
    "vitmi_debug_set_attn_f32_mfma": (I, [I]),
    "vitmi_debug_set_gelu_pair": (I, [I]),
    "vitmi_set_gelu_mode": (I, [I]),
    "vitmi_get_gelu_mode": (I, []),
    "vitmi_debug_set_nt_stagger": (I, [I, I, I]),
    "vitmi_colsum_workspace_bytes": (L, [L, I]),
    "vitmi_colsum": (I, [P, I, L, I, L, P, P, P]),
    "vitmi_patchify": (I, [P, I, P, I, I, I, I, I, I, P]),
    "vitmi_rollout_workspace_bytes": (L, [I, I, I]),
    "vitmi_rollout": (I, [P, I, I, I, I, I, I, I, I, I, P, P, P]),
    "vitmi_grad_rollout": (I, [P, P, I, I, I, I, I, I, P, P, P]),
    "vitmi_rollout_last_attn": (I, [P, I, I, I, I, I, P, P]),
    "vitmi_ff_block_fwd": (I, [I, I, I, I, P, P, P, F, P, P, P, P, P, P, P, P, P, P, P]),
    "vitmi_ff_block_bwd_workspace_bytes": (L, [I, I, I, I]),
    "vitmi_ff_block_bwd": (I, [I, I, I, I] + [P] * 21),
    "vitmi_attn_block_fwd": (I, [I, I, I, I, I, I, P, P, P, F] + [P] * 12 + [I, P]),
    "vitmi_attn_block_qkv_layout": (I, [I, I, I, I, I, I]),
    "vitmi_debug_set_qkv_planes": (I, [I]),
    "vitmi_debug_set_small_fused": (I, [I]),
    "vitmi_debug_set_wgrad_overlap": (I, [I]),
    "vitmi_attn_block_bwd_workspace_bytes": (L, [I, I, I, I, I, I]),
    "vitmi_attn_block_bwd": (I, [I, I, I, I, I, I] + [P] * 21 + [I, P]),
    "vitmi_transpose_cast_multi": (I, [P, P, I, L, P]),
    "vitmi_debug_set_ghost_norms": (I, [I]),
    "vitmi_wgrad_sqnorm_workspace_bytes": (L, [I, I, I, I, I]),
    "vitmi_wgrad_sqnorm": (I, [I, I, I, I, I, P, L, P, L, P, P, P]),
    "vitmi_seg_colsum_sqnorm": (I, [P, I, L, I, L, I, P, P, P]),
    "vitmi_seg_layernorm_sqnorm": (I, [P, P, L, P, P, I, L, I, I, P, P, P, P]),
    "vitmi_seg_combine": (I, [P, I, I, L, L, P, P, P]),
    "vitmi_seg_scale_rows": (I, [P, I, L, I, L, I, P, P, L, P]),
    "vitmi_seg_sqnorm": (I, [P, I, L, I, L, I, P, P]),
    "vitmi_seq_ord": (I, [P, I, P, I, L, P]),
    "vitmi_jigsaw_gather": (I, [P, P, I, I, I, I, I, I, P, P, I, P, P, P]),
    "vitmi_patch_weighted_mean": (I, [P, I, I, I, I, I, I, P, P, P]),
    "vitmi_patch_embed_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, I, L, P]),
    "vitmi_patch_embed_bwd_w_workspace_bytes": (L, [I, I, I, I, I, I, I]),
    "vitmi_patch_embed_bwd_w": (I, [P, I, P, L, P, P, I, I, I, I, I, I, I, P]),
    "vitmi_patch_wide_group": (I, [I, I, I, I, I, I, I, I]),
    "vitmi_patch_wide_weight": (I, [P, P, P, P, I, I, I, P]),
    "vitmi_patch_embed_fwd_wide": (I, [P, P, P, P, I, I, I, I, I, I, L, P]),
    "vitmi_patch_embed_bwd_w_wide_workspace_bytes": (L, [I, I, I, I, I, I]),
    "vitmi_patch_embed_bwd_w_wide": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "vitmi_tokens_assemble_fwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "vitmi_tokens_assemble_bwd_workspace_bytes": (L, [I, I]),
    "vitmi_tokens_assemble_bwd": (I, [P, P, P, P, I, P, I, I, I, I, P]),
    "vitmi_layernorm_fwd": (I, [P, L, P, P, P, P, P, I, L, I, F, P]),
    "vitmi_layernorm_bwd_workspace_bytes": (L, [L, I]),
    "vitmi_layernorm_bwd": (I, [P, P, L, P, P, P, P, P, P, P, P, P, I, L, I, P]),
    "vitmi_layernorm_bwd_static": (I, [P, P, L, P, P, P, P, P, P, P, P, I, L, I, P]),
    "vitmi_attn_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "vitmi_attn_bwd_workspace_bytes": (L, [I, I, I, I, I]),
    "vitmi_attn_bwd_dprobs_workspace_bytes": (L, [I, I, I, I, I]),
    "vitmi_attn_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "vitmi_meanpool_fwd": (I, [P, P, I, I, I, I, P]),
    "vitmi_meanpool_bwd": (I, [P, P, I, I, I, I, P]),
    "vitmi_add": (I, [P, P, P, I, L, P]),
    "vitmi_scale_add_clip": (I, [P, P, P, I, L, F, F, F, P]),
    "vitmi_dropout": (I, [P, P, I, L, F, U64, P]),
    "vitmi_adam_step": (I, [P, P, P, P, P, L, F, F, F, F, F, I, F, P]),
    "vitmi_adam_step_multi": (I, [P, I, P, I, I, F, F, F, F, F, I, F, P]),
    "vitmi_adam_step_matrices": (I, [P, P, I, L, F, F, F, F, F, I, F, P]),
    "vitmi_gemm_wgrad_multi_workspace_bytes": (L, [P, I, L]),
    "vitmi_gemm_wgrad_multi": (I, [P, I, L, P, P]),
    "vitmi_ff_block_bwd_du_offset": (L, [I, I, I, I]),
    "vitmi_attn_block_bwd_dqkv_offset": (L, [I, I, I, I, I, I]),
    "vitmi_small_sums_multi": (I, [P, I, P]),
    "vitmi_ff_block_bwd_sums_offset": (L, [I, I, I, I]),
    "vitmi_ff_block_bwd_sums_strips": (I, [I, I, I, I]),
    "vitmi_attn_block_bwd_sums_offset": (L, [I, I, I, I, I, I]),
    "vitmi_attn_block_bwd_sums_strips": (I, [I, I, I, I, I, I]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python visiontransformer_amd/build.py` "
            "(hipcc --offload-arch=gfx950).  This package has no other compute path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("VITMI_LIB") and not hasattr(lib, name):
            continue             # an older build loaded for A/B timing may lack the newest entry points
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class VitmiError(RuntimeError):
    pass


def check(rc, what=""):
    """Turn a negative C-ABI return code into a Python exception carrying the library's message."""
    if rc != 0:
        msg = lib.vitmi_last_error()
        raise VitmiError(f"{what or 'vitmi'} failed (code {rc}): {msg.decode() if msg else ''}")


def prof_read(family):
    ms, work, n = c_double(), c_double(), c_longlong()
    check(lib.vitmi_prof_read(family, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n)), "vitmi_prof_read")
    return ms.value, work.value, n.value
