"""ctypes binding of the C ABI in include/ssl_audio_hip.h (libssl_audio_hip.so, gfx950).

There is NO fallback: if the shared library is missing or a call fails, this raises.  The oracle/ package is
never imported from here (tests/test_boundary.py enforces it).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SA_HIP_LIB") or os.path.join(_HERE, "csrc", "libssl_audio_hip.so")     # (SA_HIP_LIB: A/B builds of the library)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ssl_audio_hip.h")

P, I32, I64, F32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class SaGemmArgs(C.Structure):
    """Mirror of `struct SaGemmArgs` (include/ssl_audio_hip.h)."""
    _fields_ = [
        ("A", P), ("lda", I64), ("a_kmajor", I32),
        ("B", P), ("ldb", I64), ("b_kmajor", I32),
        ("M", I32), ("N", I32), ("K", I32),
        ("alpha", F32),
        ("bias", P),
        ("act", I32),
        ("aux_in", P), ("aux_out", P), ("ldaux", I64),
        ("residual", P), ("ldr", I64), ("res_mod", I32),
        ("out_f32", P), ("ldo_f32", I64),
        ("out_bf16", P), ("ldo_bf16", I64),
        ("row_group", I32), ("split_k", I32), ("accumulate", I32), ("tile256", I32),
        ("colsum_out", P), ("colsum_ws", P),
        ("splitk_ws", P),
        ("asum_out", P), ("asum_ws", P), ("asum_skip_lo", I32), ("asum_skip_hi", I32),
    ]


_SIGNATURES = {
    "sa_abi_version": [],
    "sa_device_info": [C.c_char_p, I32, C.POINTER(I32)],
    "sa_gemm_bf16": [C.POINTER(SaGemmArgs), P],
    "sa_gemm_wgrad_group": [C.POINTER(SaGemmArgs), I32, P],
    "sa_cast_f32_to_bf16": [P, P, I64, P],
    "sa_cast_bf16_to_f32": [P, P, I64, P],
    "sa_transpose_bf16": [P, I32, I32, P, P],
    "sa_transpose_bf16_batch": [P, I32, I32, P],
    "sa_colsum_bf16": [P, I64, I32, I32, P, I32, P, I32, I64, P],
    "sa_colsum_workspace_bytes": [I32, I32],
    "sa_layernorm_fwd": [P, I64, P, P, P, P, I64, P, P, I32, I32, F32, P],
    "sa_layernorm_bwd": [P, I32, I64, P, I64, P, P, P, P, I64, P, P, I64, P, P, P, P, I32, I32, P],
    "sa_layernorm_bwd_workspace_bytes": [I32, I32],
    "sa_set_cu_budget": [I32],
    "sa_set_dynamic_tiles": [I32],
    "sa_lars_step": [P, P, P, I64, F32, F32, F32, F32, I32, P, P, P],
    "sa_gemm_colsum_workspace_bytes": [I32, I32],
    "sa_gemm_splitk_workspace_bytes": [I32, I32, I32],
    "sa_mix_gaussian_noise": [P, P, I64, F32, F32, P, P],
    "sa_running_norm": [P, I32, I64, P, I32, I32, F32, P, P],
    "sa_token_group_sum": [P, I32, I32, I32, I32, I32, I32, I32, F32, I32, P, P],
    "sa_mean_tokens_fwd": [P, I32, I32, I32, P, P],
    "sa_mean_tokens_bwd": [P, I32, I32, I32, P, P],
    "sa_mae_unshuffle_fwd": [P, I32, P, P, P, I32, I32, I32, P, P],
    "sa_mae_unshuffle_bwd": [P, I32, P, I32, I32, I32, P, P, P, P],
    "sa_mae_recon_loss_fwd": [P, I64, I32, P, P, I32, I32, I32, I32, I32, I32, P, P, P, P],
    "sa_mae_unshuffle_bwd_workspace_bytes": [],
    "sa_mae_recon_loss_workspace_bytes": [],
    "sa_bt_loss_workspace_bytes": [],
    "sa_mae_recon_loss_bwd": [P, I64, I32, P, P, P, P, I32, I32, I32, I32, I32, I32, P, P],
    "sa_mae_recon_loss_finalize": [P, P, P],
    "sa_maxpool3s2_fwd": [P, I32, I32, I32, I32, P, P, P, P],
    "sa_maxpool3s2_bwd": [P, P, I32, I32, I32, I32, P, P],
    "sa_subsample_fwd": [P, I32, I32, I32, I32, I32, I32, P, P],
    "sa_subsample_bwd_add": [P, I64, I32, I32, I32, I32, I32, I32, P, P],
    "sa_add_relu_fwd": [P, P, I64, P, P, P],
    "sa_relu_bwd": [P, P, P, I64, P, P],
    "sa_avgpool_fwd": [P, I32, I32, I32, P, P],
    "sa_avgpool_bwd": [P, I32, I32, I32, P, P],
    "sa_conv3x3_c1_fwd": [P, I32, I32, I32, I32, I32, P, P, I32, P, P],
    "sa_conv3x3_c1_wgrad": [P, I32, I32, I32, I32, I32, P, I32, P, P, P],
    "sa_im2col3x3_bf16": [P, I32, I32, I32, I32, I32, I32, P, I32, P],
    "sa_col2im3x3_f32": [P, I32, I32, I32, I32, I32, I32, I32, P, P],
    "sa_bn_tall_workspace_bytes": [I64, I32],
    "sa_bn_colstats_tall": [P, I64, I64, I32, P, P, P, P],
    "sa_bn_bwd_stats_tall": [P, I32, I64, P, I64, I64, I32, P, P, P, P, I32, P, P, P, P],
    "sa_maxpool2_fwd": [P, I32, I32, I32, I32, P, P, P],
    "sa_maxpool2_bwd": [P, P, I32, I32, I32, I32, P, P],
    "sa_relu_mask_fwd": [P, I64, I64, I32, P, F32, P, I64, P, I64, P],
    "sa_relu_mask_bwd": [P, I64, P, I64, I64, I32, P, F32, P, I64, P],
    "sa_nhwc_to_frames": [P, I32, I32, I32, I32, P, P, I64, P],
    "sa_frames_to_nhwc": [P, I64, P, I64, I32, I32, I32, I32, P, P],
    "sa_meanmax_time_fwd": [P, I32, I32, I32, P, P, P],
    "sa_meanmax_time_bwd": [P, P, I32, I32, I32, P, P],
    "sa_se_fwd": [P, I32, I32, I32, P, P, I32, P, P, P, P, P],
    "sa_se_bwd": [P, P, I32, I32, I32, P, P, I32, P, P, P, P, P, P, P, P],
    "sa_attention_fwd": [P, I64, I64, I32, I32, I32, I32, F32, P, I64, P, P],
    "sa_attention_bwd": [P, I64, I64, I32, I32, I32, I32, F32, P, P, I64, P, P, P],
    "sa_bn_colstats": [P, I64, I32, I32, P, P, P],
    "sa_bn_apply": [P, I64, I32, I32, P, P, P, P, I32, P, P, I64, P],
    "sa_bn_bwd_stats": [P, I32, I64, P, I64, I32, I32, P, P, P, P, I32, P, P, P],
    "sa_bn_bwd_apply": [P, I32, I64, P, I64, I32, I32, P, P, P, P, I32, P, P, F32, P, P, P, I64, P],
    "sa_bn_finalize": [P, I64, I32, I32, I32, F32, F32, P, P, P, P, P],
    "sa_matmul_f32": [P, I64, I64, P, I64, I64, P, I64, I32, I32, I32, F32, P],
    "sa_bt_loss_grad": [P, I32, F32, F32, I32, P, P, P, P],
    "sa_bt_stats2": [P, P, I64, I32, I32, P, P],
    "sa_bt_corr": [P, P, I64, I32, I32, P, I32, F32, F32, F32, P, P, P, P, P, P, P, P],
    "sa_bt_bwd_products": [P, P, I32, I32, P, F32, P, P, P],
    "sa_bt_bwd_apply": [P, P, I32, I32, P, P, P, F32, P, P, P, P],
    "sa_adamw_step": [P, P, P, P, I64, F32, F32, F32, F32, F32, I32, F32, P, P],
    "sa_ema_update": [P, P, I64, F32, P],
    "sa_adamw_step_dev": [P, P, P, P, I64, P, F32, F32, F32, F32, F32, P, P, P],
    "sa_ema_update_gated": [P, P, I64, F32, P, P],
    "sa_axpy_f32": [P, P, I64, F32, P],
    "sa_count_nonfinite": [P, I64, P, P],
    "sa_logmel_fwd": [P, I64, I32, I32, P, P, P, P, P, P, I64, I32, I32, P, P, P, F32, F32, I32, P],
    "sa_augment_views": [P, I64, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, F32, I32, P, P],
    "sa_normalize_batch": [P, P, I64, F32, P, F32, F32, P],
    "sa_patchify_bf16": [P, P, I32, I32, I32, I32, I32, P],
    "sa_fill_cls": [P, I32, I64, I32, P, P, P],
    "sa_cls_grad": [P, I32, I64, I32, P, P],
    "sa_gather_rows": [P, I64, I32, P, I32, P, I64, I32, I32, I32, P],
    "sa_scatter_add_rows": [P, I64, I32, P, I32, P, I64, I32, I32, I32, P],
}


def header_symbols():
    """Every `sa_*` function the public header declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(sa_\w+)\s*\(", text)))


class HipLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ssl_audio_amd/csrc`).  ssl_audio_amd has no CPU or PyTorch fallback.")
    h = C.CDLL(LIB_PATH)
    h.sa_last_error.restype = C.c_char_p
    h.sa_last_error.argtypes = []
    for name in header_symbols():
        if name == "sa_last_error":
            continue
        fn = getattr(h, name)  # AttributeError here = header/library mismatch, let it surface
        fn.restype = I64 if name.endswith("_bytes") else I32
        sig = _SIGNATURES.get(name)
        if sig is not None:
            fn.argtypes = sig
    _lib = h
    return h


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().sa_last_error().decode()}")
