"""ctypes binding of libvbx_hip.so (the C ABI declared in include/vbx.h).

The product path has NO fallback: if the shared library is missing or a device is not
gfx950, calls raise.  `lib()` loads lazily so that CPU-only host logic (mask helpers, DP
bucket logic, state-dict handling) stays importable without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VBX_LIB_PATH") or os.path.join(_HERE, "lib", "libvbx_hip.so")  # override: A/B builds of the library

P, I, L, F = C.c_void_p, C.c_int, C.c_long, C.c_float

VBX_GEMM_NT, VBX_GEMM_NN, VBX_GEMM_TN = 0, 1, 2
VBX_EPI_BF16, VBX_EPI_F32, VBX_EPI_QKV, VBX_EPI_GEGLU, VBX_EPI_SPLITK = 0, 1, 2, 3, 4


class GemmDesc(C.Structure):
    _fields_ = [
        ("mode", I), ("epilogue", I), ("M", I), ("N", I), ("K", I), ("lda", I), ("ldb", I), ("ldc", I),
        ("A", P), ("B", P), ("C", P), ("bias", P), ("resid", P), ("C2", P), ("splits", I),
        ("Np", I), ("H", I), ("qk_scale", F), ("q_gamma", P), ("k_gamma", P), ("rot_cos", P), ("rot_sin", P),
        ("q16", P), ("k16", P), ("qb", P), ("kb", P), ("v", P), ("q_rnorm", P), ("k_rnorm", P), ("f16", I), ("v16", P), ("C3", P),
        ("q_prescale", F), ("delta_o", P), ("delta", P),
    ]


# name -> argtypes (every function returns int unless listed in _RET)
_PROTOS = {
    "vbx_version": [],
    "vbx_check_device": [I],
    "vbx_gemm": [C.POINTER(GemmDesc), P],
    "vbx_gemm_select": [I],
    "vbx_gemm5_cu_limit": [I],
    "vbx_prof_enable": [I],
    "vbx_gemm_tn_splitk_grouped": [C.POINTER(GemmDesc), I, P],
    "vbx_splitk_reduce": [P, I, I, I, P, I, I, I, I, I, I, P],
    "vbx_rmsnorm_fwd": [P, P, P, L, P, P, I, I, I, I, I, P],
    "vbx_rmsnorm_bwd_chunks": [I],
    "vbx_rmsnorm_bwd": [P, P, L, P, P, P, P, P, P, I, I, I, I, I, P],
    "vbx_attn_fwd": [P, P, P, P, P, P, P, I, I, I, F, P],
    "vbx_attn_bwd": [P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, F, P, P],
    "vbx_attn_bwd_fused": [P, P, P, P, P, P, P, I, P, P, P, P, P, P, P, P, P, F, P, I, P, I, I, I, F, P, P],
    "vbx_attn_bwd_select": [I],
    "vbx_attn_bwd_variant": [],
    "vbx_dropout_bits_words": [I],
    "vbx_attn_dropout_bits": [P, P, I, I, C.c_ulonglong, C.c_uint, F, P],
    "vbx_dropout_rows": [P, P, L, I, I, C.c_ulonglong, C.c_uint, F, P],
    "vbx_attn_fwd_dropout": [P, P, P, P, P, P, P, I, I, I, F, P, F, P],
    "vbx_attn_bwd_dropout": [P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, F, P, P, F, P],
    "vbx_attn_bwd_fused_tiles": [I],
    "vbx_qknorm_rope_bwd": [P, P, P, P, P, P, P, P, P, P, F, P, I, P, I, I, I, F, P],
    "vbx_qknorm_rope_bwd_gpart_rows": [I],
    "vbx_pack_embed_input": [P, P, P, P, P, I, I, I, P],
    "vbx_pack_embed_input_text": [P, P, P, P, P, P, I, P, I, L, P, P, I, I, I, P],
    "vbx_cond_emb_bwd": [P, I, P, I, P, L, P, I, I, I, P],
    "vbx_pack_phoneme_input": [P, P, I, P, I, P, P, P, P, I, I, I, P],
    "vbx_rowdot": [P, P, P, P, L, I, P],
    "vbx_stack_input": [P, P, P, I, I, I, I, P],
    "vbx_stack_input_bwd": [P, P, P, I, I, I, I, P],
    "vbx_unet_cat": [P, P, F, P, P, L, I, P],
    "vbx_unet_split": [P, F, P, P, P, L, I, P],
    "vbx_unet_addskip": [P, P, P, L, P],
    "vbx_rmsnorm_fwd_f32": [P, P, P, L, P, I, I, I, I, I, P],
    "vbx_convpos_fwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "vbx_convpos_fwd_libm": [P, P, P, P, P, P, I, I, I, I, I, P],
    "vbx_convpos_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P],
    "vbx_convpos_bwd_chunks": [I, I],
    "vbx_conv_wgrad_finalize": [P, I, I, I, P, P, P],
    "vbx_time_embed_fwd": [P, P, P, P, P, P, P, I, I, I, P],
    "vbx_time_embed_bwd_scratch_floats": [I, I],
    "vbx_masked_mse_scratch_floats": [I],
    "vbx_time_embed_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, P],
    "vbx_adaln_proj_fwd": [P, P, P, P, I, I, I, I, P],
    "vbx_adaln_proj_bwd": [P, P, P, P, P, P, P, I, I, I, I, P],
    "vbx_adaln_proj_bwd_scratch_floats": [I, I, I],
    "vbx_adaln_dtemb_all": [P, P, P, P, I, I, I, I, P],
    "vbx_reduce_norm_partials": [P, P, L, I, I, I, I, P],
    "vbx_reduce_col_partials": [P, P, P, I, I, I, P],
    "vbx_gateloop_scan_fwd": [P, P, P, I, I, I, P],
    "vbx_gateloop_scan_bwd": [P, P, P, P, I, I, I, P],
    "vbx_layernorm_fwd": [P, P, P, P, P, L, I, F, P],
    "vbx_layernorm_bwd": [P, P, P, P, P, I, I, I, F, P],
    "vbx_geglu_bwd_colsum": [P, P, P, I, I, P, P],
    "vbx_geglu_bwd_colsum_slabs": [],
    "vbx_colsum_bf16_partials": [P, I, I, I, P, P],
    "vbx_colsum_slabs": [],
    "vbx_geglu_bwd": [P, P, P, I, I, P],
    "vbx_colsum_bf16": [P, I, I, I, P, I, I, I, P, P],
    "vbx_colsum_f32": [P, I, I, I, P, P, P],
    "vbx_colsum_scratch_floats": [I, I],
    "vbx_sum_rows_f32": [P, L, L, P, L, I, P],
    "vbx_masked_mse_fwd": [P, P, P, P, P, I, I, I, P],
    "vbx_masked_mse_bwd": [P, P, P, P, P, P, P, I, I, I, P],
    "vbx_cfm_inputs": [P, P, P, F, P, P, I, L, P],
    "vbx_axpy_dev": [P, P, P, I, P, L, P],
    "vbx_ode_set_time": [P, I, P, P, I, P],
    "vbx_axpy_ctr": [P, P, P, P, I, P, L, P],
    "vbx_counter_add": [P, I, P],
    "vbx_ada_select": [P, I, I, I, P, P, I, P],
    "vbx_stream_delay": [F, P],
    "vbx_pack_weight": [P, I, I, P, P, I, I, I, I, P],
    "vbx_pack_bias": [P, I, P, I, I, I, P],
    "vbx_adam_step": [P, P, P, P, L, F, F, F, F, I, P, P],
    "vbx_sumsq": [P, L, P, P, P],
    "vbx_clip_coef": [P, F, F, P, P],
    "vbx_split3_f16": [P, L, I, L, P, I, P],
    "vbx_pack_weight3": [P, I, I, P, I, I, I, I, P],
    "vbx_qknorm_rope_f32": [P, I, I, I, F, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, F, P],
    "vbx_attn_fwd_f32": [P, P, P, P, P, P, P, P, I, I, I, F, P],
    "vbx_geglu_f32": [P, P, P, P, P, L, I, P],
    "vbx_adaln_proj_f32": [P, P, P, P, I, I, I, I, P],
    "vbx_probe_tr16": [P, P, P, P],
    "vbx_probe_mfma": [I, P, P, P, P],
}

_lib = None


class VbxError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VbxError(
            f"{LIB_PATH} not found: the HIP extension is REQUIRED (no CPU/eager fallback exists). "
            "Build it with `python voicebox-pytorch_amd/build.py` (needs hipcc).")
    if not os.environ.get("VBX_LIB_PATH"):  # a library that was not rebuilt after a source edit must not be mistaken for the product
        from . import build as _build

        rec, cur = _build.recorded_hash(), _build.source_hash()
        if rec is None or cur is None:  # a library without its stamp, or shipped without csrc/: cannot tell -- say so, do not refuse
            import sys
            print(f"voicebox_pytorch_amd: {LIB_PATH} has no source-hash stamp; cannot check that it matches csrc/", file=sys.stderr)
        elif rec != cur:
            raise VbxError(f"{LIB_PATH} is stale: csrc/ or include/vbx.h changed since it was built.  "
                           "Rebuild with `python voicebox-pytorch_amd/build.py`.")
    l = C.CDLL(LIB_PATH)
    l.vbx_last_error.restype = C.c_char_p
    l.vbx_last_error.argtypes = []
    for name, argtypes in _PROTOS.items():
        fn = getattr(l, name)  # AttributeError here == header/library mismatch
        fn.argtypes = argtypes
        fn.restype = I
    l.vbx_attn_bwd_scratch_bytes.argtypes = [I, I, I]
    l.vbx_attn_bwd_scratch_bytes.restype = C.c_size_t
    l.vbx_dropout_keep_scale.argtypes = [F]
    l.vbx_dropout_keep_scale.restype = F
    l.vbx_attn_q_prescale.argtypes = [F]
    l.vbx_attn_q_prescale.restype = F
    l.vbx_adaln_dtemb_all_scratch_floats.argtypes = [I, I, I, I]
    l.vbx_adaln_dtemb_all_scratch_floats.restype = C.c_long
    _lib = l
    return l


def exported_symbols():
    return sorted(_PROTOS) + ["vbx_last_error", "vbx_attn_bwd_scratch_bytes", "vbx_dropout_keep_scale", "vbx_attn_q_prescale",
                              "vbx_adaln_dtemb_all_scratch_floats"]  # + the stage-level entries bound in engine.py


def ptr(t):
    """torch tensor / None -> raw device pointer (int / None)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream  # the HIP stream under PyTorch-ROCm


def call(name, *args):
    """Invoke a C-ABI entry point, converting tensors to pointers; raises on a non-zero return."""
    l = lib()
    conv = [ptr(a) if hasattr(a, "data_ptr") else a for a in args]
    rc = getattr(l, name)(*conv)
    if rc != 0:
        raise VbxError(f"{name} failed (rc={rc}): {l.vbx_last_error().decode()}")
    return rc
