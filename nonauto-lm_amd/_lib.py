"""ctypes binding of libmivae.so -- the C-ABI boundary declared in include/mivae.h.

No torch types cross the boundary: raw device pointers, sizes and the HIP stream handle only.
The library is mandatory: there is no CPU or eager-PyTorch fallback for the hot path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmivae.so")

P = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int
F32 = ctypes.c_float
F64 = ctypes.c_double

# name -> argument ctypes (every entry returns int status unless listed in _RET)
SIGNATURES = {
    "mivae_version": [],
    "mivae_last_error": [],
    "mivae_tokens_mask": [P, P, I64, P],
    "mivae_tokens_mask_f32": [P, P, P, I64, P],
    "mivae_sigma_head_fwd": [P, P, I64, F32, F32, I32, P],
    "mivae_sigma_head_bwd": [P, P, P, I64, F32, F32, I32, P],
    "mivae_posenc_mask": [P, P, P, P, I64, I64, I64, I32, I32, P],
    "mivae_attn_bias": [P, P, I64, I64, I32, I32, P],
    "mivae_attention_supported": [I64, I64],
    "mivae_attention_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I32, F32, P],
    "mivae_attention_bwd": [P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I64, I32, F32,
                            P],
    "mivae_gather_last": [P, P, P, P, I64, I64, I64, I32, P],
    "mivae_reparam_logq_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_reparam_logq_bwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_gauss_logprob_fwd": [P, P, P, I64, I64, I64, I64, F32, F32, I32, P],
    "mivae_gauss_logprob_bwd": [P, P, P, P, I64, I64, I64, I64, F32, F32, I32, P],
    "mivae_segment_sum": [P, P, P, I64, I64, P],
    "mivae_mask_count": [P, P, I64, I64, I64, P],
    "mivae_nice_affine_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I32, I32, P],
    "mivae_nice_affine_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I32, I32, P],
    "mivae_actnorm_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, I32, I32, P],
    "mivae_col_workspace_floats": [I64, I64, I32],
    "mivae_actnorm_bwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_actnorm_stats": [P, P, P, P, I64, I64, I32, P],
    "mivae_colsum": [P, P, P, I64, I64, I32, P],
    "mivae_colsum_partial": [P, P, I64, I64, I32, P],
    "mivae_colsum_sink": [P, P, P, I64, P, I64, I64, I32, P],
    "mivae_colsum_static_max_floats": [],
    "mivae_colsum_static_init": [],
    "mivae_colsum_static": [P, P, I64, I64, I32, P],
    "mivae_kl_fwd": [P, P, P, I64, I32, F32, P],
    "mivae_kl_bwd": [P, P, P, P, P, I64, I32, F32, P],
    "mivae_reparam_logq_fwd_rng": [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_reparam_logq_bwd_rng": [P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_philox_normal": [P, P, I64, P],
    "mivae_posterior_fwd": [P, P, I32, F32, F32, P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_posterior_bwd": [P, P, I32, P, I32, F32, F32, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_latent_tail_supported": [I64, I64, I64],
    "mivae_latent_tail_fwd": [P, P, P, P, P, P, P, I64, I64, I64, I32, F32, F32, F32, I32, P, P, P],
    "mivae_rng_advance": [P, P],
    "mivae_latent_tail_bwd": [P, P, P, P, P, P, P, P, P, I64, I64, I32, F32, F32, F32, I32, P, P],
    "mivae_maf_sequential_bwd": [P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I32, I32, I32, P],
    "mivae_maf_sequential_bwd_supported": [I64, I64],
    "mivae_maf_sequential_bwd_blocks": [I64],
    "mivae_pair_gauss_logdens": [P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_pair_sqdist": [P, P, P, I64, I64, I64, I32, P],
    "mivae_pair_sqdist_bwd": [P, P, P, P, P, I64, I64, I64, I32, I32, P],
    "mivae_rowgemm": [P, P, P, I64, I64, I32, P],
    "mivae_rowgemm_wgrad": [P, P, P, I64, I64, I32, P],
    "mivae_rowgemm_wgrad_slabs": [I64, I64],
    "mivae_slogdet_inv": [P, P, I64, P, P, P],
    "mivae_mixed_stack_supported": [I64, I64],
    "mivae_mixed_stack": [P, P, P, P, P, I64, I64, I64, I64, I64, I32, I32, P],
    "mivae_mixed_stack_train_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I32, P],
    "mivae_mixed_stack_bwd_blocks": [I64, I64, I64, I64, I32],
    "mivae_mixed_stack_bwd": [P, P, P, I32, P, P, I64, P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_nice_dw_steps": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_elbo_fwd": [P, P, P, I64, I64, F32, F32, I32, P, P],
    "mivae_elbo_bwd": [P, P, P, P, P, I64, I64, F32, F32, I32, P, P],
    "mivae_nll_fwd": [P, P, P, P, P, P, P, I64, I64, I64, F32, I32, P],
    "mivae_nll_bwd": [P, P, P, P, P, P, I64, I64, I64, F32, I32, P],
    "mivae_lce_workspace_floats": [I64, I64],
    "mivae_lce_pack_elems": [I64, I64],
    "mivae_lce_pack": [P, P, P, I64, I64, P],
    "mivae_lce_fwd": [P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P],
    "mivae_lce_bwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P],
    "mivae_linear_fwd": [P, P, P, P, I64, I64, I64, P],
    "mivae_linear_fwd_residual": [P, P, P, P, P, I64, I64, I64, I32, P],
    "mivae_linear_dgrad": [P, P, P, I64, I64, I64, P],
    "mivae_linear_wgrad_rows": [I64],
    "mivae_colsum_sink_multi": [P, I64, I64, P],
    "mivae_fold_aug_sink": [P, I64, I64, I64, I64, P, P, P],
    "mivae_linear_wgrad_group": [P, P, P, I64, I64, I64, I64, P],
    "mivae_linear_wgrad_multi": [P, I64, I64, I32, P],
    "mivae_linear_wgrad_single_group": [P, P, P, P, P, P, I64, I64, I64, P],
    "mivae_linear_wgrad_pair": [P, P, P, I64, I64, P, P, P, I64, I64, I64, P],
    "mivae_linear_dgrad_relu": [P, P, P, P, I64, I64, I64, P],
    "mivae_embedding_bwd": [P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_ffn_fwd": [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, F32, I32, P],
    "mivae_linear_fwd_group": [P, P, P, P, I64, I64, I64, I64, P],
    "mivae_linear_dgrad_group": [P, P, P, I64, I64, I64, I64, P],
    "mivae_layernorm_bwd_partial": [P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_layernorm_bwd_partial_add": [P, P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_layernorm_bwd_partial_add_twin": [P, P, P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_lce_pack2": [P, P, P, P, I64, I64, P],
    "mivae_lce_dh_workspace_floats": [I64, I64, I64],
    "mivae_lce_bwd_dh": [P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P],
    "mivae_lce_dw_slabs": [I64, I64, I64],
    "mivae_lce_dw": [P, P, P, P, P, P, P, I64, I64, I64, I64, F32, P],
    "mivae_maf_affine_fwd": [P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_maf_affine_bwd": [P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_maf_sequential_supported": [I64, I64],
    "mivae_maf_sequential": [P, P, P, P, P, P, P, P, I64, I64, I64, I32, I32, I32, P],
    "mivae_linear_wgrad": [P, P, P, I64, I64, I64, P],
    "mivae_layernorm_fwd": [P, P, P, P, P, P, I64, I64, F32, I32, I32, P],
    "mivae_layernorm_workspace_floats": [I64, I64],
    "mivae_layernorm_bwd_sink": [P, P, P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_nice_dw_sink": [P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_layernorm_bwd": [P, P, P, P, P, P, P, P, I64, I64, I32, I32, P],
    "mivae_sumsq_blocks": [I64],
    "mivae_sumsq_partial": [P, P, I64, P],
    "mivae_adam_clip_step": [P, P, P, P, P, P, I64, P, P, I64, F64, F64, F64, F32, F32, F32, F32, F32, P],
    "mivae_scalar_add": [P, F32, P],
    "mivae_accumulate_scalars": [P, P, P, I64, P],
    "mivae_hbm_copy": [P, P, I64, I32, P],
    "mivae_nice_pack_bytes": [I64, I64, I32],
    "mivae_nice_pack": [P, P, I64, I64, I64, I32, I32, P],
    "mivae_nice_stack_fwd": [P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, I32, F32, F32, I32, P],
    "mivae_nice_stack_npad": [I64],
    "mivae_nice_stack_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32, F32, I32, P],
    "mivae_nice_dw_workspace_floats": [I64, I64, I64, I64],
    "mivae_nice_dw": [P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I32, P],
    "mivae_nice_big_supported": [I64, I64, I32],
    "mivae_nice_big_pack_bytes": [I64, I64, I64, I32],
    "mivae_nice_big_pack": [P, P, I64, I64, I64, I32, P],
    "mivae_nice_big_pack_multi": [P, P, I64, I64, I64, P],
    "mivae_nice_big_fwd": [P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, I32, F32, F32, P],
    "mivae_nice_big_dw_workspace_floats": [I64, I64, I64, I64],
    "mivae_nice_big_hidden_elems": [I64, I64, I64, I64],
    "mivae_nice_big_bwd": [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64, I64, I64, I32, I32, F32,
                           F32, P],
}
_RET = {
    "mivae_last_error": ctypes.c_char_p,
    "mivae_col_workspace_floats": I64,
    "mivae_colsum_partial": I64,
    "mivae_colsum_static_max_floats": I64,
    "mivae_nice_pack_bytes": I64,
    "mivae_sumsq_blocks": I64,
    "mivae_maf_sequential_bwd_blocks": I64,
    "mivae_rowgemm_wgrad_slabs": I64,
    "mivae_lce_workspace_floats": I64,
    "mivae_lce_pack_elems": I64,
    "mivae_layernorm_bwd_partial": I64,
    "mivae_layernorm_bwd_partial_add": I64,
    "mivae_layernorm_bwd_partial_add_twin": I64,
    "mivae_lce_dh_workspace_floats": I64,
    "mivae_lce_dw_slabs": I64,
    "mivae_linear_wgrad_rows": I64,
    "mivae_layernorm_workspace_floats": I64,
    "mivae_nice_stack_npad": I64,
    "mivae_nice_dw_workspace_floats": I64,
    "mivae_mixed_stack_bwd_blocks": I64,
    "mivae_nice_big_pack_bytes": I64,
    "mivae_nice_big_dw_workspace_floats": I64,
    "mivae_nice_big_hidden_elems": I64,
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python nonauto-lm_amd/build.py` (hipcc, gfx950). "
                "The HIP extension is mandatory; there is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the .so lacks a declared symbol
            fn.argtypes = args
            fn.restype = _RET.get(name, ctypes.c_int)
        _lib = handle
    return _lib


def dtype_code(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise TypeError(f"libmivae supports float32 and bfloat16 latents, got {dtype}")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "nonauto_lm_amd: the latent hot path runs only on the GPU (HIP kernels via "
                "libmivae.so); got a CPU tensor. There is no CPU fallback by design.")


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().mivae_last_error().decode()}")


def query(name, *args):
    return getattr(lib(), name)(*args)
