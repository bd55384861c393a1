"""ctypes binding of liblcm_hip.so (the C ABI declared in include/lcmhip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C chimera-llm_amd/csrc`.
There is no CPU fallback: importing this module without the library raises.
"""

import ctypes as C
import os
from typing import List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LCMHIP_LIB selects an experimental build of the same ABI (kernel ablation studies); never a fallback
LIB_PATH = os.environ.get("LCMHIP_LIB") or os.path.join(_HERE, "liblcm_hip.so")

F32, BF16 = 0, 1
NORM_LAYERNORM, NORM_RMS, NORM_FP32, NORM_UNIT = 0, 1, 2, 3
FFN_RELU, FFN_SWIGLU = 0, 1
# LCMHIP_ACT_*: inner activation of the two-matrix FFN, by the reference's name (lcm/nn/initialization.py:121-151)
ACT = {"relu": 0, "tanh": 1, "elu": 2, "leaky_relu": 3, "selu": 4, "gelu": 5, "silu": 6, "softsign": 7, "sigmoid": 8,
       "hardsigmoid": 9, "prelu": 10}   # prelu: an FFN's inner activation only (its slope is a tensor of the layer)
POS_NONE, POS_LEARNED, POS_ROPE, POS_SINE = 0, 1, 2, 3
ORDER = {"pre": 0, "post": 1, "normformer": 2}   # LCMHIP_ORDER_*
PRED = {"sample": 0, "epsilon": 1, "v_prediction": 2}
EPI_STORE, EPI_STORE_F32, EPI_RELU, EPI_SWIGLU, EPI_RESID = 0, 1, 2, 3, 4
STATIC_SCRATCH_BYTES = 16 << 20      # LCMHIP_STATIC_SCRATCH_BYTES of include/lcmhip.h
ABI_VERSION = 401                    # LCMHIP_ABI_VERSION of include/lcmhip.h: the layouts and signatures bound below
# LCMHIP_SESSION_* flags of lcmhip_session_workspace_bytes / lcmhip_session_create
SESSION_KEEP_CONTEXT, SESSION_SEPARATE_QKV, SESSION_SEPARATE_ROPE, SESSION_NO_GRAPH = 1, 2, 4, 8

EXPORTS = [
    "lcmhip_last_error", "lcmhip_version",
    "lcmhip_model_create", "lcmhip_model_bind_weight", "lcmhip_model_finalize", "lcmhip_model_destroy",
    "lcmhip_session_workspace_bytes", "lcmhip_session_create", "lcmhip_session_destroy",
    "lcmhip_session_set_schedule", "lcmhip_session_reset", "lcmhip_context_append", "lcmhip_base_forward", "lcmhip_denoise", "lcmhip_denoise_stochastic",
    "lcmhip_denoise_positions",
    "lcmhip_session_reorder", "lcmhip_session_context_len", "lcmhip_session_num_active",
    "lcmhip_session_read_context", "lcmhip_session_profile_enable", "lcmhip_session_profile",
    "lcmhip_normalize", "lcmhip_denormalize", "lcmhip_rfft_pack", "lcmhip_norm_modulate", "lcmhip_gemm", "lcmhip_static_init", "lcmhip_gemm_static", "lcmhip_gemm_partial",
    "lcmhip_residual_norm",
    "lcmhip_ddim_cfg_step", "lcmhip_attention", "lcmhip_finalize_step", "lcmhip_stream_copy",
]


class TowerDesc(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("num_heads", C.c_int32), ("ffn_dim", C.c_int32),
                ("norm_kind", C.c_int32), ("ffn_kind", C.c_int32), ("pos_kind", C.c_int32),
                ("out_proj_bias", C.c_int32), ("rope_theta", C.c_float), ("qk_layernorm", C.c_int32),
                ("ffn_activation", C.c_int32), ("norm_order", C.c_int32), ("no_final_norm", C.c_int32),
                ("scale_residual", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("sonar_dim", C.c_int32), ("model_dim", C.c_int32), ("max_seq_len", C.c_int32),
                ("timestep_freq_dim", C.c_int32), ("dtype", C.c_int32), ("norm_eps", C.c_float),
                ("context", TowerDesc), ("denoiser", TowerDesc), ("embed_scale", C.c_float),
                ("postnet_activation", C.c_int32)]


class DenoiseParams(C.Structure):
    _fields_ = [("guidance_scale", C.c_float), ("guidance_rescale", C.c_float), ("ddim_eta", C.c_float),
                ("epsilon_scaling", C.c_float), ("clip_noise", C.c_float), ("init_noise_sigma", C.c_float),
                ("prediction_type", C.c_int32), ("clip_sample", C.c_int32), ("clip_sample_range", C.c_float),
                ("thresholding", C.c_int32), ("dynamic_thresholding_ratio", C.c_float), ("sample_max_value", C.c_float)]


class ProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("bytes", C.c_double), ("flops", C.c_double)]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C chimera-llm_amd/csrc`). The lcm package has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.lcmhip_last_error.restype = C.c_char_p
    lib.lcmhip_last_error.argtypes = []
    lib.lcmhip_version.restype = C.c_int
    lib.lcmhip_version.argtypes = []
    if lib.lcmhip_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} implements lcmhip ABI {lib.lcmhip_version()}, this package binds ABI {ABI_VERSION}: rebuild it")
    sig = {
        "lcmhip_model_create": [C.POINTER(ModelDesc), C.POINTER(vp)],
        "lcmhip_model_bind_weight": [vp, C.c_char_p, vp, i32, i32, C.POINTER(i64)],
        "lcmhip_model_finalize": [vp],
        "lcmhip_model_destroy": [vp],
        "lcmhip_session_workspace_bytes": [vp, i32, i32, i32, C.c_uint32, C.POINTER(C.c_size_t)],
        "lcmhip_session_create": [vp, i32, i32, i32, C.c_uint32, vp, C.c_size_t, C.POINTER(vp)],
        "lcmhip_session_destroy": [vp],
        "lcmhip_session_set_schedule": [vp, C.POINTER(i32), i32, C.POINTER(f32), i32, f32, i32,
                                        C.POINTER(DenoiseParams), vp],
        "lcmhip_session_reset": [vp, i32, vp],
        "lcmhip_context_append": [vp, vp, i32, i32, vp],
        "lcmhip_base_forward": [vp, vp, i32, i32, vp, f32, i32, vp, vp],
        "lcmhip_denoise": [vp, vp, vp, vp, vp],
        "lcmhip_denoise_stochastic": [vp, vp, vp, vp, vp, vp],
        "lcmhip_denoise_positions": [vp, vp, vp, i32, vp, vp, vp, vp],
        "lcmhip_session_reorder": [vp, C.POINTER(i32), i32, vp],
        "lcmhip_session_context_len": [vp],
        "lcmhip_session_num_active": [vp],
        "lcmhip_session_read_context": [vp, i32, vp, vp],
        "lcmhip_session_profile_enable": [vp, i32],
        "lcmhip_session_profile": [vp, C.POINTER(ProfileRow), i32, C.POINTER(i32), vp],
        "lcmhip_normalize": [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
        "lcmhip_denormalize": [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
        "lcmhip_rfft_pack": [vp, vp, i64, i32, i32, vp],
        "lcmhip_norm_modulate": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
        "lcmhip_gemm": [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, C.c_size_t, vp],
        "lcmhip_static_init": [],
        "lcmhip_gemm_static": [vp, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
        "lcmhip_gemm_partial": [vp, i32, vp, vp, C.c_size_t, C.POINTER(i32), i32, i32, i32, i32, vp],
        "lcmhip_residual_norm": [vp, vp, i32, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
        "lcmhip_ddim_cfg_step": [vp, vp, vp, vp, vp, i32, i32, f32, f32, C.POINTER(DenoiseParams), vp, vp],
        "lcmhip_attention": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
        "lcmhip_stream_copy": [vp, vp, i64, i32, vp],
        "lcmhip_finalize_step": [vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    }
    for name, argtypes in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    return lib


lib = _load()


def check(rc: int) -> None:
    if rc != 0:
        msg = lib.lcmhip_last_error()
        raise RuntimeError(f"lcmhip error {rc}: {msg.decode() if msg else '?'}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise ValueError(f"the HIP path supports float32 and bfloat16 weights, got {dtype}")


def ptr(t) -> C.c_void_p:
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "lcmhip needs contiguous tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def rfft_pack(x: torch.Tensor, inverse: bool) -> torch.Tensor:
    """The packed real FFT a with_fft normalizer applies to raw embeddings (and its inverse), over the last dimension,
    in fp32 (reference sonar_normalizer/builder.py:58-84 computes it in fp32 for half-precision inputs too)."""
    require_gpu(x, "embeddings")
    x32 = x.float().contiguous()
    y = torch.empty_like(x32)
    dim = x32.shape[-1]
    check(lib.lcmhip_rfft_pack(ptr(x32), ptr(y), x32.numel() // dim, dim, int(inverse), stream_ptr(x.device)))
    return y


def require_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what} must live on a GPU: the lcm package runs only on the HIP path")
