"""ctypes binding of libsd_hip.so (the C ABI declared in include/sd_hip.h).

The library is built in-tree by ``make -C speech_distill_amd/csrc`` (or ``__graft_entry__.build()``).
Loading fails loudly: there is no Python/CPU fallback behind any of these symbols.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERRORS = {-1: "SD_ERR_SHAPE", -2: "SD_ERR_ALIGN", -3: "SD_ERR_UNSUPPORTED", -4: "SD_ERR_NO_TEACHER",
          -5: "SD_ERR_WORKSPACE"}


class SdHipError(RuntimeError):
    pass


def lib_path() -> str:
    # SD_HIP_LIB: a diagnostic build of the same library (make -C speech_distill_amd/csrc stamps), measurements only
    return os.environ.get("SD_HIP_LIB") or os.path.join(_HERE, "libsd_hip.so")


class Dims(C.Structure):
    _fields_ = [("vocab", C.c_int32), ("hidden", C.c_int32), ("inter", C.c_int32), ("layers", C.c_int32),
                ("n_q", C.c_int32), ("n_kv", C.c_int32), ("head_dim", C.c_int32), ("tied", C.c_int32),
                ("eps", C.c_float), ("pad_", C.c_int32)]


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("lda", C.c_int64), ("ldb", C.c_int64),
                ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32)]


class GemmNtProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("out2", C.c_void_p), ("slabs", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32),
                ("K", C.c_int32), ("nsplit", C.c_int32)]


class ColsumProblem(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("out", C.c_void_p), ("nb", C.c_int32), ("H", C.c_int32),
                ("stride", C.c_int32), ("accumulate", C.c_int32)]


class Layer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wqkv", "wo", "wgu", "wdown", "q_gain", "k_gain", "ln1", "ln2")]


class Params(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("lm_head", C.c_void_p), ("final_norm", C.c_void_p),
                ("layers_host", C.POINTER(Layer))]


SUMSQ_PARTIALS = 2048  # include/sd_hip.h SD_SUMSQ_PARTIALS

KINDS = ["gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd_dkv", "attn_bwd_dq", "loss_fwd", "loss_bwd", "topk",
         "rmsnorm", "qknorm_rope", "swiglu", "embedding", "optim", "misc", "gemm_nt_stag"]

STAGE_CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
# name -> (restype, argtypes); must list every symbol include/sd_hip.h declares (tests check this)
PROTOTYPES = {
    "sd_abi_version": (_i, []),
    "sd_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp]),
    "sd_gemm_force_variant": (None, [_i, _i]),
    "sd_gemm_splitk_plan": (_i, [_i, _i, _i]),
    "sd_gemm_splitk_workspace_bytes": (_i64, [_i, _i, _i]),
    "sd_gemm_bf16_splitk": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp, _i64, _vp]),
    "sd_rmsnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sd_rmsnorm_fwd_slabs": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sd_rmsnorm_bwd_workspace_bytes": (_i64, [_i, _i]),
    "sd_rmsnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "sd_gemm_swiglu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_gemm_qkv_rope": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sd_gemm_bf16_splitk_partial": (_i, [_vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i, _i, _vp, _i64, _vp, _vp]),
    "sd_rmsnorm_bwd_slabs": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "sd_rmsnorm_bwd2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "sd_qknorm_rope_bwd2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "sd_qknorm_rope_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "sd_qknorm_rope_bwd_workspace_bytes": (_i64, [_i, _i, _i]),
    "sd_qknorm_rope_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp]),
    "sd_swiglu_fwd": (_i, [_vp, _vp, _i, _i, _vp]),
    "sd_swiglu_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "sd_embedding_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_embedding_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "sd_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i, _i, _i, _i, _i, _f, _vp]),
    "sd_attn_force_variant": (None, [_i]),
    "sd_attn_bwd": (_i, [_vp] * 11 + [_i64] * 7 + [_i, _i, _i, _i, _i, _f, _vp]),
    "sd_attn_bwd2": (_i, [_vp] * 11 + [_i64] * 7 + [_i, _i, _i, _i, _i, _f, _vp, _vp]),
    "sd_logsoftmax_topk": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "sd_kdloss_stats_bytes": (_i64, [_i, _i]),
    "sd_kdloss_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_kdloss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_gemm_grouped_tn": (_i, [_vp, _i, _i, _i, _vp]),
    "sd_gemm_grouped_nt": (_i, [_vp, _i, _i, _vp]),
    "sd_debug_cu_budget": (None, [_i]),
    "sd_gemm_swiglu_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_gemm_odx_delta": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _vp]),
    "sd_kdloss_fwd_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_kdloss_bwd_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_rows_scatter": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_loss_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sd_rmsnorm_bwd_partial_rows": (_i, [_i, _i]),
    "sd_qknorm_rope_bwd_partial_rows": (_i, [_i, _i, _i]),
    "sd_colsum_reduce_batch": (_i, [_vp, _i, _vp]),
    "sd_sumsq_bf16": (_i, [_vp, _i64, _vp, _vp, _vp]),
    "sd_adamw_bf16": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _vp]),
    "sd_streams_overlap": (_i, [_vp, _vp, _f, C.POINTER(C.c_int)]),
    "sd_prof_begin": (_i, []),
    "sd_prof_end": (_i, [_vp, _vp, _vp, _i]),
    "sd_prof_symbols": (_i64, [_vp, _i64]),
    "sd_qwen3_acts_bytes": (_i64, [C.POINTER(Dims), _i, _i, _i]),
    "sd_qwen3_bwd_scratch_bytes": (_i64, [C.POINTER(Dims), _i, _i]),
    "sd_qwen3_forward": (_i, [C.POINTER(Dims), C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp]),
    "sd_qwen3_forward_rows": (_i, [C.POINTER(Dims), C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i, _i, _i,
                                   _i, _vp]),
    "sd_qwen3_backward_rows": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _i64,
                                    _vp, _vp, _i, _vp, _i64, _i, _i, _i, _vp, STAGE_CB, _vp, _vp, _vp]),
    "sd_qwen3_backward": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Params), _vp, _vp, _vp, _vp, _vp, _i64,
                               _vp, _vp, _i64, _i, _i, _i, _vp, STAGE_CB, _vp, _vp, _vp]),
}


def load_lib():
    """Load libsd_hip.so once; raise SdHipError (never fall back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise SdHipError(
            f"{path} not found: build it with `make -C speech_distill_amd/csrc` (hipcc --offload-arch=gfx950). "
            "speech_distill_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -4:
        raise ValueError("Either teacher_logits or top_k must be provided")  # distillation_loss.py:120
    raise SdHipError(f"{what} failed: {ERRORS.get(rc, 'hipError_t ' + str(rc))}")
