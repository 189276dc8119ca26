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


class LoraTarget(C.Structure):   # include/sd_hip.h SdLoraTarget
    _fields_ = [(n, C.c_void_p) for n in ("w_res", "w_out", "w_grad", "a_shadow", "a_scaled", "b_scaled", "d_a", "d_b")] + \
               [("out_features", C.c_int32), ("in_features", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("lm_head", C.c_void_p), ("final_norm", C.c_void_p),
                ("layers_host", C.POINTER(Layer))]


SUMSQ_PARTIALS = 2048  # include/sd_hip.h SD_SUMSQ_PARTIALS

KINDS = ["gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd_dkv", "attn_bwd_dq", "loss_fwd", "loss_bwd", "topk",
         "rmsnorm", "qknorm_rope", "swiglu", "embedding", "optim", "misc", "gemm_nt_stag"]

STAGE_CB = C.CFUNCTYPE(None, C.c_int, C.c_void_p)

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
# name -> (restype, argtypes); must list every symbol include/sd_hip.h declares (tests check this)
_cp = C.c_char_p
# include/sd_hip_debug.h: measurement / test switches (not part of the product ABI)
DEBUG_PROTOTYPES = {
    "sd_debug_set": (_i, [_cp, _i64]),
    "sd_debug_get": (_i64, [_cp]),
    "sd_debug_keys": (_i, [_vp, _i]),
}
PROTOTYPES = {
    "sd_abi_version": (_i, []),
    "sd_gemm_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp]),
    "sd_gemm_splitk_plan": (_i, [_i, _i, _i]),
    "sd_gemm_splitk_workspace_bytes": (_i64, [_i, _i, _i]),
    "sd_gemm_bf16_splitk": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _i, _i, _vp, _i64, _vp]),
    "sd_rmsnorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "sd_rmsnorm_bwd_workspace_bytes": (_i64, [_i, _i]),
    "sd_rmsnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "sd_gemm_swiglu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_gemm_qkv_rope": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sd_embedding_fwd_ssq": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "sd_gemm_bf16_ssq": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64, _i64, _i64, _vp]),
    "sd_gemm_qkv_rope_rs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp]),
    "sd_gemm_swiglu_rs": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp]),
    "sd_qwen3_fold_supported": (_i, [C.POINTER(Dims)]),
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
    "sd_attn_bwd": (_i, [_vp] * 11 + [_i64] * 7 + [_i, _i, _i, _i, _i, _f, _vp]),
    "sd_attn_bwd2": (_i, [_vp] * 11 + [_i64] * 7 + [_i, _i, _i, _i, _i, _f, _vp, _vp]),
    "sd_logsoftmax_topk": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp]),
    "sd_kdloss_stats_bytes": (_i64, [_i, _i]),
    "sd_kdloss_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_kdloss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "sd_gemm_grouped_tn": (_i, [_vp, _i, _i, _i, _vp]),
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
    "sd_lora_plan_bytes": (_i64, [_i]),
    "sd_lora_plan_build": (_i, [C.POINTER(LoraTarget), _i, _i, _vp, _i64]),
    "sd_lora_merge": (_i, [_vp, _vp, _vp]),
    "sd_lora_project": (_i, [_vp, _vp, _vp]),
    "sd_adamw_f32_shadow": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _vp]),
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
    for name, (res, args) in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _LIB = lib
    # The HOST reads the environment, the library never does: SD_GEMM_CU_BUDGET caps the backward's persistent launches of a
    # multi-GPU run; the others are the A/B switches of tests/bench_*.py; SD_DEBUG="key=v,key=v" sets
    # any key of include/sd_hip_debug.h.
    for env, key in _ENV_KEYS.items():
        if os.environ.get(env):
            debug_set(key, int(os.environ[env]))
    for item in filter(None, os.environ.get("SD_DEBUG", "").split(",")):
        k, v = item.split("=")
        debug_set(k.strip(), int(v))
    return lib


_ENV_KEYS = {"SD_GEMM_CU_BUDGET": "gemm.cu_budget", "SD_GEMM_NO_P256": "gemm.no_p256", "SD_GEMM_NO_PERSIST": "gemm.no_persist",
             "SD_GEMM_P256_MIN_TILES": "gemm.p256_min_tiles", "SD_GEMM_GROUP_M": "gemm.group_m",
             "SD_TN_STAG_MIN": "gemm.tn_stag_min", "SD_SPLITK_MIN_KT": "gemm.splitk_min_kt",
             "SD_SPLITK_MIN_SLICE": "gemm.splitk_min_slice", "SD_FUSE_STUDENT_SWIGLU": "model.fuse_student_swiglu",
             "SD_OVERLAP_MASK": "model.overlap_mask", "SD_TOPK_NT": "topk.nt", "SD_QK_BWD_BLOCKS": "qk_bwd.blocks"}


def debug_set(key: str, value: int):
    """include/sd_hip_debug.h: sd_debug_set(key, value); raises on an unknown key."""
    rc = load_lib().sd_debug_set(key.encode(), int(value))
    if rc != 0:
        raise SdHipError(f"sd_debug_set({key!r}) failed: {ERRORS.get(rc, rc)}")


def debug_get(key: str) -> int:
    return int(load_lib().sd_debug_get(key.encode()))


def gemm_force_variant(bm: int, nst: int):
    """tests / benchmarks: force tile rows and ring depth of every later GEMM (0, 0 = heuristic).  nst | 0x100: checked
    (pointer) staging; nst | 0x400: gemm_p256_kernel with 32-deep half-line stages."""
    debug_set("gemm.checked_staging", 1 if nst & 0x100 else 0)
    debug_set("gemm.p256_unpaired", 1 if nst & 0x400 else 0)
    debug_set("gemm.force_bm", bm)
    debug_set("gemm.force_nst", (nst & 0xff) if bm else 0)


def check(rc: int, what: str):
    if rc == 0:
        return
    if rc == -4:
        raise ValueError("Either teacher_logits or top_k must be provided")  # distillation_loss.py:120
    raise SdHipError(f"{what} failed: {ERRORS.get(rc, 'hipError_t ' + str(rc))}")
