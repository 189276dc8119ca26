"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/sd_hip.h declares, and the ctypes table binds exactly that set (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "sd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", txt))
    names.discard("sd_stage_cb")
    return names


def test_header_symbols_are_exported_and_bound():
    import speech_distill_amd as sda
    from speech_distill_amd import _lib
    if not os.path.exists(sda.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(sda.lib_path())
    declared = _declared()
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/sd_hip.h but not exported by libsd_hip.so"
    assert set(_lib.PROTOTYPES) == declared, (set(_lib.PROTOTYPES) ^ declared)
    assert sda.load_lib().sd_abi_version() == 1


def test_workspace_queries_run_without_gpu():
    import speech_distill_amd as sda
    from speech_distill_amd import _lib
    lib = sda.load_lib()
    d = _lib.Dims(159488, 1024, 3072, 28, 16, 8, 128, 1, 1e-6, 0)
    train = lib.sd_qwen3_acts_bytes(ctypes.byref(d), 4, 512, 1)
    infer = lib.sd_qwen3_acts_bytes(ctypes.byref(d), 4, 512, 0)
    assert 2e9 < train < 4e9 and infer < train / 10
    # SD_SAVE_LAYER_INPUTS (gradient checkpointing): 28 layer inputs + two layer work sets instead of 28 sets
    ckpt = lib.sd_qwen3_acts_bytes(ctypes.byref(d), 4, 512, 2)
    assert infer < ckpt < train / 8
    assert lib.sd_qwen3_acts_bytes(ctypes.byref(d), 4, 512, 4) < 0  # unknown mode: SD_ERR_SHAPE, not a size
    folded = lib.sd_qwen3_acts_bytes(ctypes.byref(d), 4, 512, 3)  # SD_SAVE_NONE_FOLDED: the inference set + two ssq sets
    assert infer < folded <= infer + 2 * (4 * 512 * 64 + 256)
    assert lib.sd_kdloss_stats_bytes(4, 512) == 4 * 512 * 32
    assert lib.sd_gemm_splitk_plan(2048, 1024, 159488) > 1 and lib.sd_gemm_splitk_plan(2048, 6144, 1024) == 1


def test_cpu_tensors_are_rejected_not_routed_to_a_fallback():
    import torch
    import speech_distill_amd as sda
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sda.DistillationLoss()(torch.randn(1, 4, 64), torch.tensor([[1, 2, 3, 4]]), teacher_logits=torch.randn(1, 4, 64))
    with pytest.raises(ValueError, match="Either teacher_logits or top_k must be provided"):
        sda.DistillationLoss()(torch.randn(1, 4, 64), torch.tensor([[1, 2, 3, 4]]))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, smoke() and bench.py's cpu_baseline may touch it."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "speech_distill_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cuh")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench.split("def cpu_baseline", 1)[1].split("\ndef ", 1)
    outside = bench.split("def cpu_baseline", 1)[0] + body[1]
    # nothing outside cpu_baseline imports it (the FLOP counts live in speech_distill_amd.qwen3.Qwen3Dims)
    assert not [m.group(0) for m in re.finditer(r".*oracle.*", outside) if re.search(r"\b(import|from)\b", m.group(0))]


def test_flop_counts_of_the_product_equal_the_survey_figures():
    """student fwd 1.266 G, teacher fwd 3.531 G, step 7.33 GFLOP/token at T=512; the oracle's own
    count agrees (two independent statements of the same table)."""
    from oracle import qwen3 as Q
    from speech_distill_amd.qwen3 import Qwen3Dims
    s, t = Qwen3Dims.student_06b(), Qwen3Dims.teacher_17b()
    assert abs(s.flops_per_token(512) / 1e9 - 1.266) < 2e-3 and abs(t.flops_per_token(512) / 1e9 - 3.531) < 2e-3
    assert abs((3 * s.flops_per_token(512) + t.flops_per_token(512)) / 1e9 - 7.33) < 5e-3
    assert s.matmul_params() == 603_717_632 + 0 or abs(s.matmul_params() / 1e6 - 603.7) < 0.1
    for d, o in ((s, Q.STUDENT_06B), (t, Q.TEACHER_17B)):
        for T in (512, 2048):
            assert d.flops_per_token(T) == Q.flops_per_token(o, T)


def test_debug_interface_is_separate_documented_and_complete():
    """include/sd_hip_debug.h: the measurement switches are NOT declared in sd_hip.h, the library never
    reads the environment (no getenv in csrc/), every key the library knows is documented in the debug header, unknown
    keys are refused, `reset` restores the defaults, and the split-K slice floor cannot be set to a divisor of zero."""
    import speech_distill_amd as sda
    from speech_distill_amd import _lib
    prod = open(os.path.join(ROOT, "include", "sd_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", prod, flags=re.S)
    assert "sd_debug" not in code and "force_variant" not in code and "cu_budget" not in code
    for f in os.listdir(os.path.join(ROOT, "speech_distill_amd", "csrc")):
        if f.endswith((".hip", ".h", ".cuh")):
            assert "getenv" not in open(os.path.join(ROOT, "speech_distill_amd", "csrc", f)).read(), f
    dbg = open(os.path.join(ROOT, "include", "sd_hip_debug.h")).read()
    declared = set(re.findall(r"\b(sd_debug_[a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", dbg, flags=re.S)))
    assert declared == set(_lib.DEBUG_PROTOTYPES) == {"sd_debug_set", "sd_debug_get", "sd_debug_keys"}
    lib = sda.load_lib()
    n = lib.sd_debug_keys(None, 0)
    buf = ctypes.create_string_buffer(n)
    assert lib.sd_debug_keys(buf, n) == n
    keys = buf.value.decode().split()
    assert len(keys) >= 15
    for k in keys:
        assert re.search(r"^ \*\s+(\S+ / )?" + re.escape(k) + r"\b|" + re.escape(k.split(".")[0] + ".") + r"\S* / " +
                         re.escape(k.split(".")[1]), dbg, re.M), f"{k} is not documented in sd_hip_debug.h"
    assert lib.sd_debug_set(b"no.such.key", 1) == -3 and lib.sd_debug_get(b"no.such.key") == -2 ** 63
    try:
        _lib.debug_set("gemm.splitk_min_slice", 0)
        assert _lib.debug_get("gemm.splitk_min_slice") == 1
        assert lib.sd_gemm_splitk_plan(2048, 1024, 159488) >= 1
        _lib.debug_set("gemm.cu_budget", 5)
        assert _lib.debug_get("gemm.cu_budget") == 5
    finally:
        _lib.debug_set("reset", 0)
    assert _lib.debug_get("gemm.cu_budget") == 0 and _lib.debug_get("gemm.splitk_min_slice") == 24


def test_library_holds_no_undispatched_gemm_kernels():
    """The experiment kernels of round 3 (gemm_p1, gemm_pgroup_nt, gemm_ks, rmsnorm_fwd_slabs) are out of the product."""
    import speech_distill_amd as sda
    blob = open(sda.lib_path(), "rb").read()
    for dead in (b"gemm_p1_kernel", b"gemm_pgroup_nt_kernel", b"gemm_ks_kernel", b"rmsnorm_fwd_slabs_kernel"):
        assert dead not in blob, dead


def test_gemm_table_covers_every_baseline_shape_and_is_in_sync():
    """speech_distill_amd/csrc/sd_gemm_table.inc (compiled into the launcher) holds one entry per GEMM call
    of the distillation step at the BASELINE config 2 / 4 / 5 shapes -- the same enumeration tests/bench_tune.py measures
    -- and is exactly what scripts/make_gemm_table.py generates from the committed profile."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    calls = importlib.import_module("bench_tune").calls
    inc = open(os.path.join(ROOT, "speech_distill_amd", "csrc", "sd_gemm_table.inc")).read()
    have = set()
    for m in re.finditer(r"^\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\}", inc, re.M):
        ta, tb, epi, M, N, K, bm, nst, fl = (int(x) for x in m.groups())
        assert bm in (0, 64, 128, 256) and (bm == 0) == (nst == 0)
        have.add((ta, tb, epi, M, N, K))
    for cfg in ("c2", "c4", "c5"):
        for name, form, epi, M, N, K in calls(cfg):
            key = (int(form == "TN"), int(form in ("NN", "TN")), epi, M, N, K)
            assert key in have, f"{name} {key} has no entry in sd_gemm_table.inc"
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_gemm_table.py"),
                          os.path.join("profiles", "r04_gemm_tune.json")], cwd=ROOT, capture_output=True, text=True, check=True)
    assert gen.stdout == inc, "sd_gemm_table.inc is stale: regenerate it with scripts/make_gemm_table.py"
