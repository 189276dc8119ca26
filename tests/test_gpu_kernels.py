"""-m gpu parity tests: every HIP launcher (through the C ABI / ctypes) against the CPU oracle, the
golden fixtures generated from the reference, and a plain fp32/fp64 torch statement of the op.

Tolerances (stated per test): bf16 kernels are compared with an fp64 evaluation of the SAME bf16
inputs; one bf16 rounding of the output is 2^-9 relative (3.9e-3 max), so max-norm bounds are a few
times that and rms bounds ~2e-3.  Integer-valued inputs must come out exact.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import distill_loss, qwen3
    return type("O", (), {"L": distill_loss, "Q": qwen3})


@pytest.fixture(scope="module")
def ops():
    import speech_distill_amd.ops as ops_
    ops_.load_lib()
    # the 256x256 persistent GEMM serves only vocabulary-wide GEMMs by default; the tests reach it at small sizes
    _lib.debug_set("gemm.p256_min_tiles", 150)
    yield ops_
    _lib.debug_set("reset", 0)


from gpu_util import bf, check_close, dev, record, to_dev  # noqa: E402
from speech_distill_amd import _lib  # noqa: E402


# ------------------------------------------------------------------------------------------- GEMM
def _gemm_ref(a, b, ta, tb, r=None):
    A = a.double().T if ta else a.double()
    Bm = b.double() if tb else b.double().T
    c = A @ Bm
    if r is not None:
        c = c + r.double()
    return c


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_exact_integers(ops, ta, tb):
    """Small integers are exact in bf16 and fp32: any layout / fragment-order mistake shows as a wrong integer."""
    g = torch.Generator().manual_seed(1)
    M, N, K = 200, 136, 72
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    a[0, :] += 1.0  # asymmetric
    got = ops.gemm(to_dev(bf(a)), to_dev(bf(b)), ta, tb).float().cpu()
    ref = _gemm_ref(a, b, ta, tb).float().bfloat16().float()
    bad = (got != ref).nonzero()
    record(f"gemm_exact_ta{int(ta)}_tb{int(tb)}", n_bad=int(bad.shape[0]))
    assert bad.shape[0] == 0, f"{bad.shape[0]} wrong entries, first {bad[:5].tolist()} got {got[tuple(bad[0])]} ref {ref[tuple(bad[0])]}"


@pytest.mark.parametrize("M,N,K,ta,tb,res", [
    (128, 128, 64, False, False, False), (2048, 1024, 1024, False, False, True), (300, 520, 128, False, False, False),
    (2048, 1024, 4096, False, True, False), (333, 256, 520, False, True, True),
    (4096, 1024, 2048, True, True, False), (520, 128, 48, True, True, True), (1000, 640, 160, True, True, False),
])
def test_gemm_random(ops, M, N, K, ta, tb, res):
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn((K, M) if ta else (M, K), generator=g))
    b = bf(torch.randn((K, N) if tb else (N, K), generator=g))
    r = bf(torch.randn(M, N, generator=g) * 8) if res else None
    got = ops.gemm(to_dev(a), to_dev(b), ta, tb, residual=None if r is None else to_dev(r))
    ref = _gemm_ref(a.float(), b.float(), ta, tb, None if r is None else r.float())
    check_close(f"gemm_{M}x{N}x{K}_ta{int(ta)}tb{int(tb)}r{int(res)}", got, ref, 6e-3, 3e-3)


@pytest.mark.parametrize("bm,nst", [(256, 9), (64, 9), (256, 3), (256, 2), (128, 3), (64, 4), (64, 2), (128, 2 | 0x100),
                                    (64, 3 | 0x100)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_forced_variants_exact(ops, bm, nst, ta, tb):
    """every tile height / ring depth the heuristic can pick, on integer data (must be exact), ragged edges."""
    g = torch.Generator().manual_seed(bm + nst)
    M, N, K = 520, 264, 200
    a = torch.randint(-3, 4, (K, M) if ta else (M, K), generator=g).float()
    b = torch.randint(-3, 4, (K, N) if tb else (N, K), generator=g).float()
    lib = ops.load_lib()
    _lib.gemm_force_variant(bm, nst)
    try:
        got = ops.gemm(to_dev(bf(a)), to_dev(bf(b)), ta, tb).float().cpu()
    finally:
        _lib.gemm_force_variant(0, 0)
    # integer sums are exact in fp32; the bf16 output rounds values above 256 (RNE), so round the reference too
    assert torch.equal(got, _gemm_ref(a, b, ta, tb).float().bfloat16().float())


@pytest.mark.parametrize("bm,K", [(256, 2112), (256, 64), (256, 192)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_staggered_kernel_random(ops, ta, tb, bm, K):
    """the two-half staggered 256x128 kernel (forced), with a residual: long K so the 3-stage ring wraps many times, and
    fewer K tiles than ring stages; run 3x (race screen)."""
    g = torch.Generator().manual_seed(77)
    M, N = 1000, 520
    a = bf(torch.randn((K, M) if ta else (M, K), generator=g))
    b = bf(torch.randn((K, N) if tb else (N, K), generator=g))
    r = bf(torch.randn(M, N, generator=g) * 8)
    ref = _gemm_ref(a.float(), b.float(), ta, tb, r.float())
    lib = ops.load_lib()
    ad, bd, rd = to_dev(a), to_dev(b), to_dev(r)
    outs = []
    for _ in range(3):
        _lib.gemm_force_variant(bm, 9)
        try:
            outs.append(ops.gemm(ad, bd, ta, tb, residual=rd))
        finally:
            _lib.gemm_force_variant(0, 0)
    check_close(f"gemm_stag{bm}_K{K}_ta{int(ta)}tb{int(tb)}", outs[0], ref, 6e-3, 3e-3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("bm,nst", [(0, 0), (64, 4), (128, 3), (256, 9)])
def test_gemm_fused_epilogues_equal_the_separate_kernels(ops, bm, nst):
    """q|k|v GEMM + q/k-norm + RoPE, and gate|up GEMM + SwiGLU, fused into the GEMM epilogue: bit-identical to the
    separate launchers (same bf16 roundings), for every kernel variant that can run them; ragged M."""
    g = torch.Generator().manual_seed(31)
    B, T, Hq, Hkv, K, I = 2, 75, 4, 2, 256, 192
    M = B * T
    x = to_dev(bf(torch.randn(M, K, generator=g)))
    wqkv = to_dev(bf(torch.randn((Hq + 2 * Hkv) * 128, K, generator=g) * 0.1))
    wgu = to_dev(bf(torch.randn(2 * I, K, generator=g) * 0.1))
    qg, kg = to_dev(bf(1 + 0.2 * torch.randn(128, generator=g))), to_dev(bf(1 + 0.2 * torch.randn(128, generator=g)))
    cos, sin = ops.rope_tables(T, dev())
    lib = ops.load_lib()
    qkv_ref = ops.gemm(x, wqkv)
    qk_ref = ops.qknorm_rope_fwd(qkv_ref, qg, kg, cos, sin, T, Hq, Hkv)
    gu_ref = ops.gemm(x, wgu)
    act_ref = ops.swiglu_fwd(gu_ref)
    _lib.gemm_force_variant(bm, nst)
    try:
        qkv, qk = ops.gemm_qkv_rope(x, wqkv, qg, kg, cos, sin, T, Hq, Hkv)
        act, gu = ops.gemm_swiglu(x, wgu)
        act2, none = ops.gemm_swiglu(x, wgu, save_gu=False)
    finally:
        _lib.gemm_force_variant(0, 0)
    assert torch.equal(qkv, qkv_ref) and torch.equal(gu, gu_ref) and none is None
    assert torch.equal(qk, qk_ref), float((qk.float() - qk_ref.float()).abs().max())
    assert torch.equal(act, act_ref) and torch.equal(act2, act_ref), float((act.float() - act_ref.float()).abs().max())


def test_gemm_table_entries_change_the_kernel_not_the_result(ops):
    """Dispatch by measurement (csrc/sd_gemm_table.inc): for shapes the table overrides -- config 4's gate|up dX (split-K
    plan, 256x3 instead of the staggered kernel) and student gate|up forward (the 256x256 kernel below its usual tile
    threshold) -- the table picks ANOTHER kernel than the heuristic ("gemm.no_table" = 1) and the result is the same bits."""
    g = torch.Generator().manual_seed(44)
    cases = [((8192, 6144), (6144, 1024), False, True, True),     # dY [M, 2I] . Wgu [2I, h] -> dX [M, h]
             ((8192, 1024), (6144, 1024), False, False, False)]   # x [M, h] . Wgu [2I, h]^T
    for ashape, bshape, ta, tb, sk in cases:
        a = to_dev(bf(torch.randn(*ashape, generator=g)))
        b = to_dev(bf(torch.randn(*bshape, generator=g) * 0.05))
        outs, syms = [], []
        for no_table in (0, 1):
            _lib.debug_set("gemm.no_table", no_table)
            _lib.debug_set("gemm.p256_min_tiles", 1024)  # the product default (this module lowers it for other tests)
            try:
                ops.prof_begin()
                outs.append(ops.gemm(a, b, ta, tb, split_k=sk))
                ops.prof_end()
                syms.append(sorted(k for k in ops.prof_symbols() if k.startswith("gemm_")))
            finally:
                _lib.debug_set("gemm.no_table", 0)
                _lib.debug_set("gemm.p256_min_tiles", 150)
        assert syms[0] != syms[1], syms
        assert torch.equal(outs[0], outs[1]), syms


@pytest.mark.parametrize("H", [1024, 2048])
def test_folded_rmsnorm_pieces(ops, O, H):
    """RMSNorm folded into the projection behind it (include/sd_hip.h; the frozen teacher's inference forward): the
    producers of the row statistic (embedding lookup, residual GEMM epilogue) leave per-128-column-tile sums of squares,
    the consumers (q|k|v + q/k-norm + RoPE, gate|up + SwiGLU) turn them into rstd and scale the accumulator row.
    * producers: the bf16 outputs are bit-identical to the plain launchers; the partials equal the fp64 sums of squares
      of those bf16 values to fp32 rounding, and are bit-identical across kernel variants;
    * consumers: every kernel variant -- the one-tile kernels (rstd via a DPP sum in the epilogue preload) AND the
      persistent 256x128 kernel (rstd via the producer waves' LDS table) -- gives the same bits; against an fp64
      evaluation of  rstd * (x W'^T)  followed by the unfused epilogue arithmetic: one bf16 rounding."""
    Q = O.Q
    g = torch.Generator().manual_seed(H)
    eps = 1e-6
    V, M, T, Hq, Hkv, I, Kd = 3000, 2 * 150, 150, 4, 2, 64 * 3, 384
    nt = H // 128
    # ---- producers
    E = to_dev(bf(torch.randn(V, H, generator=g)))
    ids = to_dev(torch.randint(0, V, (M,), generator=g))
    x0, ssq0 = ops.embedding_fwd_ssq(ids, E)
    assert torch.equal(x0, ops.embedding_fwd(ids, E))
    want0 = x0.double().view(M, nt, 128).pow(2).sum(-1).t()  # tile-major [H/128, M]
    np.testing.assert_allclose(ssq0.double().cpu().numpy(), want0.cpu().numpy(), rtol=1e-5)
    a = to_dev(bf(torch.randn(M, Kd, generator=g)))
    w = to_dev(bf(torch.randn(H, Kd, generator=g) * 0.05))
    c_ref = ops.gemm(a, w, residual=x0)
    outs = []
    for bm, nst in ((0, 0), (64, 4), (128, 3), (256, 9)):
        _lib.gemm_force_variant(bm, nst)
        try:
            outs.append(ops.gemm_resid_ssq(a, w, x0))
        finally:
            _lib.gemm_force_variant(0, 0)
    x1, ssq1 = outs[0]
    for c, sq in outs:
        assert torch.equal(c, c_ref) and torch.equal(sq, ssq1)
    np.testing.assert_allclose(ssq1.double().cpu().numpy(), x1.double().view(M, nt, 128).pow(2).sum(-1).t().cpu().numpy(), rtol=1e-5)
    # ---- consumers: weights with the norm's gain folded in
    gain = bf(1 + 0.3 * torch.randn(H, generator=g))
    wqkv = bf(torch.randn((Hq + 2 * Hkv) * 128, H, generator=g) * 0.05)
    wgu = bf(torch.randn(2 * I, H, generator=g) * 0.05)
    wqkv_f, wgu_f = bf(wqkv.float() * gain.float()[None]), bf(wgu.float() * gain.float()[None])
    qg, kg = bf(1 + 0.2 * torch.randn(128, generator=g)), bf(1 + 0.2 * torch.randn(128, generator=g))
    cos, sin = ops.rope_tables(T, dev())
    res_q, res_s = [], []
    for bm, nst in ((0, 0), (64, 4), (128, 3), (256, 9)):
        _lib.gemm_force_variant(bm, nst)
        try:
            res_q.append(ops.gemm_qkv_rope_rs(x1, to_dev(wqkv_f), to_dev(qg), to_dev(kg), cos, sin, ssq1, T, Hq, Hkv, eps))
            res_s.append(ops.gemm_swiglu_rs(x1, to_dev(wgu_f), ssq1, eps, save_gu=True))
        finally:
            _lib.gemm_force_variant(0, 0)
    for (qkv, qk), (act, gu) in zip(res_q, res_s):
        assert torch.equal(qkv, res_q[0][0]) and torch.equal(qk, res_q[0][1])
        assert torch.equal(act, res_s[0][0]) and torch.equal(gu, res_s[0][1])
    # fp64 statement: rstd from the bf16 row, scale the exact product, then the unfused epilogues' roundings
    xd = x1.double().cpu()
    rstd = (xd.pow(2).mean(-1, keepdim=True) + eps).rsqrt()
    raw = bf((rstd * (xd @ wqkv_f.double().T)).float())
    check_close(f"fold_qkv_raw_H{H}", res_q[0][0], raw, 8e-3, 2e-3)
    qh = raw.float().view(2, T, Hq + 2 * Hkv, 128)
    qn = Q.rms_norm(qh[:, :, :Hq].bfloat16(), qg, eps).transpose(1, 2)
    kn = Q.rms_norm(qh[:, :, Hq:Hq + Hkv].bfloat16(), kg, eps).transpose(1, 2)
    cs, sn = cos.cpu().float(), sin.cpu().float()
    qr = Q.apply_rope(qn.float(), cs, sn).transpose(1, 2).reshape(M, Hq * 128)
    kr = Q.apply_rope(kn.float(), cs, sn).transpose(1, 2).reshape(M, Hkv * 128)
    check_close(f"fold_qk_rot_H{H}", res_q[0][1], torch.cat([qr, kr], -1), 2.5e-2, 5e-3)
    gud = bf((rstd * (xd @ wgu_f.double().T)).float())
    check_close(f"fold_gu_H{H}", res_s[0][1], gud, 8e-3, 2e-3)
    gf, uf = res_s[0][1][:, :I].float(), res_s[0][1][:, I:].float()
    assert torch.equal(res_s[0][0], bf(torch.nn.functional.silu(gf) * uf)) or \
        float((res_s[0][0].float() - torch.nn.functional.silu(gf) * uf).abs().max()) <= 4e-3 * float(uf.abs().max())
    # ---- the persistent 256x128 kernel (more tiles than CUs): its producer-wave rstd table against the one-tile kernels
    Mp, Ip = 2100, 64 * 40
    xp = to_dev(bf(torch.randn(Mp, H, generator=g)))
    rp = to_dev(bf(torch.randn(Mp, H, generator=g)))
    xp2, ssqp = ops.gemm_resid_ssq(to_dev(bf(torch.randn(Mp, 128, generator=g))), to_dev(bf(torch.randn(H, 128, generator=g) * 0.1)), rp)
    wp = to_dev(bf(torch.randn(2 * Ip, H, generator=g) * 0.05))
    _lib.debug_set("gemm.no_p256", 1)
    try:
        ops.prof_begin()
        act_p, gu_p = ops.gemm_swiglu_rs(xp2, wp, ssqp, eps, save_gu=True)
        act_p2, _ = ops.gemm_swiglu_rs(xp2, wp, ssqp, eps)
        ops.prof_end()
        assert any(k.startswith("gemm_pstag_kernel<4, false, false, 3>") for k in ops.prof_symbols()), ops.prof_symbols()
        _lib.gemm_force_variant(128, 3)
        act_1, gu_1 = ops.gemm_swiglu_rs(xp2, wp, ssqp, eps, save_gu=True)
    finally:
        _lib.gemm_force_variant(0, 0)
        _lib.debug_set("gemm.no_p256", 0)
    assert torch.equal(gu_p, gu_1) and torch.equal(act_p, act_1) and torch.equal(act_p2, act_1)
    del xp


@pytest.mark.parametrize("K", [64, 128, 192, 1088])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_gemm_persistent_kernel(ops, ta, tb, K):
    """more 256x128 tiles than CUs -> the persistent staggered kernel: one K stream across tile boundaries (K of 1, 2,
    3 and 17 steps per tile), ragged M and N edges, register epilogue; must equal the one-tile-per-workgroup kernel
    bit for bit (same accumulation order); run twice (race screen)."""
    g = torch.Generator().manual_seed(K)
    M, N = 1000, 128 * 83 + 40
    a = bf(torch.randn((K, M) if ta else (M, K), generator=g))
    b = bf(torch.randn((K, N) if tb else (N, K), generator=g))
    ad, bd = to_dev(a), to_dev(b)
    lib = ops.load_lib()
    outs = []
    for _ in range(2):
        _lib.gemm_force_variant(256, 9)
        try:
            outs.append(ops.gemm(ad, bd, ta, tb))
        finally:
            _lib.gemm_force_variant(0, 0)
    _lib.gemm_force_variant(128, 3)
    try:
        other = ops.gemm(ad, bd, ta, tb)
    finally:
        _lib.gemm_force_variant(0, 0)
    check_close(f"gemm_persist_K{K}_ta{int(ta)}tb{int(tb)}", outs[0], _gemm_ref(a.float(), b.float(), ta, tb), 6e-3, 3e-3)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], other)


@pytest.mark.parametrize("K", [64, 192, 1088])
def test_gemm_persistent_256x256_kernel(ops, K):
    """Forward (NT) GEMMs with >= 150 tiles of 256x256 take gemm_p256_kernel (8 waves, 64x128 per wave): K of 2, 6 and 34
    32-deep steps per tile, ragged M and N edges, one K stream across tile boundaries; must equal the 128x128 kernel bit
    for bit (same accumulation order over k); run twice (race screen).  The default is the paired form (two slots of
    64 KiB holding both 32-deep halves of whole 128-byte lines); _lib.gemm_force_variant(0, 0x400) runs the 32-deep half-line
    stages of round 2: both are checked."""
    g = torch.Generator().manual_seed(100 + K)
    M, N = 1000, 256 * 60 + 40
    a = bf(torch.randn(M, K, generator=g))
    b = bf(torch.randn(N, K, generator=g))
    ad, bd = to_dev(a), to_dev(b)
    lib = ops.load_lib()
    ops.prof_begin()
    outs = [ops.gemm(ad, bd), ops.gemm(ad, bd)]
    ops.prof_end()
    assert any(k.startswith("gemm_p256_kernel<0, true>") for k in ops.prof_symbols()), ops.prof_symbols()
    _lib.gemm_force_variant(0, 0x400)
    try:
        ops.prof_begin()
        outs.append(ops.gemm(ad, bd))
        ops.prof_end()
        assert any(k.startswith("gemm_p256_kernel<0, false>") for k in ops.prof_symbols()), ops.prof_symbols()
        _lib.gemm_force_variant(128, 3)
        other = ops.gemm(ad, bd)
    finally:
        _lib.gemm_force_variant(0, 0)
    assert torch.equal(outs[0], outs[2])
    check_close(f"gemm_p256_K{K}", outs[0], _gemm_ref(a.float(), b.float(), False, False), 6e-3, 3e-3)
    assert torch.equal(outs[0], outs[1])
    assert torch.equal(outs[0], other)
    # exact on small integers (any fragment / layout slip shows as a wrong integer)
    ai = torch.randint(-3, 4, (M, K), generator=g).float()
    bi = torch.randint(-3, 4, (N, K), generator=g).float()
    ai[0, :] += 1.0
    got = ops.gemm(to_dev(bf(ai)), to_dev(bf(bi))).float().cpu()
    assert torch.equal(got, (ai.double() @ bi.double().t()).float().bfloat16().float())


def test_gemm_persistent_swiglu(ops):
    """gate|up GEMM + SwiGLU on the persistent 256x256 kernel (tile = 128 gate rows | 128 up rows): bit-identical to
    GEMM + swiglu; and on the 256x128 persistent kernel ("gemm.no_p256" = 1)."""
    g = torch.Generator().manual_seed(5)
    M, K, I = 1100, 192, 64 * 90
    x = to_dev(bf(torch.randn(M, K, generator=g)))
    wgu = to_dev(bf(torch.randn(2 * I, K, generator=g) * 0.1))
    lib = ops.load_lib()
    _lib.gemm_force_variant(128, 3)
    try:
        gu_ref = ops.gemm(x, wgu)
    finally:
        _lib.gemm_force_variant(0, 0)
    act_ref = ops.swiglu_fwd(gu_ref)
    ops.prof_begin()
    act, gu = ops.gemm_swiglu(x, wgu)
    ops.prof_end()
    assert any(k.startswith("gemm_p256_kernel<3, true>") for k in ops.prof_symbols()), ops.prof_symbols()  # 5 x 45 tiles of 256x256
    act2, none = ops.gemm_swiglu(x, wgu, save_gu=False)
    assert torch.equal(gu, gu_ref) and none is None
    assert torch.equal(act, act_ref) and torch.equal(act2, act_ref)
    _lib.debug_set("gemm.no_p256", 1)
    try:
        ops.prof_begin()
        act3, gu3 = ops.gemm_swiglu(x, wgu)
        ops.prof_end()
    finally:
        _lib.debug_set("gemm.no_p256", 0)
    assert any(k.startswith("gemm_pstag_kernel<4, false, false, 3>") for k in ops.prof_symbols()), ops.prof_symbols()
    assert torch.equal(act3, act_ref) and torch.equal(gu3, gu_ref)


@pytest.mark.parametrize("bm,nst", [(0, 0), (64, 3), (128, 2), (128, 3), (256, 9)])
def test_gemm_swiglu_bwd_epilogue_equals_the_separate_kernels(ops, bm, nst):
    """down-projection dX GEMM + SwiGLU backward in its epilogue == sd_gemm_bf16 (NN) + sd_swiglu_bwd, bit for bit."""
    g = torch.Generator().manual_seed(17)
    M, H, I = 300, 256, 520
    dy = to_dev(bf(torch.randn(M, H, generator=g)))
    wdown = to_dev(bf(torch.randn(H, I, generator=g) * 0.1))
    gu = to_dev(bf(torch.randn(M, 2 * I, generator=g)))
    dact = ops.gemm(dy, wdown, False, True)
    ref = ops.swiglu_bwd(dact, gu)
    lib = ops.load_lib()
    _lib.gemm_force_variant(bm, nst)
    try:
        got = ops.gemm_swiglu_bwd(dy, wdown, gu)
    finally:
        _lib.gemm_force_variant(0, 0)
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


@pytest.mark.parametrize("K", [64, 200, 2048])
def test_gemm_grouped_tn_equals_separate_gemms(ops, K):
    """the four weight-gradient GEMMs of a layer as one persistent launch == four sd_gemm_bf16(trans_a, trans_b) calls,
    bit for bit; ragged M/N, 1..4 problems, fewer and more tiles than CUs; run twice (race screen)."""
    g = torch.Generator().manual_seed(K)
    shapes = [(4096, 1000), (1024, 2048), (6152, 1024), (520, 3072)]
    pairs = [(to_dev(bf(torch.randn(K, m, generator=g))), to_dev(bf(torch.randn(K, n, generator=g)))) for m, n in shapes]
    ref = [ops.gemm(a, b, True, True) for a, b in pairs]
    for n in (4, 1, 2):
        got = ops.gemm_grouped_tn(pairs[:n])
        again = ops.gemm_grouped_tn(pairs[:n])
        for i in range(n):
            assert torch.equal(got[i], ref[i]), (n, i, float((got[i].float() - ref[i].float()).abs().max()))
            assert torch.equal(got[i], again[i])
    # accumulate: C += A^T B, the old value joins the fp32 sum before the one rounding (== sd_gemm_bf16 with R = C)
    base = [to_dev(bf(torch.randn(m, n, generator=g) * 4)) for m, n in shapes]
    want = [ops.gemm(a, b, True, True, residual=c0) for (a, b), c0 in zip(pairs, base)]
    acc = [c0.clone() for c0 in base]
    ops.gemm_grouped_tn(pairs, accumulate_into=acc)
    for i in range(4):
        assert torch.equal(acc[i], want[i]), (i, float((acc[i].float() - want[i].float()).abs().max()))


@pytest.mark.parametrize("budget", [240, 64])
def test_persistent_weight_gradient_kernels_under_a_cu_budget(ops, budget, monkeypatch):
    """sd_debug_set("gemm.cu_budget") (SD_GEMM_CU_BUDGET in a multi-GPU run): the backward's persistent launches -- the grouped
    weight gradients and the lm_head-class TN GEMM -- on fewer workgroups than CUs give the same results bit for bit
    (a workgroup just walks more tiles), also while idle workgroups hold CU slots beside them (tests/csrc/cu_hog.hip)."""
    import ctypes
    import os
    g = torch.Generator().manual_seed(budget)
    K = 512
    shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]
    pairs = [(to_dev(bf(torch.randn(K, m, generator=g))), to_dev(bf(torch.randn(K, n, generator=g)))) for m, n in shapes]
    dl, xn = to_dev(bf(torch.randn(300, 40000, generator=g))), to_dev(bf(torch.randn(300, 1024, generator=g)))
    ref = ops.gemm_grouped_tn(pairs)
    ref_head = ops.gemm(dl, xn, True, True)  # [40000, 1024]: 1256 tiles of 256 x 128, the persistent TN kernel
    lib = ops.load_lib()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcu_hog.so")
    hog = None
    if os.path.exists(path):
        hog = ctypes.CDLL(path)
        hog.cu_hog.restype, hog.cu_hog.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    side = torch.cuda.Stream()
    _lib.debug_set("gemm.cu_budget", budget)
    try:
        for it in range(3):
            if hog is not None and it:
                assert hog.cu_hog(16, 500, side.cuda_stream) == 0
            got = ops.gemm_grouped_tn(pairs)
            got_head = ops.gemm(dl, xn, True, True)
            torch.cuda.synchronize()
            for i in range(4):
                assert torch.equal(got[i], ref[i]), (it, i)
            assert torch.equal(got_head, ref_head), it
    finally:
        _lib.debug_set("gemm.cu_budget", 0)


@pytest.mark.parametrize("M,T,Hq,H", [(2048, 512, 16, 1024), (4096, 512, 16, 1024), (300, 100, 2, 256)])
def test_gemm_odx_delta_epilogue(ops, M, T, Hq, H):
    """o-projection dX with delta = rowsum(dO * O) per (token, head) in the epilogue: d_ao bit-identical to the plain NN
    GEMM, delta equal to the fp32 row sums of the bf16 products (what attn_delta_kernel computes), ragged M."""
    g = torch.Generator().manual_seed(M + T)
    dy = to_dev(bf(torch.randn(M, H, generator=g)))
    wo = to_dev(bf(torch.randn(H, Hq * 128, generator=g) * 0.05))
    o = to_dev(bf(torch.randn(M, Hq * 128, generator=g)))
    dao, delta = ops.gemm_odx_delta(dy, wo, o, T, Hq)
    ref = ops.gemm(dy, wo, trans_b=True)
    assert torch.equal(dao, ref)
    want = (ref.float() * o.float()).view(M // T, T, Hq, 128).sum(-1).permute(0, 2, 1)
    check_close(f"odx_delta_M{M}", delta, want.double(), 1e-4)
    # and through the attention backward: delta given (o = None) == delta computed inside
    B, Hkv = M // T, max(1, Hq // 2)
    qkv = to_dev(bf(torch.randn(M, (Hq + 2 * Hkv) * 128, generator=g)))
    q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
    oo, lse = ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
    dao2, delta2 = ops.gemm_odx_delta(dy, wo, oo, T, Hq)
    a = ops.attn_bwd(q, k, v, oo, dao2, lse, B, T, Hq, Hkv)
    b = ops.attn_bwd(q, k, v, None, dao2, lse, B, T, Hq, Hkv, delta=delta2)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_gemm_split_k(ops):
    """few tiles + long K -> fp32 slabs + fixed-order reduce (the lm_head dX shape class)."""
    g = torch.Generator().manual_seed(6)
    M, N, K = 256, 128, 32768
    a, b = bf(torch.randn(M, K, generator=g)), bf(torch.randn(K, N, generator=g))
    r = bf(torch.randn(M, N, generator=g))
    lib = ops.load_lib()
    assert lib.sd_gemm_splitk_plan(M, N, K) > 1
    got = ops.gemm(to_dev(a), to_dev(b), False, True, residual=to_dev(r), split_k=True)
    check_close("gemm_split_k", got, _gemm_ref(a.float(), b.float(), False, True, r.float()), 6e-3, 3e-3)
    again = ops.gemm(to_dev(a), to_dev(b), False, True, residual=to_dev(r), split_k=True)
    assert torch.equal(got, again)


def test_gemm_accumulate_in_place(ops):
    g = torch.Generator().manual_seed(5)
    a, b = bf(torch.randn(160, 256, generator=g)), bf(torch.randn(160, 136, generator=g))
    c0 = bf(torch.randn(256, 136, generator=g) * 4)
    c = to_dev(c0.clone())
    ops.gemm(to_dev(a), to_dev(b), True, True, residual=c, out=c)
    check_close("gemm_accumulate", c, _gemm_ref(a.float(), b.float(), True, True, c0.float()), 6e-3, 3e-3)


# ------------------------------------------------------------------------------------ row kernels
@pytest.mark.parametrize("M,H", [(64, 128), (37, 1024), (256, 2048)])
def test_rmsnorm_fwd_bwd(ops, O, M, H):
    g = torch.Generator().manual_seed(M)
    x, w = bf(torch.randn(M, H, generator=g) * 2), bf(1 + 0.2 * torch.randn(H, generator=g))
    dy, dres = bf(torch.randn(M, H, generator=g)), bf(torch.randn(M, H, generator=g))
    y, rstd = ops.rmsnorm_fwd(to_dev(x), to_dev(w))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    check_close(f"rmsnorm_fwd_{M}x{H}", y, yr, 1.2e-2, 3e-3)
    (yr * dy.double()).sum().backward()
    dx, dw = ops.rmsnorm_bwd(to_dev(dy), to_dev(x), to_dev(w), rstd, dres=to_dev(dres))
    check_close(f"rmsnorm_bwd_dx_{M}x{H}", dx, xr.grad + dres.double(), 1.2e-2, 3e-3)
    check_close(f"rmsnorm_bwd_dw_{M}x{H}", dw, wr.grad, 1.2e-2, 4e-3)


def test_rmsnorm_bwd_consumes_splitk_slabs(ops):
    """dX GEMM (split-K, fp32 slabs left un-reduced) -> RMSNorm backward summing the slabs itself."""
    g = torch.Generator().manual_seed(21)
    M, H, K = 192, 1024, 16384
    a, bkn = bf(torch.randn(M, K, generator=g) * 0.05), bf(torch.randn(K, H, generator=g))
    x, w = bf(torch.randn(M, H, generator=g) * 2), bf(1 + 0.2 * torch.randn(H, generator=g))
    dres = bf(torch.randn(M, H, generator=g))
    _, rstd = ops.rmsnorm_fwd(to_dev(x), to_dev(w))
    dx, dw, nsp = ops.rmsnorm_bwd_from_splitk(to_dev(a), to_dev(bkn), to_dev(x), to_dev(w), rstd, to_dev(dres))
    assert nsp > 1
    dy = a.double() @ bkn.double()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)) * dy).sum().backward()
    check_close("rmsnorm_bwd_slabs_dx", dx, xr.grad + dres.double(), 1.2e-2, 3e-3)
    check_close("rmsnorm_bwd_slabs_dw", dw, wr.grad, 1.2e-2, 4e-3)


@pytest.mark.parametrize("B,T,Hq,Hkv", [(2, 24, 4, 2), (1, 130, 2, 1)])
def test_qknorm_rope_fwd_bwd(ops, O, B, T, Hq, Hkv):
    g = torch.Generator().manual_seed(T)
    M = B * T
    qkv = bf(torch.randn(M, (Hq + 2 * Hkv) * 128, generator=g))
    qg, kg = bf(1 + 0.2 * torch.randn(128, generator=g)), bf(1 + 0.2 * torch.randn(128, generator=g))
    cos, sin = ops.rope_tables(T, dev())
    out = ops.qknorm_rope_fwd(to_dev(qkv), to_dev(qg), to_dev(kg), cos, sin, T, Hq, Hkv)
    x = qkv.double().requires_grad_(True)
    qgr, kgr = qg.double().requires_grad_(True), kg.double().requires_grad_(True)
    q = x[:, :Hq * 128].view(B, T, Hq, 128)
    k = x[:, Hq * 128:(Hq + Hkv) * 128].view(B, T, Hkv, 128)
    c64, s64 = cos.cpu().double(), sin.cpu().double()

    def nr(t, gain):
        t = gain * (t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6))
        return O.Q.apply_rope(t.transpose(1, 2), c64, s64).transpose(1, 2)
    ref = torch.cat([nr(q, qgr).reshape(M, -1), nr(k, kgr).reshape(M, -1)], -1)
    check_close(f"qknorm_rope_fwd_T{T}", out, ref, 1.5e-2, 4e-3)
    dout = bf(torch.randn(ref.shape, generator=g))
    (ref * dout.double()).sum().backward()
    dqkv, dqg, dkg = ops.qknorm_rope_bwd(to_dev(dout), to_dev(qkv), to_dev(qg), to_dev(kg), cos, sin, T, Hq, Hkv)
    check_close(f"qknorm_rope_bwd_dx_T{T}", dqkv[:, :(Hq + Hkv) * 128], x.grad[:, :(Hq + Hkv) * 128], 1.5e-2, 4e-3)
    check_close(f"qknorm_rope_bwd_dqg_T{T}", dqg, qgr.grad, 1.5e-2, 5e-3)
    check_close(f"qknorm_rope_bwd_dkg_T{T}", dkg, kgr.grad, 1.5e-2, 5e-3)


def test_colsum_reduce_batch(ops):
    """Four column sums of different shapes in one launch (a layer's gain gradients), with and without accumulation,
    a strided partial array (the q/k gain layout [nb][256]) and the deferred-reduce form of the RMSNorm backward."""
    g = torch.Generator().manual_seed(3)
    p1 = to_dev(torch.randn(512, 1024, generator=g))
    p2 = to_dev(torch.randn(37, 2048, generator=g))
    pq = to_dev(torch.randn(300, 256, generator=g))
    o1 = to_dev(bf(torch.randn(1024, generator=g)))
    o2 = to_dev(bf(torch.zeros(2048)))
    oq, ok = to_dev(bf(torch.zeros(128))), to_dev(bf(torch.randn(128, generator=g)))
    want1 = (o1.float() + p1.sum(0)).bfloat16()
    wantk = (ok.float() + pq[:, 128:].sum(0)).bfloat16()
    ops.colsum_reduce_batch([(p1, o1, 1024, True), (p2, o2, 2048, False), (pq, oq, 128, False), (pq[:, 128:], ok, 128, True)])
    check_close("colsum_batch_1", o1, want1.double(), 8e-3)
    check_close("colsum_batch_2", o2, p2.double().sum(0), 8e-3)
    check_close("colsum_batch_q", oq, pq[:, :128].double().sum(0), 8e-3)
    check_close("colsum_batch_k", ok, wantk.double(), 8e-3)
    # deferred reduce == immediate reduce, bit for bit (same partials, same summation order)
    lib = ops.load_lib()
    M, H = 300, 1024
    x, dy = to_dev(bf(torch.randn(M, H, generator=g))), to_dev(bf(torch.randn(M, H, generator=g)))
    w = to_dev(bf(1 + 0.1 * torch.randn(H, generator=g)))
    y, rstd = ops.rmsnorm_fwd(x, w)
    dx_ref, dw_ref = ops.rmsnorm_bwd(dy, x, w, rstd)
    ws = torch.empty(lib.sd_rmsnorm_bwd_workspace_bytes(M, H), dtype=torch.uint8, device=x.device)
    dx = torch.empty_like(x)
    rc = lib.sd_rmsnorm_bwd2(dy.data_ptr(), x.data_ptr(), w.data_ptr(), rstd.data_ptr(), 0, dx.data_ptr(), 0, 0, ws.data_ptr(),
                             M, H, 0, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    nb = lib.sd_rmsnorm_bwd_partial_rows(M, H)
    dw = torch.zeros_like(w)
    ops.colsum_reduce_batch([(ws.view(torch.float32)[:nb * H].view(nb, H), dw, H, False)])
    assert torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref)


def test_swiglu_fwd_bwd(ops):
    g = torch.Generator().manual_seed(3)
    gu, dact = bf(torch.randn(70, 512, generator=g) * 2), bf(torch.randn(70, 256, generator=g))
    x = gu.double().requires_grad_(True)
    ref = torch.nn.functional.silu(x[:, :256]) * x[:, 256:]
    check_close("swiglu_fwd", ops.swiglu_fwd(to_dev(gu)), ref, 8e-3, 3e-3)
    (ref * dact.double()).sum().backward()
    check_close("swiglu_bwd", ops.swiglu_bwd(to_dev(dact), to_dev(gu)), x.grad, 8e-3, 3e-3)


def test_embedding_fwd_bwd(ops):
    g = torch.Generator().manual_seed(4)
    V, H, M = 520, 256, 96
    E = bf(torch.randn(V, H, generator=g))
    ids = torch.randint(0, V, (M,), generator=g)
    ids[10:20] = ids[0]  # duplicates
    x = ops.embedding_fwd(to_dev(ids), to_dev(E))
    assert torch.equal(x.cpu(), E[ids])
    dx = bf(torch.randn(M, H, generator=g))
    dE0 = bf(torch.randn(V, H, generator=g))
    dE = ops.embedding_bwd(to_dev(ids), to_dev(dx), to_dev(dE0.clone()))
    ref = dE0.double().index_add(0, ids, dx.double())
    check_close("embedding_bwd", dE, ref, 8e-3, 2e-3)


def test_embedding_bwd_many_duplicates_fixed_order(ops):
    """5 000 tokens over 300 ids (every id ~16 times, duplicates across the 32-token words of the kernel's bitmap): the
    gradient row of an id is base + the fp32 sum of its token rows taken in increasing token order -- bit for bit, and
    the same on every run (HF:381 embedding backward is an index_add with unspecified order)."""
    g = torch.Generator().manual_seed(14)
    V, H, M = 300, 1024, 5000
    ids = torch.randint(0, V, (M,), generator=g)
    dx = bf(torch.randn(M, H, generator=g))
    dE0 = bf(torch.randn(V, H, generator=g))
    got = ops.embedding_bwd(to_dev(ids), to_dev(dx), to_dev(dE0.clone())).cpu()
    again = ops.embedding_bwd(to_dev(ids), to_dev(dx), to_dev(dE0.clone())).cpu()
    assert torch.equal(got, again)
    want = dE0.clone()
    for v in range(V):
        rows = (ids == v).nonzero().reshape(-1)
        acc = torch.zeros(H, dtype=torch.float32)
        for i in rows.tolist():  # increasing token order, fp32, one addition at a time
            acc = acc + dx[i].float()
        want[v] = (dE0[v].float() + 1.0 * acc).bfloat16()
    assert torch.equal(got, want)


# -------------------------------------------------------------------------------------- attention
# the last three: B * Hkv is a multiple of 8, i.e. the XCD-aware workgroup numbering (attn_wg) instead of the natural one,
# with G = 2 / 1 query heads per kv head, an odd number of tile pairs and a ragged last tile
@pytest.mark.parametrize("B,T,Hq,Hkv,pad", [(2, 128, 4, 2, False), (1, 200, 2, 1, True), (2, 64, 2, 2, True),
                                            (1, 512, 4, 2, False), (1, 40, 2, 1, False), (2, 200, 8, 4, True),
                                            (3, 72, 8, 8, True), (4, 330, 4, 2, False)])
def test_attention_fwd_bwd(ops, O, B, T, Hq, Hkv, pad):
    g = torch.Generator().manual_seed(T + Hq)
    M = B * T
    q, k, v = (bf(torch.randn(M, h * 128, generator=g)) for h in (Hq, Hkv, Hkv))
    do = bf(torch.randn(M, Hq * 128, generator=g))
    kv_len = None
    if pad:
        kv_len = torch.tensor([max(1, T - 7 - 11 * b) for b in range(B)], dtype=torch.int32)
    o, lse = ops.attn_fwd(to_dev(q), to_dev(k), to_dev(v), B, T, Hq, Hkv, None if kv_len is None else to_dev(kv_len))
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref = O.Q.attention(qr.view(B, T, Hq, 128).transpose(1, 2), kr.view(B, T, Hkv, 128).transpose(1, 2),
                        vr.view(B, T, Hkv, 128).transpose(1, 2), None if kv_len is None else kv_len.long())
    ref = ref.transpose(1, 2).reshape(M, Hq * 128)
    check_close(f"attn_fwd_B{B}T{T}H{Hq}/{Hkv}p{int(pad)}", o, ref, 1.5e-2, 4e-3)
    (ref * do.double()).sum().backward()
    dq, dk, dv = ops.attn_bwd(to_dev(q), to_dev(k), to_dev(v), o, to_dev(do), lse, B, T, Hq, Hkv,
                              None if kv_len is None else to_dev(kv_len))
    check_close(f"attn_bwd_dq_B{B}T{T}p{int(pad)}", dq, qr.grad, 2e-2, 6e-3)
    check_close(f"attn_bwd_dk_B{B}T{T}p{int(pad)}", dk, kr.grad, 2e-2, 6e-3)
    check_close(f"attn_bwd_dv_B{B}T{T}p{int(pad)}", dv, vr.grad, 2e-2, 6e-3)


@pytest.mark.parametrize("B,T,Hq,Hkv,pad", [(4, 512, 16, 8, False), (1, 2048, 4, 2, False), (2, 330, 4, 2, True),
                                            (3, 72, 8, 8, True), (1, 40, 2, 1, False), (2, 576, 2, 1, True),
                                            (8, 64, 2, 1, False), (2, 1000, 4, 4, True)])
def test_attention_fwd_pipelined_equals_classic(ops, B, T, Hq, Hkv, pad):
    """attn_fwd_pipe_kernel (S^T of tile i+1 issued before tile i's softmax; the default) keeps the arithmetic and its
    order, so output and log-sum-exp equal attn_fwd_kernel's bit for bit (one to nine K/V tiles per stream, odd pair
    counts, ragged last tile, right padding, the XCD-aware and the natural workgroup numbering)."""
    from speech_distill_amd._lib import load_lib
    g = torch.Generator().manual_seed(7 * T + Hq)
    M = B * T
    qkv = bf(torch.randn(M, (Hq + 2 * Hkv) * 128, generator=g) * 1.7).to(dev())
    q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
    kv_len = None
    if pad:
        kv_len = torch.tensor([max(1, T - 5 - 37 * b) for b in range(B)], dtype=torch.int32, device=dev())
    lib = load_lib()
    outs = []
    try:
        for variant in (2, 1):
            _lib.debug_set("attn.variant", variant)
            o, lse = ops.attn_fwd(q, k, v, B, T, Hq, Hkv, kv_len)
            torch.cuda.synchronize()
            outs.append((o.clone(), lse.clone()))
    finally:
        _lib.debug_set("attn.variant", 0)
    assert torch.isfinite(outs[0][0].float()).all()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])


# ------------------------------------------------------------------------------------------ top-K
def test_topk_golden_fp32(ops):
    """G2: the reference's own train.py:80-91 outputs (fp32 logits, no ties) must be reproduced bit for bit
    in the indices and to one fp16 ulp in the values."""
    z = load_golden("g2_extract.npz")
    for name in "abc":
        Vs, K = [int(x) for x in z[f"{name}_meta"]]
        v, i = ops.logsoftmax_topk(to_dev(torch.from_numpy(z[f"{name}_logits"])), K, Vs)
        np.testing.assert_array_equal(i.cpu().numpy(), z[f"{name}_i"], err_msg=name)
        dv = np.abs(v.cpu().float().numpy() - z[f"{name}_v"].astype(np.float32))
        record(f"topk_golden_{name}", max_abs=float(dv.max()))
        assert dv.max() <= 1e-2 * 0 + 8e-3, f"{name}: values differ by {dv.max()}"


@pytest.mark.parametrize("rows,V,K,dtype", [(5, 4096, 128, "bf16"), (3, 159488, 128, "bf16"), (4, 1000, 100, "bf16"),
                                            (2, 159488, 100, "fp32"), (3, 64, 64, "bf16")])
def test_topk_vs_oracle(ops, O, rows, V, K, dtype):
    """bf16 logits have many exact ties: values must match exactly, indices as a set modulo ties, ties to the lowest index."""
    g = torch.Generator().manual_seed(V + K)
    x = torch.randn(rows, V, generator=g) * 3
    x = bf(x) if dtype == "bf16" else x
    v, i = ops.logsoftmax_topk(to_dev(x), K)
    rv, ri = O.L.extract_topk(x.float(), K)
    np.testing.assert_array_equal(i.cpu().numpy(), ri.numpy())
    d = (v.cpu().float() - rv.float()).abs().max()
    record(f"topk_{rows}x{V}_{dtype}", max_abs=float(d))
    assert d <= 8e-3


def test_topk_all_equal_row(ops):
    x = bf(torch.zeros(2, 1024))
    x[1, 500] = 1.0
    v, i = ops.logsoftmax_topk(to_dev(x), 16)
    assert i[0].cpu().tolist() == list(range(16))
    assert i[1].cpu().tolist() == [500] + list(range(15))


# ------------------------------------------------------------------------------------------- loss
G1 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "g1_loss_*.npz")))


@pytest.mark.parametrize("fname", G1)
def test_loss_golden(ops, fname):
    """G1: the reference DistillationLoss's own outputs.  fp32 fixtures: losses to 2e-5, gradient to 1e-4 of
    its max; bf16 fixtures (reference ran log-softmax in bf16): 3e-2 (SURVEY.md section 8d tolerance)."""
    from speech_distill_amd import DistillationLoss
    z = load_golden(fname)
    is_bf16 = str(z["dtype"]) == "bf16"
    cast = bf if is_bf16 else (lambda t: t)
    labels = to_dev(torch.from_numpy(z["labels"]))
    sm = to_dev(torch.from_numpy(z["speech_mask"])) if "speech_mask" in z.files else None
    for key in [k[:-7] for k in z.files if k.endswith("_losses")]:
        mode, Tm, al = key.split("_")
        s = to_dev(cast(torch.from_numpy(z["student"]))).requires_grad_(True)
        kw = (dict(teacher_logits=to_dev(cast(torch.from_numpy(z["teacher"])))) if mode == "dense" else
              dict(teacher_top_k_v=to_dev(torch.from_numpy(z["top_v"])), teacher_top_k_i=to_dev(torch.from_numpy(z["top_i"]))))
        fn = DistillationLoss(temperature=float(Tm[1:]), alpha=float(al[1:]))
        out = fn(s, labels, speech_token_mask=sm, **kw)
        got = np.array([float(x) for x in out])
        ref = z[key + "_losses"]
        rtol = 3e-2 if is_bf16 else 2e-5
        record(f"loss_{fname}_{key}", got=got.tolist(), ref=ref.tolist())
        np.testing.assert_allclose(got, ref, rtol=rtol, atol=2e-6, err_msg=key)
        out[0].backward()
        gref = torch.from_numpy(z[key + "_grad"])
        if float(gref.abs().max()) == 0:
            assert float(s.grad.abs().max()) == 0
        else:
            check_close(f"lossgrad_{fname}_{key}", s.grad, gref, 3e-2 if is_bf16 else 1e-4)


def test_loss_no_teacher_raises(ops):
    from speech_distill_amd import DistillationLoss
    with pytest.raises(ValueError, match="Either teacher_logits or top_k must be provided"):
        DistillationLoss()(to_dev(torch.randn(1, 4, 64)), to_dev(torch.tensor([[1, 2, 3, 4]])))


@pytest.mark.parametrize("mode", ["sparse", "dense"])
def test_loss_full_vocab_vs_oracle(ops, O, mode):
    """V = 159 488 (the real vocab), bf16 logits, against the fp64 oracle on the same bf16 values."""
    from speech_distill_amd import DistillationLoss
    g = torch.Generator().manual_seed(7)
    B, T, V, K = 2, 6, 159488, 128
    s = bf(torch.randn(B, T, V, generator=g) * 2)
    t = bf(torch.randn(B, T, V, generator=g) * 2)
    labels = torch.randint(152927, V, (B, T), generator=g)
    labels[:, :2] = -100
    tv, ti = O.L.extract_topk(t.float(), K)
    kw_o = dict(teacher_logits=t.float()) if mode == "dense" else dict(teacher_top_k_v=tv, teacher_top_k_i=ti)
    ref = O.L.distill_loss(s.float(), labels, temperature=2.0, alpha=0.5, return_grad=True, **kw_o)
    sd = to_dev(s).requires_grad_(True)
    kw = dict(teacher_logits=to_dev(t)) if mode == "dense" else dict(teacher_top_k_v=to_dev(tv), teacher_top_k_i=to_dev(ti))
    out = DistillationLoss(2.0, 0.5)(sd, to_dev(labels), **kw)
    got = np.array([float(x) for x in out])
    np.testing.assert_allclose(got, [float(x) for x in ref[:4]], rtol=2e-5, atol=1e-6)
    out[0].backward()
    # gradient is stored in bf16: one rounding of each entry
    check_close(f"lossgrad_fullV_{mode}", sd.grad, ref[4], 8e-3, 3e-3)
    # size-independent property: every valid row's gradient sums to zero (softmax rows sum to 1, sum q = 1)
    rs = sd.grad.float().sum(-1).abs().max()
    record(f"lossgrad_rowsum_{mode}", max_rowsum=float(rs))
    assert float(rs) < 2e-3


def test_loss_inplace_grad_matches(ops):
    from speech_distill_amd import DistillationLoss
    g = torch.Generator().manual_seed(9)
    s = bf(torch.randn(2, 9, 1000, generator=g))
    labels = torch.randint(0, 1000, (2, 9), generator=g)
    tv, ti = torch.topk(torch.log_softmax(torch.randn(2, 9, 1000, generator=g), -1), 50)
    a = to_dev(s).requires_grad_(True)
    b_leaf = to_dev(s).requires_grad_(True)
    b = b_leaf * 1  # non-leaf so the buffer may be overwritten
    DistillationLoss(2.0, 0.5)(a, to_dev(labels), teacher_top_k_v=to_dev(tv.half()), teacher_top_k_i=to_dev(ti.int()))[0].backward()
    DistillationLoss(2.0, 0.5, inplace_grad=True)(b, to_dev(labels), teacher_top_k_v=to_dev(tv.half()),
                                                  teacher_top_k_i=to_dev(ti.int()))[0].backward()
    assert torch.equal(a.grad, b_leaf.grad)


# -------------------------------------------------------------------------------------- optimizer
def test_adamw_and_clip(ops):
    g = torch.Generator().manual_seed(11)
    n = 100003
    p, gr = bf(torch.randn(n, generator=g)), bf(torch.randn(n, generator=g))
    m, v = bf(torch.randn(n, generator=g) * 0.1), bf(torch.rand(n, generator=g) * 0.1)
    pd, gd, md, vd = (to_dev(t.clone()) for t in (p, gr, m, v))
    ss = torch.zeros(1, device=dev())
    ops.sumsq(gd, ss)
    np.testing.assert_allclose(float(ss), float(gr.double().pow(2).sum()), rtol=1e-4)
    lr, b1, b2, eps, wd, step, mx = 1e-2, 0.9, 0.999, 1e-8, 0.01, 3, 1.0
    ops.adamw_(pd, gd, md, vd, lr, b1, b2, eps, wd, step, ss, mx)
    clip = min(1.0, mx / (float(gr.double().pow(2).sum().sqrt()) + 1e-6))
    P, G, Mo, Vo = p.double(), gr.double() * clip, m.double(), v.double()
    P = P * (1 - lr * wd)
    Mo = b1 * Mo + (1 - b1) * G
    Vo = b2 * Vo + (1 - b2) * G * G
    P = P - lr / (1 - b1 ** step) * Mo / (Vo.sqrt() / (1 - b2 ** step) ** 0.5 + eps)
    check_close("adamw_p", pd, P, 8e-3, 3e-3)
    check_close("adamw_m", md, Mo, 8e-3, 3e-3)
    check_close("adamw_v", vd, Vo, 8e-3, 3e-3)


# ------------------------------------------------------------------------------------------ loss rows
@pytest.mark.parametrize("B,T", [(1, 7), (4, 512), (3, 341), (5, 1000), (2, 2048)])
def test_loss_rows_kernel_equals_the_reference_rule(ops, B, T):
    """sd_loss_rows (one launch) == the shift / -100 / speech-mask rule of distillation_loss.py:31-45 as the ORACLE
    states it (oracle/distill_loss.py::_rows): same rows in the same order, same predicted labels; bit-exact
    (index work).  Also: no valid row, every row valid, and the right-padding check of both attention masks."""
    g = torch.Generator().manual_seed(B * T)
    labels = torch.randint(0, 1000, (B, T), generator=g)
    labels[torch.rand(B, T, generator=g) < 0.4] = -100
    speech = (torch.rand(B, T, generator=g) < 0.7).long()
    lens = torch.randint(1, T + 1, (B,), generator=g)
    am = (torch.arange(T)[None, :] < lens[:, None]).long()
    from oracle.distill_loss import _rows

    def oracle_rows(lab, sm):
        # the ORACLE's statement of the rule (oracle/distill_loss.py::_rows, distillation_loss.py:31-45): positions of
        # the shifted view [B, T-1] that are valid, as flat indices b*T+t into the unshifted [B, T] grid, and their labels
        _, y, valid = _rows(torch.zeros(B, T, 1), lab.long(), sm)
        pos = torch.nonzero(valid.reshape(B, T - 1))
        return pos[:, 0] * T + pos[:, 1], y[valid]
    for sm in (None, speech):
        want_r, want_l = oracle_rows(labels, sm)
        got_r, got_l = ops.loss_rows(to_dev(labels), None if sm is None else to_dev(sm), right_padded=(to_dev(am), None))
        assert torch.equal(got_r.cpu(), want_r) and torch.equal(got_l.cpu(), want_l), (B, T, sm is None)
        # the host logic's CPU branch (used by the trainer for CPU batches) states the same rule
        cpu_r, cpu_l = ops.loss_rows(labels, sm, right_padded=(am, am))
        assert torch.equal(cpu_r, want_r) and torch.equal(cpu_l, want_l)
    none_r, none_l = ops.loss_rows(to_dev(torch.full((B, T), -100)))
    assert none_r.numel() == 0 and none_l.numel() == 0
    all_r, all_l = ops.loss_rows(to_dev(torch.ones(B, T, dtype=torch.long)))
    assert all_r.numel() == B * (T - 1)
    if T > 2:
        bad = am.clone()
        bad[B - 1] = 1
        bad[B - 1, 0] = 0  # left padding in the last sequence
        with pytest.raises(ValueError, match="right-padded"):
            ops.loss_rows(to_dev(labels), None, right_padded=(to_dev(am), to_dev(bad)))
        # a float / bool mask is accepted too
        ops.loss_rows(to_dev(labels), to_dev(speech).bool(), right_padded=(to_dev(am).float(),))
    # int32 labels keep their VALUES (token ids and -100), they are not read as a 0/1 mask
    want_r, want_l = oracle_rows(labels, None)
    got_r, got_l = ops.loss_rows(to_dev(labels.to(torch.int32)), None)
    assert torch.equal(got_r.cpu(), want_r) and torch.equal(got_l.cpu(), want_l) and got_l.dtype == torch.int64
    # a mask that is not on the labels' [B, T] grid (a teacher batch padded to another length) never reaches the kernel
    with pytest.raises(ValueError, match="!= labels"):
        ops.loss_rows(to_dev(labels), None, right_padded=(to_dev(am), to_dev(torch.ones(B, T + 3, dtype=torch.long))))


def test_gradient_norm_reduction_is_bitwise_reproducible(ops):
    """sd_sumsq_bf16 feeds the clip coefficient of the fused AdamW: data-parallel ranks hold identical gradients and must
    compute the identical norm, so the reduction has a fixed order (no atomics).  Same buffer, 20 launches, one value;
    and the value is right (fp64 reference)."""
    g = torch.Generator().manual_seed(5)
    x = to_dev(bf(torch.randn(7_340_033, generator=g)))  # not a multiple of 8: tail path too
    outs = []
    for _ in range(20):
        out = torch.zeros(1, dtype=torch.float32, device=x.device)
        ops.sumsq(x, out)
        outs.append(float(out))
    assert len(set(outs)) == 1, sorted(set(outs))
    ref = float(x.double().pow(2).sum())
    assert abs(outs[0] - ref) <= 1e-5 * ref
    acc = torch.full((1,), 3.0, dtype=torch.float32, device=x.device)  # accumulates into `out`
    ops.sumsq(x, acc)
    assert abs(float(acc) - 3.0 - outs[0]) <= 1e-6 * outs[0]
    # the partial sums live in the CALLER's scratch: two reductions in flight on two streams do not disturb each other
    y = to_dev(bf(torch.randn(5_000_000, generator=g) * 3))
    ref_y = float(y.double().pow(2).sum())
    s2 = torch.cuda.Stream()
    px, py = (torch.empty(2048, dtype=torch.float32, device=x.device) for _ in range(2))
    for _ in range(10):
        ox, oy = (torch.zeros(1, dtype=torch.float32, device=x.device) for _ in range(2))
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            ops.sumsq(y, oy, py)
        ops.sumsq(x, ox, px)
        torch.cuda.synchronize()
        assert float(ox) == outs[0] and abs(float(oy) - ref_y) <= 1e-5 * ref_y
