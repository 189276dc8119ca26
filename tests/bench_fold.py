"""GPU measurement (not a pytest): the folded-RMSNorm GEMM variants against the plain ones on the same operands, at the
config-2 and config-5 token counts (teacher widths): does the row-scale / sum-of-squares epilogue cost anything by itself?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


g = torch.Generator(device=dev).manual_seed(0)
H, I, Hq, Hkv, T = 2048, 6144, 16, 8, 512
for M in (2048, 8192, 32768):
    x = torch.randn(M, H, device=dev, generator=g).bfloat16()
    r = torch.randn(M, H, device=dev, generator=g).bfloat16()
    ws = [torch.randn(H, H, device=dev, generator=g).bfloat16() * 0.02 for _ in range(40)]
    wq = [torch.randn((Hq + 2 * Hkv) * 128, H, device=dev, generator=g).bfloat16() * 0.02 for _ in range(20)]
    wg = [torch.randn(2 * I, H, device=dev, generator=g).bfloat16() * 0.02 for _ in range(8)]
    qg = torch.ones(128, device=dev, dtype=torch.bfloat16)
    cos, sin = ops.rope_tables(T, dev)
    _, ssq = ops.gemm_resid_ssq(x, ws[0], r)
    i = [0]

    def nxt(lst):
        i[0] += 1
        return lst[i[0] % len(lst)]
    t_o = timeit(lambda: ops.gemm(x, nxt(ws), residual=r))
    t_os = timeit(lambda: ops.gemm_resid_ssq(x, nxt(ws), r))
    t_q = timeit(lambda: ops.gemm_qkv_rope(x, nxt(wq), qg, qg, cos, sin, T, Hq, Hkv))
    t_qs = timeit(lambda: ops.gemm_qkv_rope_rs(x, nxt(wq), qg, qg, cos, sin, ssq, T, Hq, Hkv))
    t_g = timeit(lambda: ops.gemm_swiglu(x, nxt(wg), save_gu=False))
    t_gs = timeit(lambda: ops.gemm_swiglu_rs(x, nxt(wg), ssq))
    print(f"M={M:6d}: o+resid {t_o:8.1f} us  +ssq_out {t_os:8.1f} | qkv+rope {t_q:8.1f}  +ssq_in {t_qs:8.1f} | gate|up+swiglu {t_g:8.1f}  +ssq_in {t_gs:8.1f}",
          flush=True)
    del x, r, ws, wq, wg
    torch.cuda.empty_cache()
