"""GPU micro-benchmark (not a pytest): the four dX GEMMs of a student layer at M = 2048 as the backward runs them
(fused epilogues), per tile variant, with operands flushed from the Infinity Cache between launches (--cold)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops  # noqa: E402
from bench_pair import timeit  # noqa: E402

dev = torch.device("cuda:0")
lib = ops.load_lib()
M, h, I, QD, QKV, T, Hq = 2048, 1024, 3072, 2048, 4096, 512, 16
flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*s, sc=1.0):
    return (torch.randn(*s, device=dev, generator=g) * sc).bfloat16()


dy, wdown, gu = rnd(M, h), rnd(h, I, sc=0.02), rnd(M, 2 * I)
wo, ao = rnd(h, QD, sc=0.02), rnd(M, QD)
dqkv, wqkv = rnd(M, QKV), rnd(QKV, h, sc=0.02)
cases = (("down dX + SwiGLU bwd (N=3072, K=1024)", 2.0 * M * I * h, lambda: ops.gemm_swiglu_bwd(dy, wdown, gu)),
         ("down dX plain", 2.0 * M * I * h, lambda: ops.gemm(dy, wdown, False, True)),
         ("o dX + delta (N=2048, K=1024)", 2.0 * M * QD * h, lambda: ops.gemm_odx_delta(dy, wo, ao, T, Hq)),
         ("o dX plain", 2.0 * M * QD * h, lambda: ops.gemm(dy, wo, False, True)),
         ("qkv dX (N=1024, K=4096)", 2.0 * M * h * QKV, lambda: ops.gemm(dqkv, wqkv, False, True)))
for name, fl, fn in cases:
    for bm, nst in ((0, 0), (64, 3), (64, 4), (128, 3), (128, 4), (256, 9)):
        _lib.gemm_force_variant(bm, nst)
        try:
            t = timeit(fn, flush=flush)
            print(f"{name:40s} variant {bm or 'auto'}/{nst or ''}: {t:6.1f} us  {fl / t / 1e6:6.0f} TF/s", flush=True)
        except Exception as e:
            print(f"{name:40s} variant {bm}/{nst}: {type(e).__name__}", flush=True)
    _lib.gemm_force_variant(0, 0)
