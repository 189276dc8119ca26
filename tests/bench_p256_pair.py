"""GPU micro-benchmark (not a pytest): gemm_p256_kernel with whole-line slots (PAIR, round 3) against its 32-deep half-line
stages (round 2, sd_gemm_force_variant(0, 0x400)) and against the 256x128 persistent kernel (SD_GEMM_NO_P256 in a second
process), interleaved in one process, operands flushed from the Infinity Cache between launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops  # noqa: E402
from bench_pair import timeit  # noqa: E402

dev = torch.device("cuda:0")
lib = ops.load_lib()
flush = torch.zeros(128 * 1024 * 1024, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for name, M, K, N, sw in (("teacher lm_head", 1536, 2048, 159488, False), ("student lm_head", 1536, 1024, 159488, False),
                          ("teacher gate|up + SwiGLU", 2048, 2048, 12288, True), ("teacher lm_head B*T rows", 2048, 2048, 159488, False)):
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    fl = 2.0 * M * N * K

    def f():
        return ops.gemm_swiglu(x, w, save_gu=False)[0] if sw else ops.gemm(x, w)
    res = {}
    outs = {}
    for rnd in range(2):
        for tag, v in (("paired", 0), ("half-line", 0x400)):
            _lib.gemm_force_variant(0, v)
            outs[tag] = f()
            res.setdefault(tag, []).append(timeit(f, iters=10, warm=2, flush=flush))
    _lib.gemm_force_variant(0, 0)
    same = torch.equal(outs["paired"], outs["half-line"])
    p, h = min(res["paired"]), min(res["half-line"])
    print(f"{name:28s} identical={same}  paired {p:7.1f} us ({fl / p / 1e6:5.0f} TF/s)   half-line stages {h:7.1f} us "
          f"({fl / h / 1e6:5.0f} TF/s)", flush=True)
