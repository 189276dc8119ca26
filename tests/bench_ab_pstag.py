"""GPU micro-benchmark (not a pytest): gemm_pstag_kernel against gemm_pgroup_nt_kernel on the SAME single problem."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from bench_pair import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 2048
g = torch.Generator(device=dev).manual_seed(0)
flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None
for name, K, I in (("teacher gate|up", 2048, 6144), ("student gate|up", 1024, 3072)):
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(2 * I, K, device=dev, generator=g) * 0.02).bfloat16()
    fl = 2.0 * M * 2 * I * K
    for keep in (False, True):
        t0 = timeit(lambda: ops.gemm_swiglu(x, w, save_gu=keep), flush=flush)
        t1 = timeit(lambda: ops.gemm_grouped_nt([(x, w, 1)], swiglu=True), flush=flush)
        print(f"{name} swiglu keep_gu={keep}: pstag/p256 {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF/s)   pgroup_nt(n=1, gu kept) {t1:6.1f} us "
              f"({fl / t1 / 1e6:5.0f} TF/s)", flush=True)
    t0 = timeit(lambda: ops.gemm(x, w), flush=flush)
    t1 = timeit(lambda: ops.gemm_grouped_nt([(x, w, 1)]), flush=flush)
    print(f"{name} plain: dispatch {t0:6.1f} us ({fl / t0 / 1e6:5.0f} TF/s)   pgroup_nt(n=1) {t1:6.1f} us ({fl / t1 / 1e6:5.0f} TF/s)",
          flush=True)
