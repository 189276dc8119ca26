"""GPU micro-benchmark (not a pytest): residual projection + the RMSNorm behind it (o / down of a decoder layer at M = 2048):
today's pair (GEMM with the residual in its epilogue, then sd_rmsnorm_fwd) against the persistent K-sliced GEMM + the
slab-summing norm (sd_gemm_grouped_nt + sd_rmsnorm_fwd_slabs).  --cold: operands leave the Infinity Cache between runs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from bench_pair import timeit  # noqa: E402

dev = torch.device("cuda:0")
M = 2048
flush = torch.zeros(128 * 1024 * 1024, device=dev) if "--cold" in sys.argv else None
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in (("teacher o", 2048, 2048), ("teacher down", 2048, 6144), ("student o", 1024, 2048), ("student down", 1024, 3072)):
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    r = torch.randn(M, N, device=dev, generator=g).bfloat16()
    gain = torch.ones(N, device=dev).bfloat16()

    def today():
        return ops.rmsnorm_fwd(ops.gemm(x, w, residual=r), gain)
    t = timeit(today, flush=flush)
    print(f"{name:14s} GEMM(+residual) + norm          {t:7.1f} us", flush=True)
    for ns in (2, 3, 4):
        def sliced():
            return ops.rmsnorm_fwd_slabs(ops.gemm_grouped_nt([(x, w, ns)])[0], r, gain)
        t = timeit(sliced, flush=flush)
        print(f"{name:14s} persistent {ns} K slices + slab norm {t:7.1f} us", flush=True)
