"""Runs ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: pmc_gemm.py M N K ta tb [bm nst]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops
M, N, K, ta, tb = [int(x) for x in sys.argv[1:6]]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
a = torch.randn((K, M) if ta else (M, K), device=dev, generator=g).bfloat16()
b = torch.randn((K, N) if tb else (N, K), device=dev, generator=g).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
if len(sys.argv) > 7:
    _lib.gemm_force_variant(int(sys.argv[6]), int(sys.argv[7]))
for _ in range(5):
    ops.gemm(a, b, bool(ta), bool(tb), out=c)
torch.cuda.synchronize()
