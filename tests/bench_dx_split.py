"""GPU micro-benchmark (not a pytest): the dX GEMMs of a layer at M = 2048 (student), unsplit vs split along K with the
consumer-side slab reduce counted, for several tile shapes.  SD_SPLITK_MIN_KT is read once per process, so each setting
is a subprocess-free rerun of this script:  SD_SPLITK_MIN_KT=64 python tests/bench_dx_split.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ops.load_lib()


def timeit(fn, iters=40, warm=5):
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


def main():
    M, h = 2048, 1024
    for name, K in (("qkv dX", 4096), ("gate|up dX", 6144), ("o dX (N=2048)", 1024)):
        N = 2048 if "o dX" in name else h
        dy = (torch.randn(M, K, device=dev) * 0.1).bfloat16()
        w = (torch.randn(K, N, device=dev) * 0.02).bfloat16()
        x = torch.randn(M, N, device=dev).bfloat16()
        gain = torch.ones(N, device=dev).bfloat16()
        _, rstd = ops.rmsnorm_fwd(x, gain)
        plan = lib.sd_gemm_splitk_plan(M, N, K)
        nb = max(lib.sd_gemm_splitk_workspace_bytes(M, N, K), 16)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        for bm, nst in ((0, 0), (64, 3), (128, 3), (256, 9)):
            _lib.gemm_force_variant(bm, nst)
            nsp = C.c_int(0)

            def gemm():
                ops.check(lib.sd_gemm_bf16_splitk_partial(dy.data_ptr(), w.data_ptr(), out.data_ptr(), M, N, K, K, N, N, 0, 1,
                                                          ws.data_ptr(), nb, C.byref(nsp), torch.cuda.current_stream().cuda_stream),
                          "splitk_partial")
            try:
                t = timeit(gemm)
            except Exception as e:  # a forced variant that does not exist for this layout
                print(f"{name:16s} K={K} variant {bm}/{nst}: {e}")
                continue
            print(f"{name:16s} K={K} plan={plan} variant {bm or 'auto'}/{nst or ''}: nsplit={nsp.value} {t:6.1f} us  "
                  f"{2.0 * M * N * K / t / 1e6:6.0f} TF/s", flush=True)
        _lib.gemm_force_variant(0, 0)


if __name__ == "__main__":
    main()
