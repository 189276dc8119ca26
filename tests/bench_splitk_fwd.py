import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for name, M, N, K in (("teacher down", 2048, 2048, 6144), ("teacher o", 2048, 2048, 2048), ("student down", 2048, 1024, 3072), ("student o", 2048, 1024, 2048)):
    a = torch.randn(M, K, device=dev).bfloat16(); b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    # cold weights: rotate over 8 weight copies
    bs = [b.clone() for _ in range(8)]
    i = [0]
    def f0():
        i[0] = (i[0] + 1) % 8
        ops.gemm(a, bs[i[0]], residual=r, out=out)
    def f1():
        i[0] = (i[0] + 1) % 8
        ops.gemm(a, bs[i[0]], residual=r, out=out, split_k=True)
    t0, t1 = timeit(f0), timeit(f1)
    print(f"{name:14s} M={M} N={N} K={K}: plain {t0:6.1f} us ({2.0*M*N*K/t0/1e6:5.0f} TF/s)   split-K {t1:6.1f} us ({2.0*M*N*K/t1/1e6:5.0f} TF/s)", flush=True)
