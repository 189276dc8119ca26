"""A/B timing of the fused GEMM epilogues against the separate launchers at the real shapes (cold weights)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
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
M, T, Hq, Hkv = 2048, 512, 16, 8
cos, sin = ops.rope_tables(T, dev)
qg = torch.ones(128, device=dev, dtype=torch.bfloat16)
for tag, h, I in (("student", 1024, 3072), ("teacher", 2048, 6144)):
    x = torch.randn(M, h, device=dev, generator=g).bfloat16()
    n = max(2, int(600e6 // (4096 * h * 2)) + 1)
    wq = [(torch.randn(4096, h, device=dev, generator=g) * 0.05).bfloat16() for _ in range(n)]
    n2 = max(2, int(600e6 // (2 * I * h * 2)) + 1)
    wg = [(torch.randn(2 * I, h, device=dev, generator=g) * 0.05).bfloat16() for _ in range(n2)]
    r = [0]

    def nxt(lst):
        r[0] += 1
        return lst[r[0] % len(lst)]
    t_sep = timeit(lambda: ops.qknorm_rope_fwd(ops.gemm(x, nxt(wq)), qg, qg, cos, sin, T, Hq, Hkv))
    t_gemm = timeit(lambda: ops.gemm(x, nxt(wq)))
    t_fus = timeit(lambda: ops.gemm_qkv_rope(x, nxt(wq), qg, qg, cos, sin, T, Hq, Hkv))
    print(f"{tag} qkv : gemm {t_gemm:6.1f}  gemm+norm_rope {t_sep:6.1f}  fused {t_fus:6.1f} us", flush=True)
    t_sep = timeit(lambda: ops.swiglu_fwd(ops.gemm(x, nxt(wg))))
    t_gemm = timeit(lambda: ops.gemm(x, nxt(wg)))
    t_fus = timeit(lambda: ops.gemm_swiglu(x, nxt(wg)))
    t_fus2 = timeit(lambda: ops.gemm_swiglu(x, nxt(wg), save_gu=False))
    print(f"{tag} gu  : gemm {t_gemm:6.1f}  gemm+swiglu {t_sep:6.1f}  fused {t_fus:6.1f}  fused(no gu) {t_fus2:6.1f} us", flush=True)
