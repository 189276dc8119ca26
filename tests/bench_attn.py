"""GPU micro-benchmark (not a pytest): attention forward / backward at B*T = 2048 tokens for several T, to separate the
fixed cost of a launch from the per-K/V-tile cost.  Also Hq sweep at T=512."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=5):
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
    Hq, Hkv = 16, 8
    for B, T in ((32, 64), (16, 128), (8, 256), (4, 512), (2, 1024), (1, 2048), (8, 512), (16, 512), (4, 1024), (4, 2048)):
        M = B * T
        qkv = torch.randn(M, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
        q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
        o, lse = ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
        do = torch.randn_like(o)
        tf = timeit(lambda: ops.attn_fwd(q, k, v, B, T, Hq, Hkv))
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, T, Hq, Hkv))
        fl = 2.0 * B * Hq * T * T * 128
        print(f"B={B:3d} T={T:5d}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF/s)   bwd(delta+dq+dkv, incl. 3 zeros_like) {tb:7.1f} us",
              flush=True)


if __name__ == "__main__":
    main()
