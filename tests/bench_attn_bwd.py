"""GPU micro-benchmark (not a pytest): the two attention backward kernels separately (HIP events of the library's own
profiling scopes) over T and the number of query heads per kv head, to separate the fixed cost of a launch from the cost
of one (Q, dO) tile visit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def main():
    for Hq, Hkv in ((16, 8), (8, 8)):
        for B, T in ((32, 64), (16, 128), (8, 256), (4, 512), (2, 1024), (1, 2048), (16, 512)):
            M = B * T
            qkv = torch.randn(M, (Hq + 2 * Hkv) * 128, device=dev).bfloat16()
            q, k, v = qkv[:, :Hq * 128], qkv[:, Hq * 128:(Hq + Hkv) * 128], qkv[:, (Hq + Hkv) * 128:]
            o, lse = ops.attn_fwd(q, k, v, B, T, Hq, Hkv)
            do = torch.randn_like(o)
            for _ in range(5):
                ops.attn_bwd(q, k, v, o, do, lse, B, T, Hq, Hkv)
            torch.cuda.synchronize()
            ops.prof_begin()
            n = 30
            for _ in range(n):
                ops.attn_bwd(q, k, v, o, do, lse, B, T, Hq, Hkv)
            torch.cuda.synchronize()
            r = ops.prof_end()
            dkv, dq = r["attn_bwd_dkv"][0] / n * 1e3, r["attn_bwd_dq"][0] / n * 1e3
            n64 = (T + 63) // 64
            print(f"Hq={Hq:2d} Hkv={Hkv} B={B:3d} T={T:5d}: dK/dV {dkv:7.1f} us ({(Hq // Hkv) * n64:3d} tile visits in the heaviest "
                  f"workgroup, {((n64 + 1) // 2) * Hkv * B:4d} workgroups)   dQ {dq:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
