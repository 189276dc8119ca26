"""GPU diagnostic (not a pytest): where a K-step of gemm_pstag_kernel spends its cycles.  Needs the stamp build:
    make -C speech_distill_amd/csrc stamps && SD_HIP_LIB=/tmp/sd_stamps/libsd_hip.so python tests/bench_stamps.py
Prints, for one compute wave of each half and one producer wave of workgroup 0, the s_memtime deltas between the phase
boundaries of K-steps 4..19 of its first tile (median over the steps), in shader cycles."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_distill_amd import ops  # noqa: E402
from speech_distill_amd._lib import load_lib  # noqa: E402

dev = torch.device("cuda:0")


def main():
    lib = load_lib()
    lib.sd_debug_stamp_buffer.argtypes = [C.c_void_p]
    lib.sd_debug_stamp_buffer.restype = None
    buf = torch.zeros(12 * 96, dtype=torch.int64, device=dev)
    lib.sd_debug_stamp_buffer(buf.data_ptr())
    for name, M, N, K in (("teacher gate|up", 2048, 12288, 2048), ("student lm_head rows", 1536, 32768, 1024)):
        a = torch.randn(M, K, device=dev).bfloat16()
        b = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        os.environ["SD_GEMM_NO_P256"] = "1"
        for _ in range(3):
            ops.gemm(a, b)
        torch.cuda.synchronize()
        st = buf.cpu().view(12, 16, 6).tolist()
        print(f"== {name} M={M} N={N} K={K}")
        for w, role in ((0, "compute half 0"), (4, "compute half 1"), (8, "producer 0"), (11, "producer 3")):
            rows = st[w]
            period = [rows[g + 1][0] - rows[g][0] for g in range(15)]
            npt = 6 if w < 8 else 5
            seg = [[rows[g][p + 1] - rows[g][p] for g in range(16)] for p in range(npt - 1)]
            names = (["issue+reads issued", "vmcnt+lgkm wait", "barrier 1", "MFMA phase", "barrier 2"] if w < 8 else
                     ["issue 8 pieces", "vmcnt wait", "barrier 1", "barrier 2"])
            print(f"  wave {w:2d} ({role}): step period median {statistics.median(period):7.0f} cycles")
            for nm, sg in zip(names, seg):
                print(f"      {nm:22s} median {statistics.median(sg):7.0f}  min {min(sg):6d}  max {max(sg):6d}")


if __name__ == "__main__":
    main()
