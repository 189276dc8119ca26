"""GPU micro-benchmark (not a pytest): the LDS-DMA fill rate of a CU under the operand pattern of the 256x128x64 tile GEMM,
without any compute (tests/csrc/dma_rate.hip -> tests/libdma_rate.so, built by `make -C tests/csrc` or the hipcc line in
the source header).  Answers: is the ~61 GB/s per CU the GEMM loop stages at a property of the loop, or of the load path?"""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libdma_rate.so"))
lib.dma_rate.restype = C.c_int
lib.dma_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nbytes = 512 << 20
A = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=dev)
B = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=dev)
out = torch.zeros(256, dtype=torch.int64, device=dev)
steps = 400
print("mode 0 = GEMM sharing (8 workgroups per A panel, 4 per B panel inside an XCD); mode 1 = private streams (no L2 reuse)")
for mode in (0, 1):
    for nw, nst, kb in ((4, 3, 48), (8, 3, 48), (12, 3, 48), (16, 3, 48), (4, 6, 24), (8, 6, 24), (12, 6, 24), (4, 2, 48),
                        (8, 2, 48), (4, 12, 12), (12, 12, 12), (16, 9, 16)):
        ts = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.dma_rate(A.data_ptr(), B.data_ptr(), nbytes, nbytes, steps, mode, nw, nst, kb, out.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
            b.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
            ts.append(a.elapsed_time(b) * 1e3)
        t = sorted(ts)[1]
        ticks = out.cpu().double()
        per_cu = steps * kb * 1024 / (ticks.median().item() / 100e6) / 1e9
        print(f"mode {mode}: {nw:2d} waves, ring {nst:2d} x {kb} KiB: launch {t:7.1f} us, in-kernel median {ticks.median().item() / 100:7.1f} us "
              f"-> {per_cu:6.1f} GB/s per CU ({per_cu * 256 / 1e3:5.1f} TB/s), {steps * kb * 1024 / t / 1e3 * 256 / 1e3:5.1f} TB/s by events", flush=True)
