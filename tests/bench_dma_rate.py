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

# ---- round 4: LDS-DMA A-panel stream + concurrent buffer_load_dwordx4 -> VGPR stream of the B fragments (mix_rate_kernel)
lib.mix_rate.restype = C.c_int
lib.mix_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_void_p]
sink = torch.zeros(256 * 64, dtype=torch.int32, device=dev)
print("\nmixed path: A panel by LDS-DMA (producer waves) + B fragments by buffer_load_dwordx4 to VGPRs (compute waves), per 64-deep K-step")
print("  tile MxN, wave grid -> bytes per step through the CU's vector memory path = A_KB (DMA) + ncomp * brows * 128 B (VGPR)")
steps = 400
cases = [
    # nprod ncomp nst a_kb brows wn_groups n_cols  label
    (4, 8, 3, 32, 64, 2, 128, "256x128, waves 4x2 (64x64 each): B fetched 4x"),
    (4, 8, 3, 32, 32, 4, 128, "256x128, waves 2x4 (128x32 each): B fetched 2x"),
    (4, 8, 3, 32, 16, 8, 128, "256x128, waves 1x8 (256x16 each): B fetched 1x"),
    (0, 8, 3, 0, 64, 2, 128, "VGPR stream alone, 4x2"),
    (0, 8, 3, 0, 32, 4, 128, "VGPR stream alone, 2x4"),
    (0, 8, 3, 0, 16, 8, 128, "VGPR stream alone, 1x8"),
    (4, 8, 3, 48, 16, 8, 128, "today's 48 KiB DMA stage + 16 KiB VGPR"),
    (4, 4, 3, 32, 128, 2, 256, "256x256, 4 compute waves 2x2 (128x128 each): B fetched 2x"),
    (4, 4, 3, 32, 64, 2, 128, "256x128, 4 compute waves 2x2 (128x64 each): B fetched 2x"),
    (8, 8, 3, 32, 64, 2, 128, "256x128, 8 producers, waves 4x2"),
    (4, 8, 5, 16, 64, 2, 128, "128x128, waves 4x2 of 32x64: A 16 KiB by DMA"),
]
for nprod, ncomp, nst, a_kb, brows, wng, ncols, label in cases:
    ts = []
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.mix_rate(A.data_ptr(), B.data_ptr(), nbytes, nbytes, steps, nprod, ncomp, nst, a_kb, brows, wng, ncols,
                          out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        b.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        ts.append(a.elapsed_time(b) * 1e3)
    t = sorted(ts)[1]
    tk = out.cpu().double().median().item() / 100  # us in-kernel
    dma_b = a_kb * 1024
    vg_b = ncomp * brows * 128
    uniq = a_kb * 1024 + ncols * 128
    us_step = tk / steps
    print(f"{label:62s}: {us_step * 1e3:6.0f} ns/step, DMA {dma_b / us_step / 1e3:5.1f} + VGPR {vg_b / us_step / 1e3:5.1f} = "
          f"{(dma_b + vg_b) / us_step / 1e3:6.1f} GB/s per CU; unique operand bytes {uniq / us_step / 1e3:5.1f} GB/s per CU "
          f"(launch {t:7.1f} us)", flush=True)

# ---- round 4, third arm: the same panel stream staged through registers (coalesced buffer_load_dwordx4 -> ds_write_b128)
lib.stage_rate.restype = C.c_int
lib.stage_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long] + [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_void_p]
print("\nregister-staged fill: coalesced buffer_load_dwordx4 -> VGPR (-> ds_write_b128 into a 2-stage LDS ring when write = 1)")
for write in (0, 1):
    for nw, depth, kb in ((4, 1, 48), (4, 2, 48), (8, 1, 48), (8, 2, 48), (12, 2, 48), (16, 2, 48), (4, 2, 24), (8, 2, 24),
                          (4, 1, 64), (4, 2, 64), (8, 2, 64)):
        ts = []
        for rep in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = lib.stage_rate(A.data_ptr(), B.data_ptr(), nbytes, nbytes, steps, write, nw, depth, kb, out.data_ptr(),
                                sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
            b.record()
            torch.cuda.synchronize()
            assert rc == 0, rc
            ts.append(a.elapsed_time(b) * 1e3)
        t = sorted(ts)[1]
        tk = out.cpu().double().median().item() / 100
        per_cu = steps * kb * 1024 / tk / 1e3
        print(f"write {write}: {nw:2d} waves, {depth} step(s) of loads in flight, stage {kb} KiB: in-kernel median {tk:7.1f} us -> "
              f"{per_cu:6.1f} GB/s per CU ({per_cu * 256 / 1e3:5.1f} TB/s), launch {t:7.1f} us", flush=True)
