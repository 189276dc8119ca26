"""GPU measurement (not a pytest): known-byte launches for calibrating rocprofv3's FETCH_SIZE on the load paths the GEMMs
use.  Run as   rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o run -- python3 tests/calib_fetch.py
then   python3 scripts/fetch_calibration.py <dir> profiles/r04_fetch_calibration.json.
Every kernel below moves an exactly known number of bytes from HBM-resident buffers far larger than the 256 MiB Infinity Cache
working set of one launch (2 x 2 GiB, each launch sweeps its own fresh region where it can):
  dma_rate_kernel   mode 1: every workgroup stages a PRIVATE stream by LDS-DMA (buffer_load_dwordx4 ... lds): 256 x steps x 48 KiB
  stage_rate_kernel write 0: the shared-panel pattern through VGPRs (unique bytes = the panels; re-reads are L2 hits)
  torch copy        a plain 16-B/lane streaming read of N bytes (the guide's own calibration case)"""
import ctypes as C
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "libdma_rate.so"))
lib.dma_rate.restype = C.c_int
lib.dma_rate.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
nbytes = 2 << 30
A = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(3)
B = torch.empty(nbytes, dtype=torch.uint8, device=dev).fill_(5)
out = torch.zeros(256, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
expect = {}
# private streams: A gets 2/3 of a 48 KiB stage, B 1/3; 256 workgroups x steps: all distinct bytes while they fit the buffers
steps = 160  # 256 x 160 x 32 KiB = 1.25 GiB of A, 0.63 GiB of B: no wrap
for rep in range(3):
    assert lib.dma_rate(A.data_ptr(), B.data_ptr(), nbytes, nbytes, steps, 1, 12, 3, 48, out.data_ptr(), st) == 0
torch.cuda.synchronize()
expect["dma_rate_kernel<12, 3, 48>"] = {"bytes_per_launch": 256 * steps * 48 * 1024 + 256 * 2 * 48 * 1024, "launches": 3,
                                        "what": "LDS-DMA, private streams (every byte distinct; + the 2 prologue stages)"}
# shared panels (the GEMM's own pattern): unique bytes = 32 A panels + 64 B panels
steps2 = 400
for rep in range(3):
    assert lib.dma_rate(A.data_ptr(), B.data_ptr(), nbytes, nbytes, steps2, 0, 16, 3, 48, out.data_ptr(), st) == 0
torch.cuda.synchronize()
expect["dma_rate_kernel<16, 3, 48>"] = {"bytes_per_launch": (32 * 32 + 64 * 16) * 1024 * (steps2 + 2), "launches": 3,
                                        "what": "LDS-DMA, GEMM panel sharing: UNIQUE bytes (8 / 4 workgroups of an XCD share a panel; "
                                                "the fabric sees each panel once per XCD that uses it = once)"}
# plain streaming read + write
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
dst = torch.empty_like(src)
for rep in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
expect["copy"] = {"bytes_per_launch": 1 << 30, "launches": 3, "what": "torch copy_ of 1 GiB (elementwise kernel name varies)"}
json.dump(expect, open(os.environ.get("CALIB_OUT", "/tmp/calib_expect.json"), "w"), indent=1)
print(json.dumps(expect))
