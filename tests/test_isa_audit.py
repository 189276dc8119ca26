"""Static audit of the generated gfx950 ISA (CPU only, needs hipcc): the transposed LDS reads are issued as inline
asm that hipcc does not track (sd_common.cuh), so nothing but our own `s_waitcnt lgkmcnt(0)` orders their data.
This test compiles the two kernel files to assembly and checks that no instruction reads or overwrites a
ds_read_b64_tr_b16 destination register before the next lgkmcnt(0) wait, and that no main loop drains the LDS-DMA
prefetch with a compiler-inserted vmcnt(0) in front of those reads."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["sd_gemm.hip", "sd_attn.hip"])
def test_asm_transposed_reads_are_waited_before_use(src):
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, "k.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                    "--cuda-device-only", "-S", os.path.join(ROOT, "speech_distill_amd", "csrc", src), "-o", out],
                   check=True)
    assert "-amdgpu-mfma-vgpr-form=1" in open(os.path.join(ROOT, "speech_distill_amd", "csrc", "Makefile")).read()
    pending, n_tr, hazards, n_acc, n_mfma = {}, 0, [], 0, 0
    for ln, line in enumerate(open(out)):
        t = line.strip()
        n_acc += t.startswith("v_accvgpr")
        n_mfma += t.startswith("v_mfma")
        if not t or t.startswith((";", ".")):
            continue
        if t.endswith(":"):
            if not t.startswith(".LBB"):
                pending = {}
            continue
        op = t.split()[0]
        if op == "ds_read_b64_tr_b16":
            m = re.search(r"v\[(\d+):(\d+)\]", t)
            n_tr += 1
            for r in range(int(m.group(1)), int(m.group(2)) + 1):
                pending[r] = ln
            continue
        if op == "s_waitcnt" and "lgkmcnt(0)" in t:
            pending = {}
            continue
        if pending and not op.startswith("ds_read"):
            regs = set()
            for m in re.finditer(r"v\[(\d+):(\d+)\]", t):
                regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            regs.update(int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", t))
            if regs & set(pending):
                hazards.append((ln, t))
    # sanity: the asm reads are really there (sd_attn.hip: 3 kernels x 32 after the round-2 backward rewrite = 96)
    assert n_tr > (100 if src == "sd_gemm.hip" else 60), "expected the asm transposed reads in the kernels"
    # accumulators stay in VGPRs: no AGPR <-> VGPR copies around the VALU work (only the dK/dV kernel, whose two
    # accumulator sets exceed 256 registers, keeps a few)
    assert n_acc <= (0 if src == "sd_gemm.hip" else n_mfma * 2), (n_acc, n_mfma)
    assert not hazards, hazards[:5]
