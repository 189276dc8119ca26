#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): in-process A/B of one sd_debug knob on the distillation micro-step
# (tests/bench_knob_ab.py alternates the values in blocks inside ONE process: same box, same clocks).
#   scripts/ab_knobs.sh gemm.fwd_bump -1 0 [--batch 4 --seq-len 512 --rounds 8 --block 8 ...]
# The round-4 table of DESIGN.md section 8 came from calls of this form, e.g.
#   gemm.fwd_bump -1 0 | 0 3 7 11 15 | 3 19 18 | 3 35 65 97        gemm.cu_budget -1 0 | 0 224 192 176 160 128
#   gemm.p256_min_tiles 1024 100 (and 1024 300 --no-fold)            model.shared_layers -1 24 20 16 12 8 0
#   model.overlap_mask 31 23 30 27   attn.variant 0 2   gemm.group_m 0 2 4 8 16   gemm.no_persist 0 1
#   gemm.persist_balance 0 1   gemm.splitk_max 0 2 3   gemm.splitk_min_kt 96 64 48 32   gemm.no_table 0 1
# and, for the shape conditions, each of  --batch {1,2,3,4,6,8} --seq-len 512 | --batch 2 --seq-len 1024 | --batch 4 --seq-len 2048.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
exec python tests/bench_knob_ab.py "$@" 2>/dev/null
