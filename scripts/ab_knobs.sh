# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
python tests/bench_knob_ab.py attn.variant 0 2 --rounds 6 --block 8 2>/dev/null
python tests/bench_knob_ab.py gemm.group_m 0 2 4 8 16 --rounds 5 --block 8 2>/dev/null
python tests/bench_knob_ab.py gemm.splitk_min_kt 96 48 100000 --rounds 5 --block 8 2>/dev/null
python tests/bench_knob_ab.py qk_bwd.blocks 512 256 1024 --rounds 5 --block 8 2>/dev/null
python tests/bench_knob_ab.py gemm.no_table 0 1 --rounds 6 --block 8 2>/dev/null
