# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
python tests/bench_knob_ab.py gemm.splitk_min_kt 96 64 48 32 --rounds 8 --block 8 2>/dev/null
python tests/bench_knob_ab.py gemm.splitk_min_slice 24 16 12 32 --rounds 6 --block 8 2>/dev/null
