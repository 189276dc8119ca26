# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
python tests/bench_knob_ab.py gemm.splitk_max 0 2 3 --rounds 10 --block 6 2>/dev/null
