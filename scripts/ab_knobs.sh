# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
python tests/bench_knob_ab.py gemm.persist_balance 0 1 --rounds 20 --block 6 2>/dev/null
