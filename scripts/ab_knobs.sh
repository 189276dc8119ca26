# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
echo "dX 64 -> 128 rows under the default budget (3 = adopted forward bits, 7 = + dX bit)"
python tests/bench_knob_ab.py gemm.fwd_bump 3 7 --rounds 14 --block 8 2>/dev/null
