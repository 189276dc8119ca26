# in-process A/Bs (tests/bench_knob_ab.py)
cd $GRAFT_REPO_ROOT
echo "3 = adopted; 35 = student 128 rows x 2 stages; 65 = student 128/3, teacher 128 rows x 2 stages; 97 = both 2 stages"
python tests/bench_knob_ab.py gemm.fwd_bump 3 35 65 97 --rounds 8 --block 8 2>/dev/null
