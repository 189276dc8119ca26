# in-process A/Bs (tests/bench_knob_ab.py): which forward is enqueued first
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "teacher first"; python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 4 --block 8 2>/dev/null
echo "student first"; python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 4 --block 8 --student-first 2>/dev/null
done
