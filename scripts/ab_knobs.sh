# in-process A/Bs (tests/bench_knob_ab.py): stream priorities (the knob itself is a no-op pair, the process-level setting differs)
cd $GRAFT_REPO_ROOT
export SD_DEBUG="gemm.fwd_bump=3"
echo base; python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 2>/dev/null
echo main_high; python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 --main-priority -1 2>/dev/null
echo teacher_high; SD_STREAM_PRIORITY_TEACHER=-1 python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 2>/dev/null
echo dw_high; SD_STREAM_PRIORITY_DW=-1 python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 2>/dev/null
echo main_teacher_high; SD_STREAM_PRIORITY_TEACHER=-1 python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 --main-priority -1 2>/dev/null
echo base; python tests/bench_knob_ab.py gemm.no_table 0 0 --rounds 6 --block 8 2>/dev/null
