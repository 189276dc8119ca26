# in-process A/Bs (tests/bench_knob_ab.py): 256x256 tiles for the (unfolded) teacher's gate|up beside the student
cd $GRAFT_REPO_ROOT
python tests/bench_knob_ab.py gemm.p256_min_tiles 1024 300 --no-fold --rounds 8 --block 8 2>/dev/null
