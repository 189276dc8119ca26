# in-process A/Bs (tests/bench_knob_ab.py): the shared-GPU defaults (now shape-conditional) vs both off, over token counts
cd $GRAFT_REPO_ROOT
for shape in "1 512" "2 512" "3 512" "4 512" "6 512" "8 512" "4 2048"; do
  set -- $shape
  echo "B=$1 T=$2: cu_budget -1 + fwd_bump -1 (both off, first value) vs defaults (second)"
  SD_DEBUG="gemm.fwd_bump=-1" python tests/bench_knob_ab.py gemm.cu_budget -1 -1 --batch $1 --seq-len $2 --rounds 3 --block 5 2>/dev/null
  python tests/bench_knob_ab.py gemm.cu_budget 0 0 --batch $1 --seq-len $2 --rounds 3 --block 5 2>/dev/null
done
