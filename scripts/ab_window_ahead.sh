# HF Trainer loop: window lookahead (SD_WINDOW_AHEAD auto/0) -- the next window's first teacher pass beside the optimizer step
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_lora.py tests/test_00_gpu_torchrun.py -x -q 2>&1 | tail -2
for i in 1 2 3; do
echo "window ahead (auto)"; bash scripts/ab_loop.sh --logging_nan_inf_filter true
echo "SD_WINDOW_AHEAD=0"; SD_WINDOW_AHEAD=0 bash scripts/ab_loop.sh --logging_nan_inf_filter true
done
