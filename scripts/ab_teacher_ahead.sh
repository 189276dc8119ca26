# HF Trainer loop: EXPERIMENT SD_TEACHER_AHEAD=1 (teacher pass of micro-batch i+1 enqueued at the end of training_step(i))
cd $GRAFT_REPO_ROOT
SD_TEACHER_AHEAD=1 timeout -k 10 400 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_model.py -x -q -k "trainer or c1" 2>&1 | tail -2
for i in 1 2; do
echo "default"; bash scripts/ab_loop.sh --logging_nan_inf_filter true
echo "teacher ahead"; SD_TEACHER_AHEAD=1 bash scripts/ab_loop.sh --logging_nan_inf_filter true
done
echo "default, nan filter off"; bash scripts/ab_loop.sh --logging_nan_inf_filter false
echo "teacher ahead, nan filter off"; SD_TEACHER_AHEAD=1 bash scripts/ab_loop.sh --logging_nan_inf_filter false
