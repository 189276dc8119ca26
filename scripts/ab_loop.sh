# HF Trainer loop throughput (scripts/train.py, steady-state tokens_per_second) under host-side settings
cd $GRAFT_REPO_ROOT
run() {
  python scripts/train.py --random_init --synthetic_samples 1536 --equal_length --max_length 512 --logging_steps 8 --save_strategy no --warmup_steps 0 --num_train_epochs 1 --output_dir /tmp/sd_out --log_json /tmp/loop.json "$@" > /tmp/loop.log 2>&1
  python - "$*" <<PY
import json,sys
d=json.load(open("/tmp/loop.json"))
t=[h["tokens_per_second"] for h in d["log_history"] if "tokens_per_second" in h][2:]
print("%-70s median %.0f  max %.0f  n=%d" % (sys.argv[1] or "(defaults)", sorted(t)[len(t)//2], max(t), len(t)))
PY
}
if [ $# -gt 0 ]; then run "$@"; exit 0; fi
run
run --dataloader_num_workers 1
run --dataloader_num_workers 2
run
run --dataloader_num_workers 1
