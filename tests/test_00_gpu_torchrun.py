"""-m gpu: BASELINE config 3 in miniature -- the data-parallel Stage-2 loop exactly as a user launches it
(``torchrun ... scripts/train.py``, i.e. HF Trainer + accelerate under a multi-process launcher; reference call site
train.py:357-369,420, data parallelism HF trainer.py:1615-1626, no_sync :1757), on the real HIP kernels.

Two ranks share the one GPU of the test box (gloo as the process-group backend: RCCL needs one device per rank; the
driver's 8-GPU bench is the RCCL run) and are compared with ONE rank on the concatenated batch.  The batches have
equal numbers of target tokens on every rank, so the reference's mean-of-means (quirk Q4) equals the global token
mean and the two runs must log the same loss trajectory and end at the same parameters.

This file sorts first on purpose: the children are started before this pytest process has touched the GPU.
"""
import json
import os
import socket
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, per_device_batch, out, tag, extra=()):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", TOKENIZERS_PARALLELISM="false")
        if world == 1:
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k)
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "train.py"), "--tiny", "--random_init",
               "--synthetic_samples", "32", "--equal_length", "--max_length", "64", "--top_k", "16",
               "--per_device_train_batch_size", str(per_device_batch), "--gradient_accumulation_steps", "2",
               "--num_train_epochs", "1", "--logging_steps", "1", "--save_strategy", "no", "--learning_rate", "1e-3",
               "--warmup_steps", "0",
               "--output_dir", os.path.join(out, tag), "--log_json", os.path.join(out, tag + ".{rank}.json")]
        cmd += list(extra)
        if world > 1:
            cmd += ["--ddp_backend", "gloo"]
        log = open(os.path.join(out, f"{tag}.{rank}.log"), "w")
        procs.append((subprocess.Popen(cmd, env=env, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT), log))
    return procs


def test_two_rank_train_script_equals_one_rank_on_the_concatenated_batch():
    out = tempfile.mkdtemp()
    procs = _launch(2, 2, out, "dp2") + _launch(1, 4, out, "dp1")
    for p, log in procs:
        try:
            rc = p.wait(timeout=900)
        finally:
            if p.poll() is None:
                p.kill()
            log.close()
        assert rc == 0, open(log.name).read()[-4000:]
    r0, r1 = (json.load(open(os.path.join(out, f"dp2.{r}.json"))) for r in range(2))
    one = json.load(open(os.path.join(out, "dp1.0.json")))
    keep = os.path.join(ROOT, "gpurun_out")
    os.makedirs(keep, exist_ok=True)
    with open(os.path.join(keep, "torchrun2_vs_1.json"), "w") as f:
        json.dump({"dp2_rank0": r0, "dp2_rank1": r1, "dp1": one}, f)

    # the trainer put the HIP student under HipDataParallel (not torch DDP) on both ranks, 32 samples / (2 ranks x 2)
    # = 8 micro-batches = 4 optimizer steps per rank, communication only on the 4 last micro-batches
    for r in (r0, r1):
        assert r["wrapped"] == "HipDataParallel" and r["world_size"] == 2 and r["global_step"] == 4
        assert r["optimizer"] == "FlatAdamW"  # DistillationTrainer.create_optimizer: one fused launch, not 310 tensors
        assert r["reducer"] == {"backwards": 8, "synced": 4}
    assert one["wrapped"] == "HipDataParallel" and one["reducer"] is None and one["global_step"] == 4
    # both ranks: the same logged losses (HF gathers and averages them) and bitwise-identical parameters at the end
    loss0 = [e["loss"] for e in r0["log_history"] if "loss" in e]
    loss1 = [e["loss"] for e in r1["log_history"] if "loss" in e]
    assert loss0 == loss1 and len(loss0) == 4
    assert r0["param_checksum"] == r1["param_checksum"]
    # == one rank on the concatenated batch (bf16 gradients averaged across ranks vs summed in one backward:
    # differences are bf16 rounding of the gradient buffers, 4e-3 per element; the losses agree far better)
    loss_one = [e["loss"] for e in one["log_history"] if "loss" in e]
    assert len(loss_one) == 4
    for a, b in zip(loss_one, loss0):
        assert abs(a - b) <= 5e-3 * abs(a), (loss_one, loss0)
    gn_one = [e["grad_norm"] for e in one["log_history"] if "grad_norm" in e]
    gn_two = [e["grad_norm"] for e in r0["log_history"] if "grad_norm" in e]
    for a, b in zip(gn_one, gn_two):
        assert abs(a - b) <= 3e-2 * abs(a), (gn_one, gn_two)
    assert loss0[-1] < loss0[0]


def test_two_rank_lora_train_script_equals_one_rank():
    """The same comparison with ``--use_lora`` (train.py:180-202): the data-parallel wrapper all-reduces the flat weight
    gradient as always, every rank projects the SAME averaged dW onto its adapter (no atomics in sd_lora.hip), so the
    adapters stay bitwise identical across ranks; only the adapter and the two saved modules train."""
    out = tempfile.mkdtemp()
    lora = ["--use_lora", "--lora_r", "8", "--lora_alpha", "16", "--init_lora_weights", "pissa"]
    procs = _launch(2, 2, out, "lora2", lora) + _launch(1, 4, out, "lora1", lora)
    for p, log in procs:
        try:
            rc = p.wait(timeout=900)
        finally:
            if p.poll() is None:
                p.kill()
            log.close()
        assert rc == 0, open(log.name).read()[-4000:]
    r0, r1 = (json.load(open(os.path.join(out, f"lora2.{r}.json"))) for r in range(2))
    one = json.load(open(os.path.join(out, "lora1.0.json")))
    with open(os.path.join(ROOT, "gpurun_out", "torchrun2_vs_1_lora.json"), "w") as f:
        json.dump({"dp2_rank0": r0, "dp2_rank1": r1, "dp1": one}, f)
    for r in (r0, r1, one):
        assert r["optimizer"] == "FlatAdamW" and r["global_step"] == 4
        assert r["trainable"] == ["embed_tokens", "lm_head", "lora_A", "lora_B"]
    assert r0["reducer"] == {"backwards": 8, "synced": 4}
    assert r0["adapter_checksum"] == r1["adapter_checksum"] and r0["param_checksum"] == r1["param_checksum"]
    loss0 = [e["loss"] for e in r0["log_history"] if "loss" in e]
    loss_one = [e["loss"] for e in one["log_history"] if "loss" in e]
    assert len(loss0) == 4 and len(loss_one) == 4
    for a, b in zip(loss_one, loss0):
        assert abs(a - b) <= 5e-3 * abs(a), (loss_one, loss0)
    assert abs(one["adapter_checksum"] - r0["adapter_checksum"]) <= 2e-3 * one["adapter_checksum"]
    assert loss0[-1] < loss0[0]
