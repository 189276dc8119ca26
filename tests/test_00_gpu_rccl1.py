"""-m gpu: the RCCL calls of BASELINE config 3, driven on the one GPU of the test box.

``bench.py --force-dist`` runs the N > 1 code path with a ONE-rank "nccl" (= RCCL) process group: the process group
itself, the side streams picked under it, the row-count exchange, every bucket's collective on the communication stream
(C2-sized buckets: 31.5 MB per layer, 327 MB for the tied head), the embedding-row all-gather and the compute stream's
final wait -- the call sequence the driver's 8-GPU run executes (reference: implicit torch DDP, train.py:357-369,420;
HF trainer.py:1615-1626,1757), with RCCL doing one-rank copies instead of xGMI transfers.  Asserted from the JSON line:
exit code 0, every rank holds the same gradient (trivially one here), every element of the flat gradient reduced exactly
once, both collective forms (all_reduce | reduce_scatter + all_gather), and the loss of the ordinary single-process step.

Sorts before the other GPU tests on purpose: the children run before this pytest process has touched the GPU.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-prof",
           *extra]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_single_rank_rccl_rehearsal_of_the_data_parallel_step():
    plain = _bench(port=29611)
    assert plain["comm"] is None and plain["grad_sync_ok"] is None
    keep = {"plain_ms": plain["ms_per_step"]}
    for algo in ("allreduce", "rs_ag"):
        r = _bench("--force-dist", "--comm-algo", algo, port=29612 if algo == "allreduce" else 29613)
        c = r["comm"]
        assert r["grad_sync_ok"] is True
        assert c["single_rank_rehearsal"] and c["algo"] == algo and c["every_gradient_element_reduced_once"] is True
        # final norm + dense tied head + 28 layers + the embedding-row exchange; all three backwards of the run communicated
        stages = [b["stage"] for b in c["buckets"]]
        assert stages == [-1, -1] + list(range(27, -1, -1)) + [-2], stages
        assert abs(c["bytes_per_step"] - 2 * 603_7e5) < 0.2e9 and all(b["ms"] >= 0 for b in c["buckets"])
        assert c["stats"]["synced"] == c["stats"]["backwards"] > 0
        assert c["exposed_ms_per_step"] is not None and c["rccl_version"][0].isdigit()
        # same arithmetic as the single-process step: AVG over one rank is the identity
        for k in ("total", "task", "distill", "teacher"):
            assert abs(r["loss"][k] - plain["loss"][k]) <= 1e-6 * abs(plain["loss"][k]) + 1e-9, (k, r["loss"], plain["loss"])
        keep[algo] = {"ms_per_step": r["ms_per_step"], "comm": c}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "rccl1_rehearsal.json"), "w") as f:
        json.dump(keep, f)
