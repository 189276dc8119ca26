"""world_size-2 gloo test (CPU) of the data-parallel path: bucket plan + stage-driven all-reduce of the
flat gradient buffer, no_sync on accumulation micro-batches, mean-of-means semantics (quirk Q4)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from speech_distill_amd import ddp


def _layout(L=3, h=16, big=100):
    embed = (0, big)
    ranges, off = [], big
    for _ in range(L):
        ranges.append((off, off + 5 * h))
        off += 5 * h
    norm = (off, off + h)
    return ranges, embed, norm, off + h


def test_bucket_plan_covers_every_element_once_in_backward_order():
    ranges, embed, norm, numel = _layout()
    for lpb in (1, 2, 5):
        plan = ddp.bucket_plan(ranges, embed, norm, numel, lpb)
        cover = torch.zeros(numel, dtype=torch.int32)
        for _, a, b in plan:
            cover[a:b] += 1
        assert bool((cover == 1).all())
        stages = [s for s, _, _ in plan]
        assert stages[0] == ddp.STAGE_HEAD and stages[-1] == ddp.STAGE_EMBED
        layer_stages = [s for s in stages if s >= 0]
        assert layer_stages == sorted(layer_stages, reverse=True)
    # split tied embedding: its dense part closes with the head stage (start of backward), nothing waits for the end
    plan = ddp.bucket_plan(ranges, embed, norm, numel, 1, split_embedding=True)
    cover = torch.zeros(numel, dtype=torch.int32)
    for _, a, b in plan:
        cover[a:b] += 1
    assert bool((cover == 1).all()) and all(s != ddp.STAGE_EMBED for s, _, _ in plan)
    assert (ddp.STAGE_HEAD, embed[0], embed[1]) in plan


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ranges, embed, norm, numel = _layout()
    flat = torch.zeros(numel)
    red = ddp.FlatGradAllReduce(lambda: flat, ddp.bucket_plan(ranges, embed, norm, numel))

    def backward(scale):  # what the C runner does: fill a bucket, report its stage
        red.begin_step()
        for stage in [ddp.STAGE_HEAD] + list(range(len(ranges) - 1, -1, -1)) + [ddp.STAGE_EMBED]:
            for st, a, b in red.plan:
                if st == stage:
                    flat[a:b] += scale * (rank + 1) * torch.arange(a, b, dtype=torch.float32)
            red.on_stage(stage)
        red.finish()

    with red.no_sync():  # accumulation micro-batch: local only
        backward(1.0)
    assert not red.issued
    local = flat.clone()
    backward(1.0)  # last micro-batch: everything accumulated so far is averaged
    assert len(red.issued) == len(red.plan)
    want = 2.0 * (sum(r + 1 for r in range(world)) / world) * torch.arange(numel, dtype=torch.float32)
    ok = torch.allclose(flat, want, rtol=1e-6) and torch.allclose(local, (rank + 1) * torch.arange(numel, dtype=torch.float32))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_of_flat_gradients():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
