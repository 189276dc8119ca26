"""2-rank rehearsal of the data-parallel path on ONE GPU (gloo transport, both ranks on cuda:0):
  torchrun --nproc-per-node 2 --master-addr 127.0.0.1 tests/ddp_rehearsal.py
Checks, with real kernels and the real stage callback: (1) averaged gradients are identical on both ranks,
(2) they equal the average of the two ranks' local gradients, (3) the split tied-embedding exchange gives the
same result as reducing the whole embedding bucket at the end, (4) no_sync keeps gradients local."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import speech_distill_amd as sda  # noqa: E402
from speech_distill_amd import ddp  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    dims = sda.Qwen3Dims(1000, 256, 512, 3, 4, 2)
    g = torch.Generator().manual_seed(100 + rank)
    ids = torch.randint(0, 1000, (2, 40), generator=g).to(dev)
    ids[0, :10] = 7  # duplicates, also across ranks
    probe = torch.randn(2, 40, 1000, generator=g).to(dev)

    def run(split, sync=True):
        m = sda.HipQwen3ForCausalLM(dims, device=dev, seed=3)
        red = ddp.attach(m, split_embedding=split)
        if sync:
            (m(input_ids=ids).logits.float() * probe).sum().backward()
        else:
            with m.no_sync():
                (m(input_ids=ids).logits.float() * probe).sum().backward()
        torch.cuda.synchronize()
        return m.flat_grad.float().cpu(), red

    local, red = run(False, sync=False)
    assert not red.issued
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    whole, _ = run(False)
    split, red = run(True)
    assert any(s == ddp.STAGE_EMBED for s, _, _ in red.issued)
    for name, got in (("whole", whole), ("split", split)):
        other = [torch.empty_like(got) for _ in range(world)]
        dist.all_gather(other, got)
        assert torch.equal(other[0], other[1]), f"{name}: ranks disagree"
        err = float((got - want).abs().max() / want.abs().max())
        print(f"rank {rank} {name}: max rel err vs mean of local grads {err:.3e}", flush=True)
        assert err < 2e-2, (name, err)
    dist.barrier()
    if rank == 0:
        print("DDP REHEARSAL OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
