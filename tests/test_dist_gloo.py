'''world_size-2 gloo test of the seed-sharding path (the N > 1 data path of bench.py):
noise drawn once on the host and sliced per rank, per-rank processing, one all-gather;
the gathered result must equal the single-process result.  CPU only.'''
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _process(x):           # stands in for the (sample-independent) denoising of a shard
    return torch.tanh(x) * 2.0 + x.flatten(1).sum(dim=1).view(-1, 1, 1, 1)


def _worker(rank, world, port, per_rank, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from flexdiffuse_amd import dist as fdist
    r, ws, _ = fdist.init('gloo')
    assert (r, ws) == (rank, world)
    noise = fdist.global_noise(world * per_rank, (4, 8, 8), seed=1337)
    prompts = [f'prompt {i}' for i in range(world * per_rank)]
    mine = noise[fdist.shard_range(rank, world, per_rank)]
    assert fdist.shard(prompts, rank, world, per_rank) == prompts[rank * per_rank:(rank + 1) * per_rank]
    gathered = fdist.all_gather_samples(_process(mine))
    if rank == 0:
        torch.save(gathered, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded(tmp_path):
    from flexdiffuse_amd import dist as fdist
    world, per_rank = 2, 3
    out = str(tmp_path / 'gathered.pt')
    mp.spawn(_worker, args=(world, _free_port(), per_rank, out), nprocs=world, join=True)
    gathered = torch.load(out)
    full = _process(fdist.global_noise(world * per_rank, (4, 8, 8), seed=1337))
    assert torch.equal(gathered, full)


def test_single_process_is_identity():
    from flexdiffuse_amd import dist as fdist
    x = torch.arange(12.0).view(3, 4)
    assert fdist.all_gather_samples(x) is x
    assert fdist.shard_range(2, 8, 8) == slice(16, 24)
