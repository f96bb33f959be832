'''Seed / sample sharding across the GPUs of one node.

Samples are independent given (embeddings, initial noise): the denoising loop has no
cross-sample op (reference pipeline/flex.py:262-287; the reference only ever iterates
samples sequentially, utils.py:90).  So each rank (one process per GPU) takes a contiguous
slice of the global batch, weights are replicated, the initial noise is drawn ONCE from a
host generator and sliced (results do not depend on the world size), and the only
collective is one all-gather of the final latents / decoded images (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU in the tests).
'''
from __future__ import annotations

import os
from typing import Sequence, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int, int]:
    '''(rank, world_size, local_rank) from the torchrun environment.'''
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend: str = 'nccl'):
    '''Join the process group described by the torchrun environment (no-op for one rank unless
    FD_FORCE_DIST=1).  FD_DIST_BACKEND overrides the backend (e.g. gloo when several ranks share
    one GPU in a plumbing test: RCCL refuses duplicate devices).'''
    backend = os.environ.get('FD_DIST_BACKEND', backend)
    rank, ws, local = world()
    # FD_FORCE_DIST=1 initialises the process group (and runs the collectives through RCCL)
    # even for a single rank: exercises the N > 1 code path on a 1-GPU box
    if (ws > 1 or os.environ.get('FD_FORCE_DIST')) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=ws, **kw)
    return rank, ws, local


def shard_range(rank: int, world_size: int, per_rank: int) -> slice:
    '''Contiguous sample indices of `rank`: [rank*B, (rank+1)*B).'''
    return slice(rank * per_rank, (rank + 1) * per_rank)


def global_noise(total: int, shape: Sequence[int], seed: int) -> torch.Tensor:
    '''The whole global batch of initial latents from ONE host generator (fp32, CPU).'''
    return torch.randn((total,) + tuple(shape), generator=torch.Generator('cpu').manual_seed(seed),
                       dtype=torch.float32)


def global_img2img_noise(total: int, shape: Sequence[int], seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    '''img2img draws of the whole global batch in the pipeline's order from ONE host generator
    (reference pipeline/flex.py:189-214): the VAE posterior sample (1, *shape) -- shared by every
    sample, the reference repeats the encoded latent -- then the add_noise draw (total, *shape).
    Every rank calls this with the same seed and passes its rows of the second tensor to
    `FlexPipeline.__call__(noise=...)` together with a generator seeded with `seed` (whose first draw
    is then the same posterior sample on every rank): images do not depend on the world size.'''
    g = torch.Generator('cpu').manual_seed(seed)
    posterior = torch.randn((1,) + tuple(shape), generator=g, dtype=torch.float32)
    noise = torch.randn((total,) + tuple(shape), generator=g, dtype=torch.float32)
    return posterior, noise


def shard(items, rank: int, world_size: int, per_rank: int):
    return items[shard_range(rank, world_size, per_rank)]


def all_gather_samples(x: torch.Tensor) -> torch.Tensor:
    '''Concatenate every rank's (B, ...) tensor along dim 0 in rank order (identity when
    not distributed).'''
    if not (dist.is_available() and dist.is_initialized()) or \
            (dist.get_world_size() == 1 and not os.environ.get('FD_FORCE_DIST')):
        return x
    x = x.contiguous()
    if x.is_cuda and dist.get_backend() == 'gloo':
        # gloo moves host memory: stage through the CPU (test plumbing only; RCCL takes the
        # device tensors directly over xGMI)
        return all_gather_samples(x.cpu()).to(x.device)
    out = torch.empty((dist.get_world_size() * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype,
                      device=x.device)
    dist.all_gather_into_tensor(out, x)
    return out
