'''N > 1 data path on the GPU box: the product's RCCL all-gather and bench.py's own
rank launcher, each in child processes (a process that has initialised the GPU must not be
re-used as a launcher).  Needs an MI355X.'''
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RCCL_CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ['FD_ROOT'])
import torch.distributed as dist
from flexdiffuse_amd import Guide, SimpleGuide, build, dist as fdist
from flexdiffuse_amd.encode.clip import CLIPEncoder
rank, world, local = fdist.init('nccl')
assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
dev = torch.device('cuda', local)
sds = build.synthetic_state_dicts('mini', seed=0)
pipe, clip, tok = build.build_models(sds, 'mini', dev)
enc = CLIPEncoder(clip, tok)
prompts = fdist.shard(['a photo of a turtle', 'zeus, oil painting'], rank, world, 2)
noise = fdist.global_noise(2, (4, 8, 8), 1337)[fdist.shard_range(rank, world, 2)].to(dev)
embeds = Guide(clip, tok, device='cuda').embeds(prompt=prompts)
pipe(guide=SimpleGuide(enc, pipe.unet, 8.0, 3, embeds), init_size=(64, 64), latents=noise, output_type='np')
lat = fdist.all_gather_samples(pipe.last_latents)      # RCCL all_gather_into_tensor, world 1
img = fdist.all_gather_samples(pipe.last_images)
assert lat.data_ptr() != pipe.last_latents.data_ptr(), 'the collective did not run'
assert torch.equal(lat, pipe.last_latents) and torch.equal(img, pipe.last_images)
dist.barrier(); dist.destroy_process_group()
print('RCCL_OK', tuple(lat.shape), tuple(img.shape))
'''


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(FD_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', **kw)
    return env


def test_rccl_all_gather_of_a_pipeline_pass_world1():
    '''FD_FORCE_DIST=1: the process group is initialised with backend nccl (= RCCL) for a single
    rank, one mini pipeline pass runs, and its latents / images go through the product
    all_gather_samples on the device.'''
    r = subprocess.run([sys.executable, '-c', _RCCL_CHILD], env=_env(FD_FORCE_DIST='1', MASTER_PORT='29531'),
                       capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b'RCCL_OK (2, 4, 8, 8)' in r.stdout


def _bench(args, **env):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=_env(**env),
                       capture_output=True, timeout=1500)
    assert r.returncode == 0, (r.stdout.decode()[-500:], r.stderr.decode()[-2000:])
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'bench.py must print exactly ONE JSON line'
    return json.loads(lines[0])


MINI = ['--preset', 'mini', '--size', '64', '--ddim-steps', '3', '--batch', '2', '--steps', '1',
        '--warmup', '1', '--no-cpu-baseline', '--no-parity']


def test_bench_launches_two_ranks():
    '''`python bench.py --gpus 2` starts two ranks itself (no torchrun) and rank 0 reports
    n_gpus 2.  The box has ONE GPU, so both ranks share it (FD_BENCH_SHARE_GPU=1) and the
    collective runs over gloo (RCCL refuses two ranks on one device); what is under test is the
    launcher, the sharding and the gather of 2 x batch samples.'''
    line = _bench(['--gpus', '2'] + MINI, FD_BENCH_SHARE_GPU='1', FD_DIST_BACKEND='gloo')
    assert line['n_gpus'] == 2
    # rccl_ranks counts ranks of an RCCL ("nccl") process group only: this run's collective went
    # over gloo, so RCCL saw no rank and the line must say so
    assert line['rccl_ranks'] == 0 and line['dist_backend'] == 'gloo'
    assert line['config']['images_per_step'] == 4 and line['scaling'] == 'weak'
    assert line['all_gather']['bytes_per_rank'] > 0 and line['value'] > 0


def test_bench_launches_eight_ranks():
    '''The 8-rank shape of the driver's scaling run, on the box's ONE GPU: `bench.py --gpus 8`
    starts eight fresh children from a parent that never touches the GPU; each rank builds its own
    model, runs its shard through its own launch plan and takes part in the gather (gloo: RCCL
    refuses eight ranks on one device).  Under test: eight concurrent host loops, the launcher, the
    sharding and the gather of 8 x batch samples.'''
    line = _bench(['--gpus', '8', '--launch', 'plan'] + MINI, FD_BENCH_SHARE_GPU='1', FD_DIST_BACKEND='gloo')
    assert line['n_gpus'] == 8 and line['rccl_ranks'] == 0 and line['dist_backend'] == 'gloo'
    assert line['config']['images_per_step'] == 16 and line['value'] > 0
    assert line['host_margin']['launch'] == 'plan' and line['host_margin']['launches_per_forward'] > 50
    # and the bench's default launch mode (HIP-graph replay; capture per rank) in the same 8-rank shape
    line = _bench(['--gpus', '8'] + MINI, FD_BENCH_SHARE_GPU='1', FD_DIST_BACKEND='gloo')
    assert line['n_gpus'] == 8 and line['value'] > 0
    assert line['host_margin']['launch'] == 'graph' and line['config']['launch_note'] is None


def test_live_mini_bench_line_meets_the_contract():
    '''the contract is checked on a line bench.py prints HERE (mini preset, default launch mode), not only on a committed
    artifact: every field the driver and the judge read, device-state assertions conditional on the sampler having a source.'''
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_contract import check_bench_line
    line = _bench(MINI)
    check_bench_line(line, full=False)
    assert line['host_margin']['launch'] == 'graph' and line['config']['launch_note'] is None
    assert 'traffic_stale' in line['roofline']


def test_bench_single_rank_rccl_line():
    '''N = 1 through the RCCL path (FD_FORCE_DIST=1): same line shape, rccl_ranks 1, and the
    roofline / all_gather objects are present and self-consistent.'''
    line = _bench(['--gpus', '1', '--launch', 'plan'] + MINI, FD_FORCE_DIST='1', MASTER_PORT='29533')
    assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1 and line['dist_backend'] == 'nccl'
    assert line['all_gather']['backend'].startswith('RCCL') and line['all_gather']['bytes_per_rank'] > 0
    hm = line['host_margin']
    assert hm['launch'] == 'plan' and 0 < hm['host_ms_per_forward'] and hm['launches_per_forward'] > 50
    rf = line['roofline']
    assert rf['bound'] == 'mfma' and rf['peak'] == 2516.6 and 0 < rf['frac'] < 1
    assert rf['families_fit_in_step'] and 0 < rf['frac_best_kernel'] < 1
