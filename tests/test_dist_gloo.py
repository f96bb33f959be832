'''world_size-2 gloo tests of the seed-sharding path (the N > 1 data path of bench.py,
SURVEY 8e; the reference's only data-parallel axis is the sequential loop at utils.py:90):
prompts and CPU-drawn noise sliced contiguously per rank, rank-local guidance + denoising +
decode, one all-gather of final latents and decoded images.  The gathered result must equal
the single-process result.  CPU only: the rank-local stage is the mini CPU oracle (the HIP
path has no CPU fallback), the sharding / gather code is the product's flexdiffuse_amd.dist.'''
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


PROMPTS = ['a photo of a turtle', 'zeus, god of thunder', 'a deer in a forest', 'neon city at night']


def _hot_path_shard(prompts, noise, steps=2):
    '''Guide.embeds (image-guided, Linear) -> CFG DDIM loop -> decode of one shard, on the mini
    CPU oracle: what one rank does with its slice (bench.py one_pass).'''
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import guide_ref, pipeline_ref
    from test_oracle_clip import synth_image
    torch.set_num_threads(2)
    sds = build.synthetic_state_dicts('mini', seed=0)
    ucfg, vcfg, ccfg = build.configs('mini')
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size)
    g = guide_ref.GuideRef(sds['clip'], ccfg, tok)
    embeds = g.embeds(prompt=list(prompts), guide=synth_image(20, 512, 512),
                      guide_threshold_mult=0.0, guide_clustered=0.0, guide_linear=(0.0, 0.5))
    lat, _ = pipeline_ref.denoise(sds['unet'], ucfg, embeds, g.prompt(''), noise, steps, 8.0)
    return lat, pipeline_ref.decode_image(sds['vae'], vcfg, lat)


def _worker_hot_path(rank, world, port, per_rank, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from flexdiffuse_amd import dist as fdist
    fdist.init('gloo')
    noise = fdist.global_noise(world * per_rank, (4, 8, 8), seed=1337)
    lat, img = _hot_path_shard(fdist.shard(PROMPTS, rank, world, per_rank),
                               noise[fdist.shard_range(rank, world, per_rank)])
    all_lat, all_img = fdist.all_gather_samples(lat), fdist.all_gather_samples(img)
    if rank == 0:
        torch.save((all_lat, all_img), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hot_path_equals_single_process(tmp_path):
    '''Rank-local Guide.embeds on the rank's own prompts + its noise slice + gather of latents
    and images == one process over the whole batch (results do not depend on the world size).'''
    from flexdiffuse_amd import dist as fdist
    world, per_rank = 2, 2
    out = str(tmp_path / 'hot.pt')
    mp.spawn(_worker_hot_path, args=(world, _free_port(), per_rank, out), nprocs=world, join=True)
    all_lat, all_img = torch.load(out)
    lat, img = _hot_path_shard(PROMPTS, fdist.global_noise(world * per_rank, (4, 8, 8), seed=1337))
    assert all_lat.shape == lat.shape and all_img.shape == img.shape
    # per-sample arithmetic is identical; only the GEMM batch differs (2 vs 4 rows per matmul)
    assert float((all_lat - lat).abs().max()) <= 1e-4 * float(lat.abs().max())
    assert float((all_img - img).abs().max()) <= 1e-4
    assert float((lat[0] - lat[2]).abs().max()) > 1e-3      # the prompts / noise differ per sample


def _img2img_shard(prompts, posterior, noise, steps=4, strength=0.5):
    '''img2img rank-local stage on the mini CPU oracle: encode -> posterior sample -> add_noise with the
    rank's rows -> loop from t_start -> decode (reference pipeline/flex.py:181-221, 262-287).'''
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import clip_ref, pipeline_ref
    torch.set_num_threads(2)
    sds = build.synthetic_state_dicts('mini', seed=0)
    ucfg, vcfg, ccfg = build.configs('mini')
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size)
    embeds = clip_ref.text_hidden(sds['clip'], ccfg, tok(list(prompts)).input_ids)
    uncond = clip_ref.text_hidden(sds['clip'], ccfg, tok('').input_ids)
    image = torch.linspace(-1, 1, 3 * 16 * 16).view(1, 3, 16, 16)
    lat0, t_start = pipeline_ref.img2img_init(sds['vae'], vcfg, image, posterior, noise, steps, strength,
                                              noise.shape[0])
    lat, _ = pipeline_ref.denoise(sds['unet'], ucfg, embeds, uncond, lat0, steps, 8.0, t_start=t_start)
    return lat, pipeline_ref.decode_image(sds['vae'], vcfg, lat)


def _worker_img2img(rank, world, port, per_rank, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from flexdiffuse_amd import dist as fdist
    fdist.init('gloo')
    post, noise = fdist.global_img2img_noise(world * per_rank, (4, 8, 8), seed=1337)
    lat, img = _img2img_shard(fdist.shard(PROMPTS, rank, world, per_rank), post,
                              noise[fdist.shard_range(rank, world, per_rank)])
    all_lat, all_img = fdist.all_gather_samples(lat), fdist.all_gather_samples(img)
    if rank == 0:
        torch.save((all_lat, all_img), out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_img2img_equals_single_process(tmp_path):
    '''The img2img branch (BASELINE configs[3]) is shard-invariant too: the posterior sample is the first
    draw of the one host generator (shared by all samples, as the reference repeats the encoded latent),
    the add_noise rows are slices of the second -- `dist.global_img2img_noise`, what bench.py --img2img and
    `FlexPipeline.__call__(noise=...)` use.'''
    from flexdiffuse_amd import dist as fdist
    world, per_rank = 2, 2
    out = str(tmp_path / 'i2i.pt')
    mp.spawn(_worker_img2img, args=(world, _free_port(), per_rank, out), nprocs=world, join=True)
    all_lat, all_img = torch.load(out)
    post, noise = fdist.global_img2img_noise(world * per_rank, (4, 8, 8), seed=1337)
    lat, img = _img2img_shard(PROMPTS, post, noise)
    assert all_lat.shape == lat.shape == (4, 4, 8, 8)
    assert float((all_lat - lat).abs().max()) <= 1e-4 * float(lat.abs().max())
    assert float((all_img - img).abs().max()) <= 1e-4
    assert float((lat[0] - lat[3]).abs().max()) > 1e-3
    # and the draws are the pipeline's: one generator, posterior first
    g = torch.Generator('cpu').manual_seed(1337)
    assert torch.equal(post, torch.randn((1, 4, 8, 8), generator=g))
    assert torch.equal(noise, torch.randn((4, 4, 8, 8), generator=g))


def test_single_process_is_identity():
    from flexdiffuse_amd import dist as fdist
    x = torch.arange(12.0).view(3, 4)
    assert fdist.all_gather_samples(x) is x
    assert fdist.shard_range(2, 8, 8) == slice(16, 24)


def test_bench_launcher_eight_gloo_ranks_plumbing():
    '''`bench.py --gpus 8` from a parent that imports neither torch nor HIP: eight fresh children
    rendezvous on 127.0.0.1 over gloo, take their shards of the host-drawn global batch, gather and
    max-reduce the clock; rank 0 reports n_gpus 8.  FD_BENCH_PLUMBING=1 leaves the GPU hot path out
    (there is no GPU here and the product has no CPU fallback) -- the line says it is not a
    benchmark.  The same launcher drives the real 8-GPU run.'''
    import json
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(FD_BENCH_PLUMBING='1', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, bench, '--gpus', '8', '--preset', 'mini', '--size', '64', '--batch', '2'],
                       env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['gather_equals_global_batch'] and line['backend'] == 'gloo'
    assert line['rccl_ranks'] == 0 and 'NOT a benchmark' in line['metric']


def test_bench_parent_never_loads_torch():
    '''The parent of an N-rank run starts its children with the standard library only: importing
    bench and calling launch_ranks must not pull torch (and with it the HIP runtime) into the
    process.'''
    code = ('import sys; sys.argv = ["bench.py", "--gpus", "2"]; sys.path.insert(0, %r); import bench; '
            'import os; os.environ["FD_BENCH_PLUMBING"] = "1"; rc = bench.launch_ranks(2); '
            'assert "torch" not in sys.modules and "numpy" not in sys.modules, "parent loaded torch"; '
            'sys.exit(rc)') % ROOT
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, '-c', code], env=dict(env, OMP_NUM_THREADS='1'), capture_output=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert b'"n_gpus": 2' in r.stdout


def test_bench_dead_rank_takes_the_others_down():
    '''A rank that dies before the rendezvous must not leave its peers waiting in it: the launcher
    terminates the rest and reports the failure (here: no device visible in any child).'''
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    if torch.cuda.device_count() == 0:
        r = subprocess.run([sys.executable, bench, '--gpus', '2', '--preset', 'mini'], env=env,
                           capture_output=True, timeout=300)
        assert r.returncode != 0 and b'device(s) visible' in r.stderr and not r.stdout.strip()


def test_bench_refuses_rank_count_it_cannot_start():
    '''`bench.py --gpus N` must never fall back to one GPU silently: with fewer than N visible
    devices (none in the CPU container) it exits non-zero, and under a torchrun environment a
    WORLD_SIZE different from --gpus is an error too.'''
    bench = os.path.join(ROOT, 'bench.py')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    if torch.cuda.device_count() < 3:
        # every child checks its own device before the rendezvous and exits 2
        r = subprocess.run([sys.executable, bench, '--gpus', '3'], env=env, capture_output=True, timeout=300)
        assert r.returncode != 0 and b'device(s) visible' in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, bench, '--gpus', '1'], env=dict(env, WORLD_SIZE='2', RANK='0'),
                       capture_output=True, timeout=300)
    assert r.returncode != 0 and b'does not match WORLD_SIZE' in r.stderr
