#!/usr/bin/env python3
'''Benchmark of the image-guided denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment: this process (which never touches the GPU)
starts N child ranks of itself, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
relays rank 0's JSON line and exits with the worst child status.  Under torchrun the
environment's WORLD_SIZE must equal --gpus.  Fewer than N visible devices is an error
(no silent fallback to one GPU).

One "step" = one pass of the whole hot path over one batch per GPU (BASELINE.json
configs[1]): Guide.embeds (CLIP text + ViT image towers, Linear image guidance tween) ->
50 DDIM steps of the SD1.5 UNet at 512x512 with classifier-free guidance 8, batch 8 per
GPU -> VAE decode -> (N > 1) RCCL all-gather of final latents and decoded images.
Weights are seeded synthetic tensors of the exact SD1.5 / CLIP ViT-L/14 architecture and
inputs are synthetic (no checkpoints or datasets exist offline).  Prints ONE JSON line.

Other BASELINE configs run through the same entry point (not the headline line):
    --guidance clustered_threshold          configs[2] guidance parameters (c3)
    --preset sd21 --size 768                configs[4] SD2.1-size UNet + OpenCLIP ViT-H guide (c5)
'''
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# numpy / torch are imported inside the functions that need them: the parent of an N-rank run
# (launch_ranks) starts its children with the standard library only and never loads the HIP
# runtime -- a process that has touched the GPU must not be the one that spawns the ranks.

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2516.6          # MI355X dense fp16 MFMA: 256 CU x 2.4 GHz x 4096 FLOP/clk/CU
HBM_PEAK_GBPS = 8000.0
# BASELINE.md sec. 3 / SURVEY App. B: UNet forward GFLOP per sample and VAE decode TFLOP
UNET_GFLOP = {('sd15', 256): 180.1, ('sd15', 512): 803.3, ('sd15', 768): 2148.1, ('sd21', 768): 2149.1}
VAE_TFLOP = {256: 0.622, 512: 2.515, 768: 5.754}

# the three guidance parameterisations BASELINE.json names
GUIDANCE = {
    'linear': dict(guide_threshold_mult=0.0, guide_clustered=0.0, guide_linear=(0.0, 0.5),
                   guide_max_guidance=0.5),
    'clustered_threshold': dict(guide_threshold_floor=0.75, guide_threshold_mult=0.25,
                                guide_clustered=0.25, guide_linear=(0.0, 0.0),
                                guide_max_guidance=0.35, guide_header_max=0.0),
    'defaults': dict(),
}

# reference guidance stage measured with the reference's own guidance.py / encode/clip.py in the
# build container (8 host cores, torch 2.10 CPU fp32, random-weight ViT-L/14):
# the reference source cannot travel to the GPU box, so these are constants.
REFERENCE_GUIDANCE_STAGE = {
    'source': 'BASELINE.md sec. 2 (reference guidance.py / encode/clip.py imported in the build '
              'container, 8 cores)',
    'tween_ms_per_prompt': 83.0, 'map_emb_image_guide_ms': 109.0, 'clip_image_s': 0.58,
    'clip_prompt_s': 0.047, 'guide_embeds_s_per_prompt': [0.52, 0.67],
    'host_syncs_per_tween': 19700,
}


def flops_per_image(preset, size, evals):
    '''2 UNet forwards (CFG) per executed DDIM evaluation + one VAE decode (img2img runs
    int(steps * strength) evaluations and its one VAE encode per CALL is not priced).'''
    u, v = UNET_GFLOP.get((preset, size)), VAE_TFLOP.get(size)
    if u is None or v is None:
        return None
    return 2 * evals * u * 1e9 + v * 1e12


def synth_image(seed, w, h):
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h + 8, w + 8, 3)).astype(np.float32)
    c = np.pad(np.cumsum(np.cumsum(a, 0), 1), ((1, 0), (1, 0), (0, 0)))
    blur = ((c[8:, 8:] - c[:-8, 8:] - c[8:, :-8] + c[:-8, :-8]) / 64.0)[:h, :w]
    blur = (blur - blur.min()) / (blur.max() - blur.min()) * 255.0
    return Image.fromarray(blur.astype(np.uint8), 'RGB')


def synth_prompts(n, seed=1):
    import numpy as np
    rng = np.random.default_rng(seed)
    words = ['photo', 'turtle', 'forest', 'zeus', 'city', 'painting', 'deer', 'storm', 'neon',
             'ancient', 'river', 'portrait', 'rock', 'monkey', 'anime', 'golden', 'light']
    return [' '.join(rng.choice(words, size=int(rng.integers(5, 20)))) for _ in range(n)]


def cpu_baseline(sds, cfgs, steps, size):
    '''Oracle (torch fp32 restatement of the reference path) timed on the host cores over a
    bounded sample: one CFG UNet evaluation (2 forwards) of ONE image plus one VAE decode,
    extrapolated to steps x UNet + decode per image.'''
    import torch
    from oracle import unet_ref, vae_ref
    ucfg, vcfg, _ = cfgs
    g = torch.Generator().manual_seed(0)
    h = size // 8
    x = torch.randn((2, 4, h, h), generator=g)
    ctx = torch.randn((2, 77, ucfg.cross_attention_dim), generator=g)
    t0 = time.time()
    unet_ref.unet_forward(sds['unet'], ucfg, x, 500, ctx)
    t_unet = time.time() - t0
    t0 = time.time()
    vae_ref.vae_decode(sds['vae'], vcfg, x[:1])
    t_vae = time.time() - t0
    per_image = steps * t_unet + t_vae
    return {'value': 1.0 / per_image, 'unit': 'images/sec', 'cores': torch.get_num_threads(),
            'kind': 'port',
            'sample': f'1 CFG UNet evaluation (2 forwards, 1 image, {h}x{h} latents) = {t_unet:.2f} s '
                      f'and 1 VAE decode = {t_vae:.2f} s on {torch.get_num_threads()} threads '
                      f'({os.cpu_count()} cpus); extrapolated to {steps} steps + decode per image',
            'reference_guidance_stage': REFERENCE_GUIDANCE_STAGE}


def parity_c1(sds, cfgs, pipe, enc, tok, steps=10, hw=256, guidance=8.0):
    '''BASELINE configs[0] shape (256x256, 10 DDIM steps, batch 1, CFG 8) on the same SD1.5
    weights: GPU fp16 path vs the CPU fp32 oracle with identical ids and CPU-drawn noise.'''
    import torch
    from flexdiffuse_amd import SimpleGuide
    from oracle import clip_ref, pipeline_ref
    ucfg, vcfg, ccfg = cfgs
    prompt = 'a photo of a turtle in a forest, oil painting'
    emb_dev = enc.prompt(prompt)
    pipe(guide=SimpleGuide(enc, pipe.unet, guidance, steps, emb_dev), init_size=(hw, hw),
         generator=torch.Generator('cpu').manual_seed(1337), output_type='np')
    text_sd = {k: v for k, v in sds['clip'].items() if k.startswith('text_model')}
    emb_ref = clip_ref.text_hidden(text_sd, ccfg, tok(prompt).input_ids)
    unc_ref = clip_ref.text_hidden(text_sd, ccfg, tok('').input_ids)
    lat0 = torch.randn((1, 4, hw // 8, hw // 8), generator=torch.Generator('cpu').manual_seed(1337))
    t0 = time.time()
    lat_ref, used = pipeline_ref.denoise(sds['unet'], ucfg, emb_ref, unc_ref, lat0, steps, guidance)
    img_ref = pipeline_ref.decode_image(sds['vae'], vcfg, lat_ref)
    cpu_s = time.time() - t0
    lat = pipe.last_latents.float().cpu()
    return {'config': f'SD1.5 {hw}x{hw}, {steps} DDIM steps, batch 1, CFG {guidance} (BASELINE configs[0])',
            'psnr_db': pipeline_ref.psnr(pipe.last_images.cpu(), img_ref),
            'latent_max_abs_err': float((lat - lat_ref).abs().max()),
            'timesteps_equal': used == [int(t) for t in pipe.scheduler.timesteps],
            'cpu_oracle_seconds': cpu_s, 'tolerance': 'PSNR >= 40 dB'}


def parity_c2(sds, cfgs, pipe, guide_ctx, enc):
    '''BASELINE configs[1] at batch 1 (512x512, 50 DDIM steps, CFG 8, Linear image guidance):
    the device path (Guide.embeds -> FlexPipeline) against the CPU fp32 oracle's final latents
    of the same sample, cached in tests/golden/c2_oracle.npz by tests/golden/make_c2_oracle.py
    (100 fp32 UNet forwards, ~13 min on 8 cores); only the oracle's VAE decode runs here.'''
    import hashlib
    import numpy as np
    import torch
    path = os.path.join(ROOT, 'tests', 'golden', 'c2_oracle.npz')
    if not os.path.exists(path):
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from make_c2_oracle import C2, c2_inputs
    from flexdiffuse_amd import SimpleGuide
    from oracle import pipeline_ref
    ucfg, vcfg, ccfg = cfgs
    o = np.load(path)
    steps, size = int(o['steps'][0]), int(o['size'][0])
    prompt, img, lat0 = c2_inputs(size)
    assert hashlib.sha256(lat0.numpy().tobytes()).digest() == o['lat0_sha'].tobytes()
    embeds = guide_ctx.embeds(prompt=prompt, guide=img, **C2['embeds_kw'])
    pipe(guide=SimpleGuide(enc, pipe.unet, C2['guidance'], steps, embeds), init_size=(size, size),
         latents=lat0, output_type='np')
    lat_ref = torch.from_numpy(o['latents'])
    img_ref = pipeline_ref.decode_image(sds['vae'], vcfg, lat_ref)
    emb_ref = torch.from_numpy(o['embeds'])
    lat = pipe.last_latents.float().cpu()
    return {'config': f'SD1.5 {size}x{size}, {steps} DDIM steps, batch 1, CFG {C2["guidance"]}, Linear '
                      f'image guidance (BASELINE configs[1], sample 0)',
            'psnr_db': pipeline_ref.psnr(pipe.last_images.cpu(), img_ref),
            'latent_max_abs_err': float((lat - lat_ref).abs().max()),
            'latent_ref_max_abs': float(lat_ref.abs().max()),
            'guided_embeds_max_abs_err': float((embeds.float().cpu() - emb_ref).abs().max()),
            'timesteps_equal': [int(t) for t in o['timesteps']] == [int(t) for t in pipe.scheduler.timesteps],
            'oracle': 'tests/golden/c2_oracle.npz (oracle output cached by tests/golden/make_c2_oracle.py; '
                      f'{float(o["cpu_seconds"].sum()):.0f} s on {int(o["threads"][0])} threads)',
            'tolerance': 'PSNR >= 40 dB'}


# Every launch of the roofline leg is bracketed (stride 1): the leg replays the recorded launch plan, so a bracket costs
# stream time only in that untimed pass.  FD_BENCH_EVENT_STRIDE=n samples every n-th launch of a family instead.
EVENT_STRIDE = max(1, int(os.environ.get('FD_BENCH_EVENT_STRIDE', '1')))
FAMILY_NAMES = {0: 'gemm', 1: 'attention', 2: 'groupnorm', 3: 'other_kernels'}


def aggregate_brackets(fam, tag, ms, work, executed, empty_ms, stride=1):
    '''Per kernel family, from the individual event brackets of one pass (flexdiffuse_amd.hip.prof_drain):
    the brackets are grouped by (family, tag, declared work) -- one group = one launch shape -- and a group contributes
    MEDIAN(bracket) - empty bracket, times its launch count: a host stall that lands between an event record and its
    launch (round 5: one 39 ms stall, scaled by the sampling stride, became 273 ms of "GroupNorm") moves one bracket of
    one group and leaves the median where it was.  `raw_ms` keeps the plain sum for comparison.'''
    import numpy as np
    out = {}
    fam, tag, ms, work, executed = (np.asarray(a) for a in (fam, tag, ms, work, executed))
    for code, name in FAMILY_NAMES.items():
        sel = fam == code
        rec = {'ms': 0.0, 'raw_ms': float(ms[sel].sum()) * stride, 'work': float(work[sel].sum()) * stride,
               'executed': float(executed[sel].sum()) * stride, 'launches': int(sel.sum()) * stride, 'groups': 0,
               'worst_bracket_over_median': 1.0}
        keys = {}
        for i in np.nonzero(sel)[0]:
            keys.setdefault((int(tag[i]), float(work[i])), []).append(float(ms[i]))
        for v in keys.values():
            med = float(np.median(v))
            rec['ms'] += max(med - empty_ms, 0.0) * len(v) * stride
            if med > 0:
                rec['worst_bracket_over_median'] = max(rec['worst_bracket_over_median'], max(v) / med)
        rec['groups'] = len(keys)
        out[name] = rec
    return out


def time_budget(fams, ms_per_step):
    '''Where a step's wall time goes, summing to ms_per_step by construction: the four kernel families of the roofline
    leg and the remainder (launch gaps on the device, torch's own fill / copy kernels, the device->host copy of the
    images, host work the device waits for).  `valid` is False when the families alone exceed the step.'''
    b = {k: fams[k]['ms'] for k in ('gemm', 'attention', 'groupnorm', 'other_kernels')}
    kernels = sum(b.values())
    b['gaps_and_host'] = ms_per_step - kernels
    return b, kernels <= ms_per_step


def best_kernel_leg(dev):
    '''The level-0 ResBlock conv3x3 (16 x 64 x 64 x 320 -> 320, 120.8 GFLOP, the largest single
    share of the pass) timed alone: 20 launches between two events on the launch stream.'''
    import torch
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(3)
    x = ops.Act((torch.randn((16 * 4096, 320), generator=g) * 0.5).half().to(dev), 16, 64, 64)
    w = ops.prep_conv(torch.randn((320, 320, 3, 3), generator=g) * 0.02, torch.zeros(320), dev)
    for _ in range(3):
        ops.conv2d(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.conv2d(x, w)
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / n
    tflops = 2.0 * 65536 * 320 * 2880 / (us * 1e-6) / 1e12
    return {'kernel': 'level-0 conv3x3 16x64x64x320->320 (implicit GEMM M 65536, N 320, K 2880)',
            'avg_launch_us': us, 'achieved': tflops, 'frac': tflops / MFMA_PEAK_TFLOPS}


GEMM_SOURCES = ('common.h', 'gemm.hip', 'gemm_epilogue.h', 'gemm_pp.hip')


def sources_sha() -> str:
    '''sha256 over the sources of the dominant kernel family (csrc/: the GEMM / implicit-GEMM files + common.h), the key that
    ties a committed PMC traffic record (profiles/r*_pmc_traffic.json `sources_sha`) to the code it was measured on.'''
    import hashlib
    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(ROOT, 'flexdiffuse_amd', 'csrc', name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()


def pick_traffic_record():
    '''(hbm bytes per launch of the dominant kernel, description, other kernels, stale flag) from the committed PMC records:
    rocprofv3 cannot run inside this process, so `roofline.traffic` is a per-launch measurement taken by tools/pmc_traffic_r05.sh.
    The newest record whose `sources_sha` equals today's sources is used; if none matches, the newest record is reported with
    `traffic_stale: true` (measured on other code).'''
    import glob
    import re
    recs = []
    for path in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')):
        try:
            with open(path) as f:
                tj = json.load(f)
            tj['hbm_bytes'], tj['algorithmic_bytes'], tj['problem']      # noqa: B018 -- required keys
        except (OSError, KeyError, ValueError):
            continue
        m = re.match(r'r(\d+)_', os.path.basename(path))
        recs.append((int(m.group(1)) if m else 0, os.path.basename(path), tj))
    if not recs:
        return None, None, None, True
    recs.sort(key=lambda r: r[0], reverse=True)
    try:
        sha = sources_sha()
    except OSError:
        sha = None
    match = [r for r in recs if sha and r[2].get('sources_sha') == sha]
    _, name, tj = (match or recs)[0]
    stale = not match
    of = (f"{tj['problem']}: {tj['hbm_bytes'] / 1e6:.1f} MB per launch vs {tj['algorithmic_bytes'] / 1e6:.1f} MB algorithmic; "
          f"per-launch PMC measurement from profiles/{name} ({'measured on OTHER sources: stale' if stale else 'sources_sha matches this tree'}), "
          'not re-measured by this run')
    other = {k: {'hbm_bytes': v['hbm_bytes'], 'algorithmic_bytes': v['algorithmic_bytes'],
                 'over_algorithmic': v.get('traffic_over_algorithmic')}
             for k, v in tj.get('other_kernels', {}).items()} or None
    return tj['hbm_bytes'], of, other, stale


def _bdf_key(bdf):
    '''(bus, device, function) of a PCI address in any of the spellings amdsmi / sysfs / torch use.'''
    import re
    m = re.search(r'([0-9a-f]{2}):([0-9a-f]{2})\.([0-7])\s*$', str(bdf).strip().lower())
    return (int(m.group(1), 16), int(m.group(2), 16), int(m.group(3))) if m else None


def devmon_collect(proc, t0: float, t1: float, device_bdf=None) -> dict:
    '''Stops the sampler child (tools/devmon.py) and averages its samples inside [t0, t1] -- the timed region.
    Makes "fast box / slow box" a number: the MFMA loops are power-capped, so the sustained gfx clock moves the
    result by a few percent between boxes of one pool.'''
    out = {'avg_sclk_mhz': None, 'avg_power_w': None, 'clock_samples': 0, 'clock_source': None}
    if proc is None:
        return out
    try:
        raw, _ = proc.communicate(input=b'', timeout=20)
        rec = json.loads(raw.decode().strip().splitlines()[-1])
    except Exception:      # noqa: BLE001 -- the sampler is best effort
        try:
            proc.kill()
        except OSError:
            pass
        return out
    out['clock_source'] = rec.get('source')
    # which physical GPU was sampled, and is it the one that was timed (amdsmi / sysfs ignore *_VISIBLE_DEVICES)
    out['clock_device_bdf'] = rec.get('bdf')
    if rec.get('bdf') and device_bdf and _bdf_key(rec['bdf']) and _bdf_key(device_bdf):
        out['clock_device_matches'] = _bdf_key(rec['bdf']) == _bdf_key(device_bdf)
    if rec.get('errors') and not rec.get('samples'):
        out['clock_errors'] = rec['errors'][:2]
    inside = [s for s in rec.get('samples', []) if t0 <= s[0] <= t1]
    for key, col in (('avg_sclk_mhz', 1), ('avg_power_w', 2), ('avg_mclk_mhz', 3)):
        vals = [s[col] for s in inside if len(s) > col and s[col] is not None]
        if vals:
            out[key] = sum(vals) / len(vals)
            if col == 1:
                out['min_sclk_mhz'], out['max_sclk_mhz'] = min(vals), max(vals)
    out['clock_samples'] = len(inside)
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    '''Parent of an N-rank run: starts N fresh children, one rank per GPU over RCCL, relays
    rank 0's line and returns the worst child status.  Standard library only -- this process
    never imports torch, never counts devices through HIP and never touches the GPU; each child
    checks for itself that its device exists (`--gpus N` with fewer than N visible devices makes
    every child exit 2 before the rendezvous).  A child that dies takes the others down with it
    instead of leaving them in the rendezvous.'''
    port = _free_port()
    threads = max(1, (os.cpu_count() or n) // n)      # host threads of torch's CPU ops per rank
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        env.setdefault('OMP_NUM_THREADS', str(threads))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, p in enumerate(procs):          # exactly the processes started above
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write((out[0] if out else b'').decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def check_rank_device(local_rank: int, world: int) -> int:
    '''Device ordinal of this rank, or exit 2: no silent fallback to fewer GPUs.
    FD_BENCH_SHARE_GPU=1 (plumbing tests on a 1-GPU box) lets several ranks share devices.'''
    import torch
    visible = torch.cuda.device_count()
    share = os.environ.get('FD_BENCH_SHARE_GPU') == '1'
    if visible < 1 or (visible < world and not share):
        print(f'bench.py: --gpus {world} but only {visible} device(s) visible', file=sys.stderr)
        sys.exit(2)
    return local_rank % visible


def plumbing_rank(args) -> None:
    '''FD_BENCH_PLUMBING=1 (tests only, never a bench result): what a rank does AROUND the hot
    path -- join the process group, take its shard of prompts and of the host-drawn noise, gather,
    max-reduce the clock -- with no GPU work in between, so that the N-rank launcher and the
    sharding can be exercised where no GPU exists.  The line says so in `metric` and `data`.'''
    import torch
    import torch.distributed as dist
    from flexdiffuse_amd import dist as fdist
    rank, world, _ = fdist.init('gloo')
    B = args.batch
    prompts = fdist.shard(synth_prompts(B * world), rank, world, B)
    h = args.size // 8
    noise = fdist.global_noise(B * world, (4, h, h), 1337)
    mine = noise[fdist.shard_range(rank, world, B)]
    if dist.is_initialized():
        dist.barrier()
    t0 = time.time()
    gathered = fdist.all_gather_samples(mine)
    elapsed = time.time() - t0
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ok = bool(torch.equal(gathered, noise)) and len(prompts) == B
    if rank == 0:
        print(json.dumps({'metric': 'plumbing-only (launcher + sharding + gather; NOT a benchmark)',
                          'value': 0.0, 'unit': 'images/sec', 'n_gpus': world, 'steps': 0, 'warmup': 0,
                          'data': 'plumbing-only (no GPU work)', 'rccl_ranks': 0,
                          'gather_equals_global_batch': ok, 'gather_ms': 1e3 * elapsed,
                          'backend': dist.get_backend() if dist.is_initialized() else 'none',
                          'omp_num_threads': os.environ.get('OMP_NUM_THREADS')}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--preset', default='sd15', choices=['sd15', 'sd21', 'mini', 'mini2'])
    ap.add_argument('--guidance', default='linear', choices=sorted(GUIDANCE))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--graph', action='store_true', help='same as --launch graph')
    ap.add_argument('--launch', default='graph', choices=['graph', 'plan', 'eager'],
                    help='graph (default): the UNet forward is replayed from a captured HIP graph -- one host call per '
                         'denoising step, so a host stall (shared hosts of the GPU pool: one 30 ms stall in four passes seen in '
                         'plan mode) cannot starve the device; same kernels and, since round 4, the same device time as plan '
                         '(profiles/r04_session_ab.txt); plan: the recorded launch plan (same kernels, same order, eager '
                         'launches, no per-op host work); eager: every op goes through the Python front each step')
    ap.add_argument('--img2img', action='store_true',
                    help='BASELINE configs[3] (c4): start from a synthetic init image of --size, '
                         '--strength 0.6 => int(steps * strength) UNet evaluations (pipeline/flex.py:181-221)')
    ap.add_argument('--strength', type=float, default=0.6)
    ap.add_argument('--scheduler', default='ddim', choices=['ddim', 'pndm', 'lms'],
                    help='ddim (BASELINE metric); pndm: the scheduler the reference harness actually passes '
                         '(SD-v1-4 ships PNDM, utils.py:70; PLMS: steps + 1 UNet evaluations); lms: K-LMS')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    if env_world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} does not match WORLD_SIZE={env_world}', file=sys.stderr)
        sys.exit(2)
    if os.environ.get('FD_BENCH_PLUMBING') == '1':
        plumbing_rank(args)

    # clock / power sampler: a child process (stdlib + amdsmi, never HIP), started before this process's
    # first GPU call; rank 0's device only
    devmon = None
    if int(os.environ.get('RANK', '0')) == 0 and os.environ.get('FD_BENCH_DEVMON', '1') != '0':
        try:
            devmon = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'devmon.py')],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                      env=dict(os.environ, FD_DEVMON_INDEX=os.environ.get('LOCAL_RANK', '0')))
        except OSError:
            devmon = None

    import torch
    os.environ['LOCAL_RANK'] = str(check_rank_device(int(os.environ.get('LOCAL_RANK', '0')), env_world))
    import torch.distributed as dist
    from flexdiffuse_amd import dist as fdist
    rank, world, local_rank = fdist.init('nccl')
    if dist.is_initialized():
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    if world == 1:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', local_rank if world > 1 else 0)

    from flexdiffuse_amd import Guide, SimpleGuide, build, hip, ops
    from flexdiffuse_amd.encode.clip import CLIPEncoder
    hip.lib()
    info = hip.device_info(dev.index or 0)

    t_setup = time.time()
    sds = build.synthetic_state_dicts(args.preset, seed=0)
    cfgs = build.configs(args.preset)
    pipe, clip, tok = build.build_models(sds, args.preset, dev, vae_encoder=args.img2img)
    if args.scheduler != 'ddim':
        from flexdiffuse_amd.scheduler import LMSDiscreteScheduler, PNDMScheduler
        pipe.scheduler = PNDMScheduler() if args.scheduler == 'pndm' else LMSDiscreteScheduler()
    pipe.use_graph = args.graph or args.launch == 'graph'
    pipe.pause_gc = True          # with the gc.freeze() below: no collector pauses inside the timed loops
    pipe.use_plan = args.launch == 'plan' and not args.graph
    guide_ctx = Guide(clip, tok, device='cuda')
    enc = CLIPEncoder(clip, tok)
    B, N = args.batch, world
    prompts = fdist.shard(synth_prompts(B * N), rank, N, B)
    guide_img = synth_image(2, 512, 512)
    hw = args.size
    # the whole global batch of noise is drawn once on the host and sliced per rank, so the
    # images do not depend on the number of GPUs
    noise = fdist.global_noise(B * N, (4, hw // 8, hw // 8), 1337)[fdist.shard_range(rank, N, B)].to(dev)
    init_image = None
    if args.img2img:
        # c4: the (1,3,H,W) tensor in [-1,1] the pipeline takes when it is not handed a PIL image
        # (`preprocess` would resize a PIL image's long side to 512, encode/clip.py:15-39)
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
        from make_c45_oracle import init_tensor
        init_image = init_tensor(3, hw).to(dev)
        # the global batch's img2img draws in the pipeline's order (posterior sample, then add_noise rows)
        # from one host generator; this rank's rows of the second
        i2i_noise = fdist.global_img2img_noise(B * N, (4, hw // 8, hw // 8), 1337)[1][
            fdist.shard_range(rank, N, B)].to(dev)
    ddim_evals = args.ddim_steps if not args.img2img else min(int(args.ddim_steps * args.strength), args.ddim_steps)
    if args.scheduler == 'pndm':
        ddim_evals += 1          # PLMS warm-up: the second timestep is evaluated twice
    t_setup = time.time() - t_setup
    gather = {'bytes_per_rank': 0, 'ms': 0.0, 'calls': 0}

    def one_pass(time_gather=False):
        embeds = guide_ctx.embeds(prompt=prompts, guide=guide_img, **GUIDANCE[args.guidance])
        sg = SimpleGuide(enc, pipe.unet, 8.0, args.ddim_steps, embeds)
        if init_image is not None:
            # img2img: VAE-encode the init image once; the posterior sample is the first draw of the
            # same host generator on every rank, the add_noise rows are this rank's slice of the global
            # draw (shard-invariant, like txt2img); t_start from strength
            out = pipe(guide=sg, init_image=init_image, strength=args.strength, noise=i2i_noise,
                       generator=torch.Generator('cpu').manual_seed(1337), output_type='np')
        else:
            out = pipe(guide=sg, init_size=(hw, hw), latents=noise, output_type='np')
        # one RCCL all-gather of the final latents and of the decoded images (identity at N=1)
        if time_gather:
            torch.cuda.synchronize()
            tg = time.time()
        all_latents = fdist.all_gather_samples(pipe.last_latents)
        all_images = fdist.all_gather_samples(pipe.last_images)
        if time_gather:
            torch.cuda.synchronize()
            gather['ms'] += 1e3 * (time.time() - tg)
            gather['calls'] += 1
            gather['bytes_per_rank'] = (pipe.last_latents.numel() * pipe.last_latents.element_size()
                                        + pipe.last_images.numel() * pipe.last_images.element_size())
        assert all_latents.shape[0] == (B * N if dist.is_initialized() else B)
        return out, all_latents, all_images

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # (graph capture happens in the first pass; if it fails on this box FlexPipeline itself falls back to the launch plan
    # and says so in `graph_fallback`)
    for w in range(args.warmup):
        one_pass()
    launch_note = pipe.graph_fallback
    # The heap now holds the models, the tokenizer tables and the recorded launch plans (millions of long-lived
    # objects): a full collection over it is a 50-100 ms host stall, and the collector would schedule one whenever a
    # pass's temporaries push the young generations over their thresholds.  Collect once, then move everything that
    # survived into the permanent generation -- later collections only look at what a pass itself allocates.
    import gc
    gc.collect()
    gc.freeze()
    sync()
    t0 = time.time()
    for k in range(args.steps):
        one_pass()
    sync()
    t1 = time.time()
    elapsed = t1 - t0
    try:
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        timed_bdf = '%04x:%02x:%02x.0' % (getattr(pr, 'pci_domain_id', 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:      # noqa: BLE001 -- older torch builds do not expose the PCI address
        timed_bdf = None
    device_state = devmon_collect(devmon, t0, t1, timed_bdf)
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64,
                            device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- host margin: host time to ISSUE one CFG UNet forward vs device time to run it (the
    # device is fed as long as the host is the faster of the two)
    margin = None
    if rank == 0:
        sg = SimpleGuide(enc, pipe.unet, 8.0, args.ddim_steps,
                         guide_ctx.embeds(prompt=prompts, guide=guide_img, **GUIDANCE[args.guidance]))
        lat_m = pipe.loop_latents(noise)
        ctx_m = sg.stacked_embeds()
        for _ in range(2):
            pipe._unet_eps(lat_m, 500, ctx_m, 2)
        torch.cuda.synchronize()
        n_m = 5
        tm0 = time.time()
        for i in range(n_m):
            pipe._unet_eps(lat_m, 480 - 20 * i, ctx_m, 2)
        tm1 = time.time()
        torch.cuda.synchronize()
        tm2 = time.time()
        margin = {'host_ms_per_forward': 1e3 * (tm1 - tm0) / n_m,
                  'device_ms_per_forward': 1e3 * (tm2 - tm0) / n_m,
                  'launch': 'graph' if pipe.use_graph else ('plan' if pipe.use_plan else 'eager'),
                  'launches_per_forward': pipe.plan_launches(),
                  'note': 'issue time of 5 back-to-back CFG UNet forwards vs their completion time'}
        margin['host_over_device'] = margin['host_ms_per_forward'] / margin['device_ms_per_forward']

    # ---- roofline leg: one extra UNTIMED pass with a pair of HIP events on the launch stream around every launch of
    # the library (the four families of FD_FAMILY_*).  Same launch mode as the timed region: where that replays a HIP
    # graph the leg replays the recorded LAUNCH PLAN -- the same kernels, tiles and order issued by one host call per
    # forward -- instead of the eager Python front (round 5's leg, whose host stalls ended up inside brackets).
    # Aggregation: median per launch shape x count (aggregate_brackets).  If the families still do not fit in the step
    # the leg is run once more (`roofline_retries`); `roofline_valid` says whether the final numbers fit.
    fams, empty_ms, roofline_retries, roofline_valid = None, 0.0, 0, False
    ms_per_step = 1e3 * elapsed / args.steps
    if rank == 0:
        was_graph, was_plan = pipe.use_graph, pipe.use_plan
        if pipe.use_graph:
            pipe.use_graph, pipe.use_plan = False, True
        one_pass()                        # records the plan (and warms the mode) with the recorder off
        torch.cuda.synchronize()
        for attempt in range(2 if world == 1 else 1):      # (N > 1: every rank takes part in each pass's all-gather -- no unilateral retry)
            empty_ms = hip.prof_calibrate(256)
            hip.prof_set_stride(EVENT_STRIDE)
            hip.prof_enable(True)
            one_pass(time_gather=(attempt == 0))
            torch.cuda.synchronize()
            hip.prof_enable(False)
            fams = aggregate_brackets(*hip.prof_drain(), empty_ms, EVENT_STRIDE)
            _, roofline_valid = time_budget(fams, ms_per_step)
            roofline_retries = attempt
            if roofline_valid:
                break
        pipe.use_graph, pipe.use_plan = was_graph, was_plan
    elif dist.is_initialized():
        one_pass()
        one_pass(time_gather=True)       # every rank takes part in the extra passes' all-gathers
        torch.cuda.synchronize()

    import resource
    rss_mib = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0
    rss_max = rss_mib
    if dist.is_initialized():
        t_rss = torch.tensor([rss_mib], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t_rss, op=dist.ReduceOp.MAX)
        rss_max = float(t_rss[0])
    if rank == 0:
        images = B * N * args.steps
        value = images / elapsed
        g, att, gn = fams['gemm'], fams['attention'], fams['groupnorm']
        achieved = (g['work'] / (g['ms'] * 1e-3)) / 1e12 if g['ms'] > 0 else 0.0
        executed = (g['executed'] / (g['ms'] * 1e-3)) / 1e12 if g['ms'] > 0 else 0.0
        budget, _ = time_budget(fams, ms_per_step)
        fam_sum = g['ms'] + att['ms'] + gn['ms']
        # HBM bytes of the dominant kernel (level-0 conv3x3) from its PMC pass: rocprofv3 cannot
        # run inside this process, so the committed per-launch measurement is reported with its
        # source (it is NOT re-measured by this run)
        traffic, traffic_of, traffic_other, traffic_stale = pick_traffic_record()
        fpi = flops_per_image(args.preset, hw, ddim_evals)
        headline = (args.preset == 'sd15' and hw == 512 and args.ddim_steps == 50 and args.scheduler == 'ddim'
                    and args.guidance == 'linear' and B == 8 and not args.img2img)
        cfg_name = {('sd15', 'linear', False): 'BASELINE configs[1]',
                    ('sd15', 'clustered_threshold', False): 'BASELINE configs[2] guidance',
                    ('sd15', 'linear', True): 'BASELINE configs[3]',
                    ('sd21', 'linear', False): 'BASELINE configs[4]'}.get(
                        (args.preset, args.guidance, args.img2img), 'non-headline')
        model = {'sd15': 'SD1.5', 'sd21': 'SD2.1 + OpenCLIP ViT-H/14 guide'}.get(args.preset, args.preset)
        mode = (f'img2img strength {args.strength} ({ddim_evals} of {args.ddim_steps} DDIM steps)'
                if args.img2img else f'{args.ddim_steps}-step {args.scheduler.upper()}')
        line = {
            'metric': '512x512 50-step images/sec/node (SD1.5, batch=8/GPU, Linear image guidance)'
                      if headline else f'{hw}x{hw} {mode} images/sec/node ({model}, '
                                       f'batch={B}/GPU, {args.guidance} image guidance)',
            'value': value, 'unit': 'images/sec', 'n_gpus': N, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'fp16',
            'data': 'synthetic',
            'config': {'workload': f'{model} {hw}x{hw} {mode} + {args.guidance} image '
                                   f'guidance, CFG 8, batch={B}/GPU ({cfg_name})',
                       'images_per_step': B * N, 'parallelism': f'seed-sharded x{N}',
                       'preset': args.preset, 'guidance': args.guidance, 'img2img': args.img2img,
                       'unet_evaluations_per_image': ddim_evals, 'launch': margin['launch'],
                       'scheduler': args.scheduler, 'launch_note': launch_note},
            # ranks that took part in an RCCL (torch.distributed "nccl") process group; 0 for any
            # other backend (gloo plumbing runs) and for an undistributed N=1 run
            'rccl_ranks': dist.get_world_size() if dist.is_initialized() and dist.get_backend() == 'nccl' else 0,
            'dist_backend': dist.get_backend() if dist.is_initialized() else None,
            'host_margin': margin, 'host_ms_per_forward': margin['host_ms_per_forward'],
            'all_gather': {'bytes_per_rank': gather['bytes_per_rank'],
                           'ms': gather['ms'] / max(gather['calls'], 1),
                           'backend': ('RCCL (torch.distributed nccl)' if dist.get_backend() == 'nccl'
                                       else dist.get_backend()) if dist.is_initialized() else 'none (N=1)'},
            'roofline': {
                'bound': 'mfma', 'kernel': 'k_gemm_f16 (implicit-GEMM conv3x3 / GEMM family)',
                'achieved': achieved, 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / MFMA_PEAK_TFLOPS,
                # MACs the kernels really issued / time / peak: `achieved` prices every launch at the ALGORITHMIC work of
                # the op it implements (a parity-decomposed upsample convolution executes 4/9 of its 3x3 MACs, a Winograd
                # F(2x2,3x3) convolution 4/9 as well); this is the hardware-side rate of the same launches
                'executed': executed, 'executed_frac': executed / MFMA_PEAK_TFLOPS,
                'traffic': traffic, 'traffic_of': traffic_of,
                # false only when the PMC record was collected on exactly these GEMM sources (sha256 recorded with it)
                'traffic_stale': traffic_stale,
                'traffic_other_kernels': traffic_other,
                # one untimed pass in the timed region's launch mode (graph -> the recorded launch plan), every launch
                # bracketed by HIP events on its stream; per launch shape MEDIAN bracket - empty bracket, x launch count
                'leg_launch': 'plan' if margin['launch'] == 'graph' else margin['launch'],
                'aggregation': 'median per (family, launch shape) x launches',
                'sampled_every': EVENT_STRIDE, 'sampled_launches': g['launches'] // EVENT_STRIDE,
                'launches': g['launches'], 'launch_shapes': g['groups'],
                'empty_bracket_us': 1e3 * empty_ms,
                'kernel_ms_per_pass': g['ms'], 'kernel_ms_per_pass_plain_sum': g['raw_ms'] - g['launches'] * empty_ms,
                'avg_launch_us': 1e3 * g['ms'] / g['launches'] if g['launches'] else None,
                'attention_tflops': (att['work'] / (att['ms'] * 1e-3)) / 1e12 if att['ms'] else 0.0,
                'attention_ms_per_pass': att['ms'],
                # GroupNorm priced at SURVEY 8(d)'s 4 B/element (fp16 read + write)
                'groupnorm_gbps': (gn['work'] / (gn['ms'] * 1e-3)) / 1e9 if gn['ms'] else 0.0,
                'groupnorm_frac_of_hbm': ((gn['work'] / (gn['ms'] * 1e-3)) / 1e9 / HBM_PEAK_GBPS) if gn['ms'] else 0.0,
                'groupnorm_ms_per_pass': gn['ms'],
                'other_kernels_ms_per_pass': fams['other_kernels']['ms'],
                'families_ms_per_pass': fam_sum,
                'families_fit_in_step': fam_sum <= ms_per_step,
                # sums to ms_per_step: the four families of library launches + everything else (device-side launch gaps,
                # torch's fill / copy kernels, the device->host copy of the images, host work the device waits for)
                'time_budget_ms': budget,
                'roofline_valid': roofline_valid, 'roofline_retries': roofline_retries,
                'worst_bracket_over_median': {k: v['worst_bracket_over_median'] for k, v in fams.items()},
                'end_to_end_frac_of_mfma_roofline':
                    (value / N) * fpi / (MFMA_PEAK_TFLOPS * 1e12) if fpi else None,
            },
            'device': dict(info, **device_state), 'setup_s': t_setup,
            # peak resident set of a rank's host process (MiB): rank 0's, and the largest over the ranks
            'host_rss_mib': {'rank0': rss_mib, 'max_over_ranks': rss_max},
        }
        line['roofline']['best_kernel'] = best_kernel_leg(dev)
        line['roofline']['frac_best_kernel'] = line['roofline']['best_kernel']['frac']
        if N == 1 and not args.no_parity and args.preset == 'sd15':
            line['parity'] = {'c1': parity_c1(sds, cfgs, pipe, enc, tok)}
            c2 = parity_c2(sds, cfgs, pipe, guide_ctx, enc)
            if c2 is not None:
                line['parity']['c2'] = c2
        if N == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(sds, cfgs, args.ddim_steps, hw)
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
