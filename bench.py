#!/usr/bin/env python3
'''Benchmark of the image-guided denoising hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch per GPU (BASELINE.json
configs[1]): Guide.embeds (CLIP text + ViT image towers, Linear image guidance tween) ->
50 DDIM steps of the SD1.5 UNet at 512x512 with classifier-free guidance 8, batch 8 per
GPU -> VAE decode -> (N > 1) RCCL all-gather of final latents and decoded images.
Weights are seeded synthetic tensors of the exact SD1.5 / CLIP ViT-L/14 architecture and
inputs are synthetic (no checkpoints or datasets exist offline).  Prints ONE JSON line.
'''
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2516.6          # MI355X dense fp16 MFMA: 256 CU x 2.4 GHz x 4096 FLOP/clk/CU
FLOPS_PER_IMAGE = 82.84e12         # BASELINE.md sec. 3, config c2 (100 UNet forwards + VAE decode)


def synth_image(seed, w, h):
    from PIL import Image
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h + 8, w + 8, 3)).astype(np.float32)
    c = np.pad(np.cumsum(np.cumsum(a, 0), 1), ((1, 0), (1, 0), (0, 0)))
    blur = ((c[8:, 8:] - c[:-8, 8:] - c[8:, :-8] + c[:-8, :-8]) / 64.0)[:h, :w]
    blur = (blur - blur.min()) / (blur.max() - blur.min()) * 255.0
    return Image.fromarray(blur.astype(np.uint8), 'RGB')


def synth_prompts(n, seed=1):
    rng = np.random.default_rng(seed)
    words = ['photo', 'turtle', 'forest', 'zeus', 'city', 'painting', 'deer', 'storm', 'neon',
             'ancient', 'river', 'portrait', 'rock', 'monkey', 'anime', 'golden', 'light']
    return [' '.join(rng.choice(words, size=int(rng.integers(5, 20)))) for _ in range(n)]


def cpu_baseline(sds, cfgs, steps, parity_args=None):
    '''Oracle (torch fp32 restatement of the reference path) timed on the host cores over a
    bounded sample: one CFG UNet evaluation (2 forwards) of ONE image at 64x64 latents plus
    one 512x512 VAE decode; extrapolated to steps x UNet + decode per image.  With
    `parity_args` the same leg also runs the oracle on the BASELINE configs[0] shape and
    reports the GPU path's PSNR against it (the oracle is the checker, never the product).'''
    from oracle import unet_ref, vae_ref
    ucfg, vcfg, _ = cfgs
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 4, 64, 64), generator=g)
    ctx = torch.randn((2, 77, ucfg.cross_attention_dim), generator=g)
    t0 = time.time()
    unet_ref.unet_forward(sds['unet'], ucfg, x, 500, ctx)
    t_unet = time.time() - t0
    t0 = time.time()
    vae_ref.vae_decode(sds['vae'], vcfg, x[:1])
    t_vae = time.time() - t0
    per_image = steps * t_unet + t_vae
    parity = parity_leg(sds, cfgs, *parity_args) if parity_args else None
    return {'parity': parity, 'value': 1.0 / per_image, 'unit': 'images/sec', 'cores': torch.get_num_threads(),
            'kind': 'port',
            'sample': f'1 CFG UNet evaluation (2 forwards, 1 image, 64x64 latents) = {t_unet:.2f} s '
                      f'and 1 VAE decode = {t_vae:.2f} s on {torch.get_num_threads()} threads '
                      f'({os.cpu_count()} cpus); extrapolated to {steps} steps + decode per image'}


def parity_leg(sds, cfgs, pipe, enc, tok, steps=10, hw=256, guidance=8.0):
    '''BASELINE configs[0] shape (256x256, 10 DDIM steps, batch 1, CFG 8) on the same SD1.5
    weights: GPU fp16 path vs the CPU fp32 oracle with identical ids and CPU-drawn noise.'''
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


# sampling stride of the per-launch HIP events: coprime with the launch pattern of a UNet step
# (220 GEMM, 33 attention, 61 GroupNorm launches), so over 50 steps every shape is visited
EVENT_STRIDE = 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU')
    ap.add_argument('--ddim-steps', type=int, default=50)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--preset', default='sd15')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the UNet from a captured HIP graph')
    args = ap.parse_args()

    import torch.distributed as dist
    from flexdiffuse_amd import dist as fdist
    rank, world, local_rank = fdist.init('nccl')
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
    pipe, clip, tok = build.build_models(sds, args.preset, dev, vae_encoder=False)
    pipe.use_graph = args.graph
    guide_ctx = Guide(clip, tok, device='cuda')
    enc = CLIPEncoder(clip, tok)
    B, N = args.batch, world
    prompts = fdist.shard(synth_prompts(B * N), rank, N, B)
    guide_img = synth_image(2, 512, 512)
    hw = args.size
    # the whole global batch of noise is drawn once on the host and sliced per rank, so the
    # images do not depend on the number of GPUs
    noise = fdist.global_noise(B * N, (4, hw // 8, hw // 8), 1337)[fdist.shard_range(rank, N, B)].to(dev)
    t_setup = time.time() - t_setup

    def one_pass():
        embeds = guide_ctx.embeds(prompt=prompts, guide=guide_img, guide_threshold_mult=0.0,
                                  guide_clustered=0.0, guide_linear=(0.0, 0.5),
                                  guide_max_guidance=0.5)
        sg = SimpleGuide(enc, pipe.unet, 8.0, args.ddim_steps, embeds)
        out = pipe(guide=sg, init_size=(hw, hw), latents=noise, output_type='np')
        # one RCCL all-gather of the final latents and of the decoded images (identity at N=1)
        all_latents = fdist.all_gather_samples(pipe.last_latents)
        all_images = fdist.all_gather_samples(pipe.last_images)
        return out, all_latents, all_images

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for w in range(args.warmup):
        hip.prof_enable(w == args.warmup - 1)   # fills the event pool outside the timed region
        one_pass()
        torch.cuda.synchronize()
        for fam in (ops.FAMILY_GEMM, ops.FAMILY_ATTENTION, ops.FAMILY_GROUPNORM):
            hip.prof_collect(fam)
        hip.prof_enable(False)

    sync()
    t0 = time.time()
    for k in range(args.steps):
        if k == args.steps - 1 and not os.environ.get('FD_BENCH_NO_EVENTS'):
            # HIP events on the launch stream around every EVENT_STRIDE-th launch of each kernel
            # family during the last timed pass (one pair per launch costs ~10 % of that pass)
            hip.prof_set_stride(EVENT_STRIDE)
            hip.prof_enable(True)
        one_pass()
    sync()
    elapsed = time.time() - t0
    hip.prof_enable(False)
    if dist.is_initialized():
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    fam = {}
    for name, code in (('gemm', ops.FAMILY_GEMM), ('attention', ops.FAMILY_ATTENTION),
                       ('groupnorm', ops.FAMILY_GROUPNORM)):
        ms, work, n = hip.prof_collect(code)
        fam[name] = {'ms': ms, 'work': work, 'launches': n}

    if rank == 0:
        images = B * N * args.steps
        value = images / elapsed
        g = fam['gemm']
        achieved = (g['work'] / (g['ms'] * 1e-3)) / 1e12 if g['ms'] > 0 else 0.0
        att = fam['attention']
        gn = fam['groupnorm']
        # HBM bytes of the dominant kernel (level-0 conv3x3) from its PMC pass: rocprofv3 cannot
        # run inside this process, so the committed per-launch measurement is reported
        traffic, traffic_of = None, None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                                   'r01_pmc_traffic.json')) as f:
                tj = json.load(f)
            traffic = tj['hbm_bytes']
            traffic_of = (f"{tj['problem']}: {tj['hbm_bytes'] / 1e6:.1f} MB per launch vs "
                          f"{tj['algorithmic_bytes'] / 1e6:.1f} MB algorithmic (profiles/r01_pmc_traffic.json)")
        except (OSError, KeyError, ValueError):
            pass
        line = {
            'metric': '512x512 50-step images/sec/node (SD1.5, batch=8/GPU, Linear image guidance)',
            'value': value, 'unit': 'images/sec', 'n_gpus': N, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'fp16',
            'data': 'synthetic',
            'config': {'workload': f'SD1.5 {hw}x{hw} {args.ddim_steps}-step DDIM + Linear image '
                                   f'guidance, CFG 8, batch={B}/GPU (BASELINE configs[1])',
                       'images_per_step': B * N, 'parallelism': f'seed-sharded x{N}',
                       'preset': args.preset},
            'roofline': {
                'bound': 'mfma', 'kernel': 'k_gemm_f16 (implicit-GEMM conv3x3 / GEMM family)',
                'achieved': achieved, 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': achieved / MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_of': traffic_of,
                # sums over the sampled launches (every EVENT_STRIDE-th of each family), scaled to
                # the pass; `achieved` is the ratio of sampled work to sampled time
                'sampled_every': EVENT_STRIDE, 'sampled_launches': g['launches'],
                'launches': g['launches'] * EVENT_STRIDE,
                'kernel_ms_per_pass': g['ms'] * EVENT_STRIDE,
                'avg_launch_us': 1e3 * g['ms'] / g['launches'] if g['launches'] else None,
                'attention_tflops': (att['work'] / (att['ms'] * 1e-3)) / 1e12 if att['ms'] else 0.0,
                'attention_ms_per_pass': att['ms'] * EVENT_STRIDE,
                'groupnorm_gbps': (gn['work'] / (gn['ms'] * 1e-3)) / 1e9 if gn['ms'] else 0.0,
                'groupnorm_ms_per_pass': gn['ms'] * EVENT_STRIDE,
                'end_to_end_frac_of_mfma_roofline':
                    (value / N) * FLOPS_PER_IMAGE / (MFMA_PEAK_TFLOPS * 1e12)
                    if (hw == 512 and args.ddim_steps == 50 and args.preset == 'sd15') else None,
            },
            'device': info, 'setup_s': t_setup,
        }
        if N == 1 and not args.no_cpu_baseline:
            sds32 = sds if args.preset != 'sd15' else sds
            pargs = (pipe, enc, tok) if (not args.no_parity and args.preset == 'sd15') else None
            line['cpu_baseline'] = cpu_baseline(sds32, cfgs, args.ddim_steps, pargs)
            line['parity'] = line['cpu_baseline'].pop('parity')
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
