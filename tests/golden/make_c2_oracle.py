'''Oracle side of the headline configuration (BASELINE configs[1]) at batch 1 -- CPU only.

SD1.5 architecture with the seeded synthetic weights, 512x512, 50 DDIM steps, CFG 8, Linear
image guidance (0.0 -> 0.5, threshold / clustered off, max 0.5): the fp32 CPU oracle runs
the WHOLE path -- `oracle.guide_ref.GuideRef.embeds` (CLIP text + ViT-L/14 towers, map +
tween; reference guidance.py:337-474) -> `oracle.pipeline_ref.denoise` (reference
pipeline/flex.py:262-287 x 50, pipeline/guide.py:46-64) -- and stores the guided embeddings,
the executed timestep list and the final latents as data in tests/golden/c2_oracle.npz.
The GPU test (tests/test_gpu_models.py::test_sd15_c2_headline_psnr) and bench.py's parity leg
decode those latents with the oracle VAE and compare the device path's image against it, so
the 100 fp32 UNet forwards (~13 min on 8 cores) are not repeated on every GPU box.

The oracle is `parity unpinned` for the UNet / VAE / DDIM part (diffusers 0.3.0 is absent,
oracle/__init__.py); this file pins nothing new, it only caches the oracle's own output.

`--guidance clustered_threshold` writes c3_oracle.npz: the same sample under BASELINE configs[2]'s guidance
parameters (Clustered 0.25 + Threshold (0.75, 0.25), linear off, max 0.35, header cap 0: bench.py GUIDANCE).

`--scheduler pndm|lms` writes c2_pndm_oracle.npz / c2_lms_oracle.npz: the headline sample under the scheduler the reference's harness
really passes (SD-v1-4 ships PNDM, utils.py:70; PLMS: 51 UNet evaluations for 50 steps) and under K-LMS (pipeline/flex.py:236-238, 270-274
sigma scaling), through oracle/sched_ref.py; the guided embeddings are taken from c2_oracle.npz.

Usage:  python tests/golden/make_c2_oracle.py [--steps 50] [--size 512] [--guidance linear|clustered_threshold] [--scheduler ddim|pndm|lms]
'''
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

C2 = dict(prompt_index=0, guide_seed=2, noise_seed=1337, guidance=8.0,
          embeds_kw=dict(guide_threshold_mult=0.0, guide_clustered=0.0, guide_linear=(0.0, 0.5),
                         guide_max_guidance=0.5))


def c2_inputs(size: int = 512):
    '''(prompt, guide image, initial latents) of sample 0 of bench.py's workload.'''
    import bench
    from flexdiffuse_amd import dist as fdist
    prompt = bench.synth_prompts(8)[C2['prompt_index']]
    img = bench.synth_image(C2['guide_seed'], 512, 512)
    lat0 = fdist.global_noise(8, (4, size // 8, size // 8), C2['noise_seed'])[:1].clone()
    return prompt, img, lat0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--guidance', default='linear', choices=['linear', 'clustered_threshold'])
    ap.add_argument('--scheduler', default='ddim', choices=['ddim', 'pndm', 'lms'])
    ap.add_argument('--out', default=None)
    ap.add_argument('--threads', type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    if args.scheduler != 'ddim':
        return other_scheduler(args)
    import bench
    embeds_kw = C2['embeds_kw'] if args.guidance == 'linear' else bench.GUIDANCE[args.guidance]
    if args.out is None:
        args.out = os.path.join(HERE, 'c2_oracle.npz' if args.guidance == 'linear' else 'c3_oracle.npz')
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import guide_ref, pipeline_ref
    sds = build.synthetic_state_dicts('sd15', seed=0)
    ucfg, vcfg, ccfg = build.configs('sd15')
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size,
                             model_max_length=ccfg.text.max_position_embeddings)
    prompt, img, lat0 = c2_inputs(args.size)
    t0 = time.time()
    g = guide_ref.GuideRef(sds['clip'], ccfg, tok)
    embeds = g.embeds(prompt=prompt, guide=img, **embeds_kw)
    text = g.prompt(prompt)
    uncond = g.prompt('')
    t_embed = time.time() - t0
    t0 = time.time()
    lat, used = pipeline_ref.denoise(sds['unet'], ucfg, embeds, uncond, lat0, args.steps,
                                     C2['guidance'])
    t_loop = time.time() - t0
    np.savez_compressed(
        args.out, latents=lat.numpy().astype(np.float32), timesteps=np.array(used, dtype=np.int64),
        embeds=embeds.numpy().astype(np.float32), text=text.numpy().astype(np.float32),
        lat0_sha=np.frombuffer(hashlib.sha256(lat0.numpy().tobytes()).digest(), dtype=np.uint8),
        steps=np.array([args.steps]), size=np.array([args.size]),
        cpu_seconds=np.array([t_embed, t_loop]), threads=np.array([torch.get_num_threads()]),
        prompt=np.array(prompt))
    print(f'wrote {args.out}: embeds {t_embed:.1f} s, {args.steps}-step loop {t_loop:.1f} s on '
          f'{torch.get_num_threads()} threads; latents std {float(lat.std()):.3f}')


def other_scheduler(args):
    '''The headline sample (Linear guidance) under PNDM / K-LMS: embeddings from c2_oracle.npz, loop from oracle/sched_ref.py.'''
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import clip_ref, pipeline_ref, sched_ref
    assert args.guidance == 'linear' and args.size == 512
    out = args.out or os.path.join(HERE, f'c2_{args.scheduler}_oracle.npz')
    sds = build.synthetic_state_dicts('sd15', seed=0)
    ucfg, vcfg, ccfg = build.configs('sd15')
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size, model_max_length=ccfg.text.max_position_embeddings)
    o = np.load(os.path.join(HERE, 'c2_oracle.npz'))
    prompt, img, lat0 = c2_inputs(args.size)
    assert hashlib.sha256(lat0.numpy().tobytes()).digest() == o['lat0_sha'].tobytes() and str(o['prompt']) == prompt
    embeds = torch.from_numpy(o['embeds'])
    text_sd = {k: v for k, v in sds['clip'].items() if k.startswith('text_model')}
    uncond = clip_ref.text_hidden(text_sd, ccfg, tok('').input_ids)
    n = [0]
    t0 = time.time()

    def eps_fn(x, t):
        n[0] += 1
        print(f'  evaluation {n[0]} t={float(t):.1f} |x|max={float(x.abs().max()):.3f} ({time.time() - t0:.0f} s)', flush=True)
        return pipeline_ref.noise_pred(sds['unet'], ucfg, x, t, embeds, uncond, C2['guidance'])
    if args.scheduler == 'pndm':
        lat, used = sched_ref.pndm_loop(eps_fn, lat0, args.steps)
        extra = dict(timesteps=np.array(used, dtype=np.int64))
    else:
        lat, sigmas = sched_ref.lms_loop(eps_fn, lat0, args.steps)
        extra = dict(sigmas=np.asarray(sigmas, dtype=np.float64))
    np.savez_compressed(out, latents=lat.numpy().astype(np.float32), evaluations=np.array([n[0]]),
                        lat0_sha=np.frombuffer(hashlib.sha256(lat0.numpy().tobytes()).digest(), dtype=np.uint8),
                        steps=np.array([args.steps]), size=np.array([args.size]), cpu_seconds=np.array([time.time() - t0]),
                        threads=np.array([torch.get_num_threads()]), prompt=np.array(prompt), **extra)
    print(f'wrote {out}: {n[0]} CFG evaluations in {time.time() - t0:.0f} s; latents std {float(lat.std()):.3f}')


if __name__ == '__main__':
    main()
