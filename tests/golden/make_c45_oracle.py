'''Oracle side of BASELINE configs[3] (c4) and configs[4] (c5) at batch 1 -- CPU only.

c4: SD1.5 architecture, img2img + Linear image guidance at 768x768 (96x96 latents), 50 DDIM
    steps, strength 0.6 => the 30 UNet evaluations of timesteps[20:], CFG 8
    (reference pipeline/flex.py:181-221 init, :262-287 loop, pipeline/guide.py:46-64 CFG).
c5: SD2.1-size UNet (v-prediction, linear projections, head dim 64, context 1024) + OpenCLIP
    ViT-H/14 guide, 768x768 txt2img, 50 DDIM steps, CFG 8, Linear image guidance.  The reference
    has no behaviour for c5 (it hard-codes SD-v1-4 + CLIP-L, utils.py:24-25): the parity target
    is this restatement.

Same pattern as make_c2_oracle.py: the fp32 CPU oracle runs the WHOLE path of sample 0 of
bench.py's workload (GuideRef.embeds -> [img2img_init] -> denoise) and stores the guided
embeddings, the executed timestep list and the final latents as DATA in
tests/golden/c4_oracle.npz / c5_oracle.npz, so the 60 / 100 fp32 UNet forwards at 96x96
(~20 / ~35 CPU-minutes) are not repeated on every GPU box.  The GPU tests
(tests/test_gpu_models.py::test_sd15_c4_img2img_psnr, ::test_sd21_c5_psnr) decode those latents
with the oracle VAE and compare the device image against it.

The oracle is `parity unpinned` for the UNet / VAE / DDIM part (diffusers 0.3.0 is absent,
oracle/__init__.py); this file pins nothing new, it only caches the oracle's own output.

Usage:  python tests/golden/make_c45_oracle.py c4|c5 [--threads 6]
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

LINEAR = dict(guide_threshold_mult=0.0, guide_clustered=0.0, guide_linear=(0.0, 0.5),
              guide_max_guidance=0.5)
CONFIGS = {
    'c4': dict(preset='sd15', size=768, steps=50, guidance=8.0, batch=4, strength=0.6,
               init_seed=3, guide_seed=2, gen_seed=1337, embeds_kw=LINEAR),
    'c5': dict(preset='sd21', size=768, steps=50, guidance=8.0, batch=8, strength=None,
               guide_seed=2, noise_seed=1337, embeds_kw=LINEAR),
}


def init_tensor(seed: int, size: int) -> torch.Tensor:
    '''The img2img init image as the (1,3,H,W) fp32 tensor in [-1,1] the pipeline takes when it is
    not handed a PIL image (pipeline/flex.py:182-184; `preprocess` would resize a PIL image's long
    side to 512, encode/clip.py:15-39, so a 768x768 run passes the tensor).'''
    import bench
    a = np.asarray(bench.synth_image(seed, size, size), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1)[None].contiguous() * 2.0 - 1.0


def inputs(name: str):
    '''Sample 0 of bench.py's workload for the config: dict with prompt, guide image and either
    (init image tensor, posterior noise, noise) for c4 or the initial latents for c5.'''
    import bench
    from flexdiffuse_amd import dist as fdist
    c = CONFIGS[name]
    out = dict(prompt=bench.synth_prompts(c['batch'])[0],
               guide=bench.synth_image(c['guide_seed'], 512, 512))
    h = c['size'] // 8
    if c['strength'] is not None:
        out['init'] = init_tensor(c['init_seed'], c['size'])
        # the pipeline draws the posterior sample first, then the noise (pipeline/flex.py:189-213),
        # both from the caller's generator; batch 1
        g = torch.Generator('cpu').manual_seed(c['gen_seed'])
        out['posterior_noise'] = torch.randn((1, 4, h, h), generator=g)
        out['noise'] = torch.randn((1, 4, h, h), generator=g)
    else:
        out['lat0'] = fdist.global_noise(c['batch'], (4, h, h), c['noise_seed'])[:1].clone()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('config', choices=sorted(CONFIGS))
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    c = CONFIGS[args.config]
    out_path = args.out or os.path.join(HERE, f'{args.config}_oracle.npz')
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import guide_ref, pipeline_ref
    sds = build.synthetic_state_dicts(c['preset'], seed=0)
    ucfg, vcfg, ccfg = build.configs(c['preset'])
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size,
                             model_max_length=ccfg.text.max_position_embeddings)
    inp = inputs(args.config)
    t0 = time.time()
    g = guide_ref.GuideRef(sds['clip'], ccfg, tok)
    embeds = g.embeds(prompt=inp['prompt'], guide=inp['guide'], **c['embeds_kw'])
    uncond = g.prompt('')
    del g
    t_embed = time.time() - t0
    t0 = time.time()
    extra = {}
    if c['strength'] is not None:
        lat0, t_start = pipeline_ref.img2img_init(sds['vae'], vcfg, inp['init'], inp['posterior_noise'],
                                                  inp['noise'], c['steps'], c['strength'], 1)
        extra['noisy_init'] = lat0.numpy().astype(np.float32)
    else:
        lat0, t_start = inp['lat0'], 0
    sha = hashlib.sha256((inp['noise'] if 'noise' in inp else lat0).numpy().tobytes()).digest()
    t_init = time.time() - t0
    t0 = time.time()
    lat, used = pipeline_ref.denoise(sds['unet'], ucfg, embeds, uncond, lat0, c['steps'], c['guidance'],
                                     t_start=t_start,
                                     callback=lambda t, x: print(f'  t={t} |x|max={float(x.abs().max()):.3f} '
                                                                 f'({time.time() - t0:.0f} s)', flush=True))
    t_loop = time.time() - t0
    np.savez_compressed(
        out_path, latents=lat.numpy().astype(np.float32), timesteps=np.array(used, dtype=np.int64),
        embeds=embeds.numpy().astype(np.float32), uncond=uncond.numpy().astype(np.float32),
        noise_sha=np.frombuffer(sha, dtype=np.uint8), steps=np.array([c['steps']]),
        size=np.array([c['size']]), t_start=np.array([t_start]),
        cpu_seconds=np.array([t_embed, t_init, t_loop]), threads=np.array([torch.get_num_threads()]),
        prompt=np.array(inp['prompt']), **extra)
    print(f'wrote {out_path}: embeds {t_embed:.1f} s, init {t_init:.1f} s, {len(used)}-evaluation loop '
          f'{t_loop:.1f} s on {torch.get_num_threads()} threads; latents std {float(lat.std()):.3f}')


if __name__ == '__main__':
    main()
