'''Oracle side of a FULL-SIZE CompositeGuide run (reference composition/guide.py:32-139, utils.py:168-207) -- CPU only.

SD1.5 architecture with the seeded synthetic weights, 512x512, 20 DDIM steps, CFG 8, a background prompt and two entity boxes (one of them
clipped by the canvas): every step is ONE UNet batch over [uncond | background | entity 1 | entity 2], the rectangular latent blend
`bg + blend * (entity - bg)` per box and the CFG combine (oracle/sched_ref.composite_noise_pred, pinned on the reference's own
CompositeGuide by tests/golden/backhalf_goldens.npz), then a DDIM step.  The final latents are cached as data in
tests/golden/composite_oracle.npz so that the 80 fp32 UNet forwards (~9 CPU-minutes on 8 threads) are not repeated on every GPU box;
tests/test_gpu_models.py::test_sd15_composite_guide_full_size_psnr decodes them with the oracle VAE and compares the device image.

Usage:  python tests/golden/make_composite_oracle.py [--threads 8]
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

COMPOSITE = dict(size=512, steps=20, guidance=8.0, seed=9, background='a forest at dawn, oil painting',
                 # (prompt, (offset x, y) px, (width, height) px, blend)
                 entities=[('a deer standing in tall grass', (32, 192), (256, 288), 0.8),
                           ('a red bird on a branch', (352, 64), (224, 160), 0.5)])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    from flexdiffuse_amd import build
    from flexdiffuse_amd.tokenizer import SyntheticTokenizer
    from oracle import clip_ref, ddim_ref, sched_ref, unet_ref
    c = COMPOSITE
    sds = build.synthetic_state_dicts('sd15', seed=0)
    ucfg, vcfg, ccfg = build.configs('sd15')
    tok = SyntheticTokenizer(vocab_size=ccfg.text.vocab_size, model_max_length=ccfg.text.max_position_embeddings)
    text_sd = {k: v for k, v in sds['clip'].items() if k.startswith('text_model')}
    th = lambda p: clip_ref.text_hidden(text_sd, ccfg, tok(p).input_ids)
    ents = [(th(p), tuple(v // 8 for v in off), tuple(v // 8 for v in size), blend) for p, off, size, blend in c['entities']]
    uncond, bg = th(''), th(c['background'])
    h = c['size'] // 8
    x = torch.randn((1, 4, h, h), generator=torch.Generator('cpu').manual_seed(c['seed']))
    sha = hashlib.sha256(x.numpy().tobytes()).digest()
    acp = ddim_ref.alphas_cumprod()
    used, t0 = [], time.time()
    for t in ddim_ref.timesteps(c['steps']):
        fn = lambda lat, emb: unet_ref.unet_forward(sds['unet'], ucfg, lat, int(t), emb)
        eps = sched_ref.composite_noise_pred(fn, x, uncond, bg, ents, c['guidance'])
        x = ddim_ref.ddim_step(eps, int(t), x, acp, c['steps'])
        used.append(int(t))
        print(f'  t={int(t)} |x|max={float(x.abs().max()):.3f} ({time.time() - t0:.0f} s)', flush=True)
    out = os.path.join(HERE, 'composite_oracle.npz')
    np.savez_compressed(out, latents=x.numpy().astype(np.float32), timesteps=np.array(used, dtype=np.int64),
                        noise_sha=np.frombuffer(sha, dtype=np.uint8), cpu_seconds=np.array([time.time() - t0]),
                        threads=np.array([torch.get_num_threads()]))
    print(f'wrote {out}: {len(used)} steps x {2 + len(ents)} UNet samples in {time.time() - t0:.0f} s; latents std {float(x.std()):.3f}')


if __name__ == '__main__':
    main()
