'''Guide.embeds (tokenizer + CLIP text tower + ViT-L/14 image tower + tween) per request, wall clock with a device sync: the towers through their launch
plans (default) against the eager Python front (FD_CLIP_PLAN=0), interleaved.  python tools/ab_embeds.py'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from flexdiffuse_amd import Guide, build
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('clip', 'unet', 'vae'))
pipe, clip, tok = build.build_models(sds, 'sd15', dev, vae_encoder=False)
g = Guide(clip, tok, device='cuda')
prompts = bench.synth_prompts(8); img = bench.synth_image(2, 512, 512)


def once():
    torch.cuda.synchronize(); t0 = time.time()
    e = g.embeds(prompt=prompts, guide=img, **bench.GUIDANCE['linear'])
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0), e


res = {'1': [], '0': []}
outs = {}
for rnd in range(6):
    for arm in ('1', '0'):
        os.environ['FD_CLIP_PLAN'] = arm
        once()
        ts = []
        for _ in range(5):
            t, e = once()
            ts.append(t)
        res[arm].append(min(ts))
        outs[arm] = e
del os.environ['FD_CLIP_PLAN']
for arm, name in (('1', 'towers from launch plans'), ('0', 'eager front')):
    v = sorted(res[arm])
    print(f'Guide.embeds, 8 prompts + 1 guide image, {name}: {" ".join(f"{t:.2f}" for t in res[arm])} ms -> median {0.5 * (v[2] + v[3]):.2f} ms')
print('outputs bit-identical:', torch.equal(outs['1'], outs['0']))
