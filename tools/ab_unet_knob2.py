'''Like tools/ab_unet_knob.py for several knobs at once: arms = every listed ops.<KNOB> True / False combination named on the command line as
"KNOB=0/1,KNOB2=0/1" ("-" = defaults); one process, plans recorded per arm, interleaved rounds.
    python tools/ab_unet_knob2.py - GN_PARTS=0 GN_PARTS=0,GN_FINISH_FUSE=0'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, hip, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
arms = sys.argv[1:] or ['-']
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
t_dev = torch.full((1,), 400.0, device=dev)
plans = {}
for arm in arms:
    saved = {}
    temb = None
    if arm != '-':
        for kv in arm.split(','):
            k, v = kv.split('=')
            if k == 'TEMB':      # the ResBlocks' time biases precomputed (FlexPipeline does this once per request): no time-embedding GEMMs in the forward
                temb = unet.time_bias(400.0, 1).expand(16, -1).contiguous() if v != '0' else None
                continue
            saved[k] = getattr(ops, k)
            setattr(ops, k, v != '0')
    unet.forward_nhwc(x, t_dev, ctx, rep=2, temb=temb)
    pool = torch.cuda.MemPool(); plan = hip.Plan()
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        eps = unet.forward_nhwc(x, t_dev, ctx, rep=2, temb=temb)
    plans[arm] = (plan, pool, eps, len(plan))
    for k, v in saved.items(): setattr(ops, k, v)
torch.cuda.synchronize()
res = {a: [] for a in arms}
for r in range(6):
    for a in arms:
        plan = plans[a][0]
        for _ in range(3): plan.replay()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): plan.replay()
        torch.cuda.synchronize()
        res[a].append(1e3 * (time.time() - t0) / 20)
base = plans[arms[0]][2]
for a in arms:
    v = sorted(res[a])
    d = float((plans[a][2].float() - base.float()).abs().max())
    print(f'{a}: {plans[a][3]} launches per forward; ms per forward {" ".join(f"{t:.3f}" for t in res[a])}; median {0.5 * (v[2] + v[3]):.3f}; max |eps - eps(first arm)| {d:.3g}')
