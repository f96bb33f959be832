'''UNet-forward A/B over a module-level knob of flexdiffuse_amd.ops inside ONE process (both arms see the same clocks): the full-size SD1.5
forward (CFG batch 16, 64x64 latents) through the recorded launch plan, arms interleaved, N rounds of 20 forwards each.
    python tools/ab_unet_knob.py GN_FINISH_FUSE [rounds = 6]'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, hip, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
knob = sys.argv[1]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
t_dev = torch.full((1,), 400.0, device=dev)
plans = {}
for val in (True, False):
    setattr(ops, knob, val)
    unet.forward_nhwc(x, t_dev, ctx, rep=2)
    pool = torch.cuda.MemPool()
    plan = hip.Plan()
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        eps = unet.forward_nhwc(x, t_dev, ctx, rep=2)
    plans[val] = (plan, pool, eps, len(plan))
setattr(ops, knob, True)
torch.cuda.synchronize()
res = {True: [], False: []}
for r in range(rounds):
    for val in (True, False):
        plan = plans[val][0]
        for _ in range(3): plan.replay()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): plan.replay()
        torch.cuda.synchronize()
        res[val].append(1e3 * (time.time() - t0) / 20)
same = torch.equal(plans[True][2], plans[False][2])
for val in (True, False):
    v = sorted(res[val])
    print(f'{knob}={val}: {plans[val][3]} launches per forward; ms per forward {" ".join(f"{t:.3f}" for t in res[val])}; median {v[len(v) // 2]:.3f}')
print(f'outputs of the two arms bit-identical: {same}')
