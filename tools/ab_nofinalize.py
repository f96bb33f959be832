'''Upper bound of what folding the LayerNorm-statistics finalise launches (fd_ln_finalize_stats_f32: 33 per UNet forward at the
32x32 / 16x16 / 8x8 levels) into their consumers could return: the CFG-batch-16 forward with the finalise launches SKIPPED (a stale
statistics buffer is handed to the consumers: timing only, results wrong) against the normal forward, one process, interleaved.
Eager launches are host-bound on a slow host, so both arms run from a recorded launch plan.'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, hip, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
t_dev = torch.full((1,), 400.0, device=dev)
real = ops.ln_finalize_stats
cache = {}
def fake(parts, N, eps=1e-5):
    k, M, _ = parts.shape
    if M not in cache:
        cache[M] = real(parts, N, eps)
    return cache[M]
plans = {}
for name, fn in (('normal', real), ('no finalise launches', fake)):
    ops.ln_finalize_stats = fn
    unet.forward_nhwc(x, t_dev, ctx, rep=2)
    pool = torch.cuda.MemPool()
    plan = hip.Plan()
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        unet.forward_nhwc(x, t_dev, ctx, rep=2)
    plans[name] = (plan, pool, len(plan))
ops.ln_finalize_stats = real
for rep in range(3):
    for name, (plan, pool, n) in plans.items():
        for _ in range(3): plan.replay()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(30): plan.replay()
        torch.cuda.synchronize()
        print(f'{name:24s} {n} launches  {1e3 * (time.time() - t0) / 30:.3f} ms per CFG forward', flush=True)
