'''Timing-only ceiling for "LayerNorm statistics from the producers at 640 / 1280 channels": the UNet forward with
fd_ln_row_stats_f16 replaced by a cached result of the same shape (stale statistics: numerically plausible, not correct).'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
x = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
real = ops.ln_row_stats
cache, calls = {}, [0]
def cached(h, *a, **k):
    calls[0] += 1
    key = tuple(h.shape)
    if key not in cache:
        cache[key] = real(h, *a, **k)
    return cache[key]
def run(n=20):
    for _ in range(3): unet.forward_nhwc(x, 400, ctx, rep=2)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): unet.forward_nhwc(x, 400, ctx, rep=2)
    torch.cuda.synchronize()
    return 1e3 * (time.time() - t0) / n
for rep in range(3):
    ops.ln_row_stats = real
    a = run()
    ops.ln_row_stats = cached; calls[0] = 0
    b = run()
    print(f'real statistics {a:.3f} ms | cached (no launch) {b:.3f} ms | {calls[0] / 23:.0f} statistics launches per forward', flush=True)
