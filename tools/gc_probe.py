'''Are the sporadic ~50 ms stalls Python GC pauses?  Time many short GEMM loops with and without gc.'''
import sys, os, gc, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops, build
dev = torch.device('cuda:0')
# a realistic heap: the full synthetic SD1.5 state dicts + models
sds = build.synthetic_state_dicts('sd15', seed=0)
pipe, clip, tok = build.build_models(sds, 'sd15', dev, vae_encoder=False)
a = torch.randn((16384, 640), device=dev).half(); w = ops.prep_linear(torch.randn((640, 640)) * 640 ** -0.5, torch.randn(640), dev)
def loop(n=200):
    worst, tot = 0.0, 0.0
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): ops.gemm(a, w)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        worst = max(worst, dt); tot += dt
    return tot / n * 1e3, worst * 1e3
for mode in ('gc on', 'gc off', 'gc on', 'gc off'):
    if mode == 'gc off': gc.disable()
    else: gc.enable()
    mean, worst = loop()
    print(mode, f'mean {mean:.3f} ms per 30 launches, worst {worst:.2f} ms', 'gc counts', gc.get_count(), flush=True)
gc.enable()
t0 = time.perf_counter(); n = gc.collect(); print('full collection', f'{(time.perf_counter()-t0)*1e3:.1f} ms', n, 'objects tracked', len(gc.get_objects()))
