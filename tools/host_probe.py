'''How far ahead of the device is the host?  Issue 20 CFG UNet evaluations back to back and compare
the time at which Python returns from the last launch with the time at which the device is done.'''
import sys, os, gc, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0)
pipe, clip, tok = build.build_models(sds, 'sd15', dev, vae_encoder=False)
B = 8
x = torch.randn((B, 4, 64, 64), device=dev)
ctx = torch.randn((2 * B, 77, 768), device=dev).half()
unet = pipe.unet
for _ in range(3): unet.forward_nhwc(x, 500, ctx, rep=2)
torch.cuda.synchronize()
gc.disable()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(20): unet.forward_nhwc(x, 500 - i, ctx, rep=2)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'issue {1e3*(t1-t0)/20:.2f} ms / forward, device done {1e3*(t2-t0)/20:.2f} ms / forward, host lead at the end {1e3*(t2-t1):.1f} ms', flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): unet.forward_nhwc(x, 400 - i, ctx, rep=2)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
