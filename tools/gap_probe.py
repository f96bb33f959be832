'''Device idle time inside a UNet forward: run under `rocprofv3 --kernel-trace` and compare the sum
of kernel durations with the span from the first kernel start to the last kernel end.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0)
pipe, clip, tok = build.build_models(sds, 'sd15', dev, vae_encoder=False)
B = 8
x = torch.randn((B, 4, 64, 64), device=dev)
ctx = torch.randn((2 * B, 77, 768), device=dev).half()
for _ in range(3): pipe.unet.forward_nhwc(x, 500, ctx, rep=2)
torch.cuda.synchronize()
import time; time.sleep(0.5)     # a visible gap in the trace before the measured block
for i in range(10): pipe.unet.forward_nhwc(x, 400 - i, ctx, rep=2)
torch.cuda.synchronize()
