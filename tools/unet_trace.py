'''Ten UNet forwards at the headline shape (CFG batch 16, 64x64 latents) for
`rocprofv3 --kernel-trace --stats -- python3 tools/unet_trace.py` (per-kernel time of a step).'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
from flexdiffuse_amd.unet import UNet2DConditionModel
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
B = 8
x = torch.randn((B, 4, 64, 64), device=dev)
ctx = torch.randn((2 * B, 77, 768), device=dev).half()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(n + 2):
    unet.forward_nhwc(x, 400 - i, ctx, rep=2)
torch.cuda.synchronize()
