'''Does running the two CFG halves as two concurrent streams (two B=8 UNet forwards) hide the
kernel-boundary drain / fill bubbles of one B=16 forward?  Eager launches, 10 forwards each.'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
dev = torch.device('cuda:0')
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
cfg = build.configs('sd15')[0]
uA = UNet2DConditionModel(sds['unet'], cfg, dev)
uB = UNet2DConditionModel(sds['unet'], cfg, dev)
x = torch.randn((8, 4, 64, 64), device=dev)
ctx = torch.randn((16, 77, 768), device=dev).half()
cA, cB = ctx[:8].contiguous(), ctx[8:].contiguous()
s2 = torch.cuda.Stream()
n = 10

def joint():
    return uA.forward_nhwc(x, 400, ctx, rep=2)

def split():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        ops.WS_SLOT = 1
        b = uB.forward_nhwc(x, 400, cB, rep=1)
        ops.WS_SLOT = 0
    a = uA.forward_nhwc(x, 400, cA, rep=1)
    cur.wait_stream(s2)
    return a, b

ref = joint().float()
a, b = split()
torch.cuda.synchronize()
got = torch.cat([a, b]).float()
print('max |joint - split|', (ref - got).abs().max().item(), 'of', ref.abs().max().item(), flush=True)
for name, f in (('joint B=16', joint), ('split 2 x B=8', split), ('joint B=16', joint), ('split 2 x B=8', split)):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize()
    print(f'{name}: {1e3 * (time.time() - t0) / n:.2f} ms per CFG forward', flush=True)
def single():
    return uA.forward_nhwc(x, 400, cA, rep=1)
for _ in range(2): single()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(n): single()
torch.cuda.synchronize()
print(f'one B=8 forward alone: {1e3 * (time.time() - t0) / n:.2f} ms', flush=True)

# the same through captured graphs (no host launch cost in the picture)
def capture(f):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = f()
    return g, out
gj, _ = capture(joint)
gs, _ = capture(split)
for name, g in (('graph joint B=16', gj), ('graph split 2 x B=8', gs), ('graph joint B=16', gj), ('graph split 2 x B=8', gs)):
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    print(f'{name}: {1e3 * (time.time() - t0) / n:.2f} ms per CFG forward', flush=True)
