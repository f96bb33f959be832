'''Deep-level convolutions (16x16 and 8x8 maps at CFG batch 16): (tile, split-K) A/B incl. the 3-stage 256x160 tile (24).
   python tools/ab_deepconv.py "0:0 13:8 24:8 24:4 20:4"   (tile:split pairs; 0:0 = the library's rule)'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
arms = [tuple(int(v) for v in a.split(':')) for a in (sys.argv[1] if len(sys.argv) > 1 else '0:0 13:8 24:8 24:4').split()]
def timeit(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 640, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280)]
print('arms (tile:split):', arms, ' shapes (B, H, Cin, Cout):', shapes)
g = torch.Generator().manual_seed(0)
data = []
for (B, H, Cin, Cout) in shapes:
    x = ops.Act((torch.randn((B * H * H, Cin), generator=g) * 0.7).half().to(dev), B, H, H)
    w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5, torch.randn(Cout, generator=g), dev)
    data.append((x, w))
for rep in range(2):
    for (t, s) in arms:
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, s
        row = []
        for (B, H, Cin, Cout), (x, w) in zip(shapes, data):
            try:
                ms = timeit(lambda: ops.conv2d(x, w))
                row.append(f'{ms * 1e3:6.1f}us/{2 * B * H * H * Cout * 9 * Cin / ms / 1e9:5.0f}')
            except Exception as ex:
                row.append('   refused   ')
        print(f'tile {t:2d} split {s}:', ' '.join(row), flush=True)
ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
