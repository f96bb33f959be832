'''Level-2/3 convs (M = 4096 / 1024, N = 1280): tile/split A/B incl. the 256x256 tile with odd splits.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(16,16,640,1280),(16,16,1280,1280),(16,16,1920,1280),(16,16,2560,1280),(16,8,1280,1280),(16,8,2560,1280),(16,32,1280,1280)]
for (t, sp) in [(0, 0), (13, 2), (15, 3), (15, 4), (15, 2), (15, 6), (15, 12), (15, 1)]:
    row = []
    for (B, H, Cin, Cout) in shapes:
        x = ops.Act(torch.randn((B * H * H, Cin), device=dev).half(), B, H, H)
        w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5, torch.randn(Cout), dev)
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, sp
        try:
            ms = timeit(lambda: ops.conv2d(x, w))
        except Exception as e:
            ms = float('nan')
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
        row.append(f'{ms*1e3:.0f}/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    print('tile', t, 'split', sp, ' '.join(row), flush=True)
