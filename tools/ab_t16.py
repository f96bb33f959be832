'''Tile 16 (256x320, 64x80 wave tiles) vs 13 on the shapes where it applies; run once per
library variant (FD_LIB_PATH).'''
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
shapes = [(16,64,320,320),(16,32,1280,1280),(16,32,640,640),(16,32,1280,640),(16,32,1920,640),(16,32,320,640),(16,16,1280,1280),(16,16,2560,1280),(16,16,640,1280)]
split = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else '13,16').split(',')]
for t in tiles:
    row = []
    for (B, H, Cin, Cout) in shapes:
        x = ops.Act(torch.randn((B * H * H, Cin), device=dev).half(), B, H, H)
        w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5, torch.randn(Cout), dev)
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, (split if (H <= 16 and t) else 1 if t else 0)
        ms = timeit(lambda: ops.conv2d(x, w))
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
        row.append(f'{ms*1e3:.0f}/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    print(os.environ.get('FD_LIB_PATH', 'default')[-14:], 'tile', t, ' '.join(row), flush=True)
