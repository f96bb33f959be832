'''8x8 / 16x16-resolution conv3x3 (M = 1024 / 4096, N = 1280): tile x split-K sweep incl. the finish kernel.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, H, Cin, Cout) in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (16, 16, 1280, 1280), (16, 16, 2560, 1280)]:
    x = ops.Act(torch.randn((B*H*H, Cin), device=dev).half(), B, H, H)
    w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9*Cin) ** -0.5, torch.randn(Cout), dev)
    b2 = torch.randn((B, Cout), device=dev)
    row = []
    for (t, sp) in [(0, 0), (13, 8), (13, 4), (13, 2), (9, 8), (9, 4), (9, 2), (12, 4), (12, 2), (20, 2), (20, 4), (10, 4), (10, 2), (14, 4), (14, 8), (3, 2), (3, 4)]:
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, sp
        try: us = timeit(lambda: ops.conv2d(x, w, bias2=b2, ld_bias2=Cout))
        except Exception as e: us = float('nan')
        row.append(f'({t},{sp}):{us:.0f}')
    ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
    print((B, H, Cin, Cout), ' '.join(row), flush=True)
