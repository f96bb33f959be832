'''Is the conv / GEMM loop power-limited?  Same kernels on random fp16 data vs all-ones data
(lower toggle rate -> higher sustained clock when the chip is at its power limit).'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=80):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
# AB_TILES=tile:split,tile:split,tile:split forces the three convolutions' tiles (e.g. the ping-pong tiles 30:1,32:1,30:4)
forced = [tuple(int(v) for v in t.split(':')) for t in os.environ['AB_TILES'].split(',')] if os.environ.get('AB_TILES') else None
for rep in range(2):
    for kind in ('random', 'ones', 'zeros'):
        row = []
        for ci, (B, H, Cin, Cout) in enumerate([(16,64,320,320),(16,32,640,640),(16,16,1280,1280)]):
            ops.FORCE_TILE, ops.FORCE_SPLIT = forced[ci] if forced else (0, 0)
            mk = (lambda *s: torch.randn(*s)) if kind == 'random' else (lambda *s: torch.ones(*s)) if kind == 'ones' else (lambda *s: torch.zeros(*s))
            x = ops.Act(mk(B*H*H, Cin).half().to(dev), B, H, H)
            w = ops.prep_conv(mk(Cout, Cin, 3, 3) * (0.02 if kind == 'random' else 1e-3), torch.zeros(Cout), dev)
            us = timeit(lambda: ops.conv2d(x, w))
            row.append(f'conv({B},{H},{Cin},{Cout}) {us:.0f}us {2*B*H*H*Cout*9*Cin/us/1e6:.0f}TF')
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
        M, C = 65536, 320
        a = (torch.randn(M, C) if kind == 'random' else torch.ones(M, C) if kind == 'ones' else torch.zeros(M, C)).half().to(dev)
        wg = ops.prep_geglu((torch.randn(8*C, C) if kind == 'random' else torch.ones(8*C, C)) * C ** -0.5, torch.zeros(8*C), dev)
        us = timeit(lambda: ops.gemm(a, wg, act=ops.ACT_GEGLU))
        row.append(f'geglu {us:.0f}us')
        wl = ops.prep_linear((torch.randn(C, C) if kind == 'random' else torch.ones(C, C)) * C ** -0.5, torch.zeros(C), dev)
        us = timeit(lambda: ops.gemm(a, wl), 300)
        row.append(f'lin320 {us:.1f}us')
        print(kind, ' | '.join(row), flush=True)
