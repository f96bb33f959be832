'''One GEMM shape, library choice vs forced (tile, split) candidates, interleaved rounds (settles sweep rows where "auto" and the
same forced tile disagree).  python tools/ab_tile_one.py M N K K2 res tile:split [tile:split ...]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
M, N, K, K2, res = (int(v) for v in sys.argv[1:6])
cands = [(0, 0)] + [tuple(int(v) for v in c.split(':')) for c in sys.argv[6:]]
g = torch.Generator().manual_seed(0)
a = (torch.randn((M, K), generator=g)).half().to(dev)
a2 = torch.randn((M, K2), generator=g).half().to(dev) if K2 else None
w = ops.prep_linear(torch.randn((N, K + K2), generator=g) * (K + K2) ** -0.5, torch.randn(N, generator=g), dev)
r = torch.randn((M, N), generator=g).half().to(dev) if res else None
out = torch.empty((M, N), dtype=torch.float16, device=dev)


def timeit(n=20):
    for _ in range(3):
        ops.gemm(a, w, a2=a2, residual=r, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        ops.gemm(a, w, a2=a2, residual=r, out=out)
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for rnd in range(3):
    row = []
    for t, s in cands:
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, s
        try:
            row.append(f'({t},{s}) {timeit():6.1f}')
        except Exception as e:
            row.append(f'({t},{s}) refused')
    print('  '.join(row), flush=True)
