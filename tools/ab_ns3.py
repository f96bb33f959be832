'''3 LDS stages (2 K-tiles in flight) for the 128x160 tiles on the low-resolution linear GEMMs:
tiles 12 / 9 (2 stages) vs 19 / 20 (3 stages); also checks the result against torch.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(4096,1280,1280),(4096,1280,5120),(4096,1280,2560),(16384,640,640),(16384,640,2560),(16384,1280,640),(1024,1280,1280),(1024,1280,5120)]
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else '0,12,9,20').split(',')]
data = []
for (M, N, K) in shapes:
    a = torch.randn((M, K), device=dev).half(); w = ops.prep_linear(torch.randn((N, K)) * K ** -0.5, torch.randn(N), dev)
    r = torch.randn((M, N), device=dev).half()
    ref = (a[:256].float() @ w.w.float().t() + w.bias[:N] + r[:256].float())
    data.append((a, w, r, ref))
for t in tiles * 2:
    row = []
    for (M, N, K), (a, w, r, ref) in zip(shapes, data):
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, (1 if t else 0)
        try:
            out = ops.gemm(a, w, residual=r)
            err = float((out[:256].float() - ref).abs().max() / ref.abs().max())
            us = timeit(lambda: ops.gemm(a, w, residual=r))
            row.append(f'{us:.1f}' + ('' if err < 5e-3 else f'(ERR {err:.3f})'))
        except Exception as e:
            row.append('nan')
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
    print('tile', t, ' '.join(row), flush=True)
