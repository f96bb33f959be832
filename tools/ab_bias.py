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
for (M, N, K) in [(65536,320,320),(65536,320,1280),(16384,640,640),(4096,1280,1280),(1024,1280,1280)]:
    a = torch.randn((M, K), device=dev).half(); w = ops.prep_linear(torch.randn((N, K)) * K ** -0.5, torch.randn(N), dev)
    r = torch.randn((M, N), device=dev).half()
    row = []
    for ub in (True, False):
        row.append(f'{timeit(lambda: ops.gemm(a, w, use_bias=ub))*1e3:.1f}')
    row.append(f'res {timeit(lambda: ops.gemm(a, w, residual=r))*1e3:.1f}')
    print((M, N, K), 'bias / no bias us:', ' '.join(row), flush=True)
for (M, C) in [(65536, 320), (16384, 640), (4096, 1280)]:
    a = torch.randn((M, C), device=dev).half(); w = ops.prep_geglu(torch.randn((8 * C, C)) * C ** -0.5, torch.randn(8 * C), dev)
    print('geglu', (M, C), f'{timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))*1e3:.1f}', flush=True)
