'''Short-K linear GEMMs (transformer projections): tile A/B.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=400):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
lin = [(65536,320,320),(65536,640,320),(65536,320,1280),(16384,640,640),(16384,1280,640),(16384,640,2560),(4096,1280,1280),(4096,2560,1280),(4096,1280,5120)]
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else '0,13,12,16').split(',')]
res = len(sys.argv) > 2
for t in tiles * 2:
    row = []
    for (M, N, K) in lin:
        a = torch.randn((M, K), device=dev).half(); w = ops.prep_linear(torch.randn((N, K)) * K ** -0.5, torch.randn(N), dev)
        r = torch.randn((M, N), device=dev).half() if res else None
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, (1 if t else 0)
        try:
            ms = timeit(lambda: ops.gemm(a, w, residual=r))
        except Exception as e:
            ms = float('nan')
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
        row.append(f'{ms*1e3:.0f}/{2*M*N*K/ms/1e9:.0f}')
    print(os.environ.get('FD_LIB_PATH', 'default')[-14:], 'tile', t, ' '.join(row), flush=True)
