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
row = []
for (M, C) in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (616, 768)]:
    x = torch.randn((M, C), device=dev).half(); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    ms = timeit(lambda: ops.layernorm(x, g, b))
    row.append(f'{ms*1e3:.1f}us/{M*C*4/ms/1e6:.0f}')
print('RPW', os.environ.get('FD_LN_RPW', '2'), ' '.join(row), flush=True)
