import sys, os; sys.path.insert(0,'/root/repo')
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
out=[]
for (M,C) in [(65536,320),(16384,640),(4096,1280),(1024,1280)]:
    a = torch.randn((M,C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C,C))*C**-0.5, torch.randn(8*C), dev)
    out.append(f'{timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))*1e3:.1f}')
print(os.environ.get('FD_LIB_PATH','default')[-12:], ' '.join(out))
