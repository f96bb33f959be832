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
for (B,HW,C) in [(16,4096,320),(16,1024,640),(16,256,1280),(16,64,1280)]:
    a = torch.randn((B*HW,C), device=dev).half(); w = ops.prep_linear(torch.randn((C,C))*C**-0.5, None, dev)
    out.append(f'{timeit(lambda: ops.gemm_vt(a, w, B, HW, (HW+7)//8*8))*1e3:.1f}')
print(os.environ.get('FD_LIB_PATH','default')[-12:], ' '.join(out))
