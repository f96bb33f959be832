import sys, os; sys.path.insert(0,'/root/repo')
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
out=[]
for (B,N,heads,d,Nk) in [(16,4096,8,40,4096),(16,1024,8,80,1024),(16,256,8,160,256),(16,4096,8,40,77)]:
    C=heads*d
    q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*Nk,C),device=dev).half()
    vt = torch.randn((B,C,(Nk+7)//8*8),device=dev).half()
    out.append(f'{timeit(lambda: ops.attention(q,k,vt,B,heads,N,Nk,d))*1e3:.1f}')
print(os.environ.get('FD_LIB_PATH','default')[-12:], ' '.join(out))
