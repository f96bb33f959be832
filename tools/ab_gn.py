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
for (B,HW,C) in [(16,4096,320),(16,4096,640),(16,4096,960),(16,1024,640),(16,1024,1280),(16,1024,1920),(16,256,1280),(16,256,2560),(8,262144,128),(8,65536,256),(8,16384,512)]:
    x = ops.Act(torch.randn((B*HW,C),device=dev).half(),B,HW,1)
    g = torch.ones(C,device=dev); b=torch.zeros(C,device=dev)
    ms = timeit(lambda: ops.groupnorm(x,g,b,32,1e-5,True))
    out.append(f'{ms*1e3:.0f}us/{B*HW*C*6/ms/1e6:.0f}')
print(os.environ.get('FD_LIB_PATH','default')[-12:], ' '.join(out))
