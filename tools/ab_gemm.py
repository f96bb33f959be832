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
tag = os.environ.get('FD_GEMM_PERSIST','1')
out=[]
for (M,N,K) in [(65536,320,320),(65536,320,1280),(16384,640,640),(4096,1280,1280),(65536,320,640),(16384,640,2560)]:
    a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
    out.append(f'{timeit(lambda: ops.gemm(a, w))*1e3:.1f}')
for (M,C) in [(65536,320),(16384,640),(4096,1280)]:
    a = torch.randn((M,C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C,C))*C**-0.5, torch.randn(8*C), dev)
    out.append(f'{timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))*1e3:.1f}')
for (B,H,Cin,Cout) in [(16,64,320,320),(16,32,1280,640),(16,16,1280,1280)]:
    x = ops.Act(torch.randn((B*H*H,Cin), device=dev).half(), B,H,H)
    w = ops.prep_conv(torch.randn((Cout,Cin,3,3))*(9*Cin)**-0.5, torch.randn(Cout), dev)
    out.append(f'{timeit(lambda: ops.conv2d(x, w), n=10)*1e3:.1f}')
print('persist='+tag, ' '.join(out))
