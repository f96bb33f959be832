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
for t in (13, 15, 14, 16, 13, 15):
    ops.FORCE_TILE = t; ops.FORCE_SPLIT = 1
    out=[]
    for (M,N,K) in [(8192,8000,8192),(16384,1280,8192),(65536,320,2880),(65536,640,5760)]:
        a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
        ms = timeit(lambda: ops.gemm(a, w)); out.append(f'{ms*1e3:.0f}us/{2*M*N*K/ms/1e9:.0f}')
    for (B,H,Cin,Cout) in [(16,64,320,320),(16,32,1280,640)]:
        x = ops.Act(torch.randn((B*H*H,Cin), device=dev).half(), B,H,H)
        w = ops.prep_conv(torch.randn((Cout,Cin,3,3))*(9*Cin)**-0.5, torch.randn(Cout), dev)
        ms = timeit(lambda: ops.conv2d(x, w)); out.append(f'{ms*1e3:.0f}us/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    print('tile', t, ' '.join(out))
