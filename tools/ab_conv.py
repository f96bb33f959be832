'''Conv / large-K GEMM tile A/B (FORCE_TILE) on the UNet and VAE conv shapes.'''
import sys, os; sys.path.insert(0,'/root/repo')
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
tiles = [int(t) for t in (sys.argv[1].split(',') if len(sys.argv) > 1 else '0,13,15')]
def timeit(fn, n=60):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
shapes = [(16,64,320,320),(16,64,640,320),(16,64,960,320),(16,32,640,640),(16,32,1280,640),(16,32,1920,640),(16,16,1280,1280),(16,16,2560,1280),(16,8,2560,1280)]
lin = [(65536,320,1280),(16384,640,2560),(4096,1280,5120)]
res = {}
for t in tiles * 2:
    ops.FORCE_TILE = t
    ops.FORCE_SPLIT = (int(os.environ.get('AB_SPLIT', '1')) if t else 0)
    row = []
    for (B,H,Cin,Cout) in shapes:
        x = ops.Act(torch.randn((B*H*H,Cin), device=dev).half(), B,H,H)
        w = ops.prep_conv(torch.randn((Cout,Cin,3,3))*(9*Cin)**-0.5, torch.randn(Cout), dev)
        ms = timeit(lambda: ops.conv2d(x, w))
        row.append(f'{ms*1e3:.0f}us/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    for (M,N,K) in lin:
        a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
        ms = timeit(lambda: ops.gemm(a, w))
        row.append(f'{ms*1e3:.0f}us/{2*M*N*K/ms/1e9:.0f}')
    print('tile', t, ' '.join(row))
