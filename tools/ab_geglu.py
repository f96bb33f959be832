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
for tile in (0, 1, 10, 6, 14, 3, 11):
    ops.FORCE_TILE = tile
    out=[]
    for (M,C) in [(65536,320),(16384,640),(4096,1280),(1024,1280)]:
        a = torch.randn((M,C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C,C))*C**-0.5, torch.randn(8*C), dev)
        out.append(f'{timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))*1e3:.1f}')
    # VAE-like convs / linear N=512
    for (M,N,K) in [(32768,512,512),(524288,256,512)]:
        a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
        out.append(f'{timeit(lambda: ops.gemm(a, w))*1e3:.1f}')
    print('tile', tile, ' '.join(out))
