'''Linear / GEGLU GEMM timing per forced tile (argv[1] = comma list, 0 = auto).'''
import sys, os; sys.path.insert(0,'/root/repo')
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
tiles = [int(t) for t in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0'])]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
shapes = [(65536,320,320),(65536,640,320),(65536,320,1280),(16384,640,640),(16384,1280,640),(16384,640,2560),(4096,1280,1280),(4096,2560,1280),(4096,1280,5120)]
print('shapes', shapes, '+ geglu 320/640/1280')
for t in tiles:
    ops.FORCE_TILE = t; ops.FORCE_SPLIT = 1 if t else 0
    out=[]
    for (M,N,K) in shapes:
        a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
        r = torch.randn((M,N), device=dev).half()
        try: out.append(f'{timeit(lambda: ops.gemm(a, w, residual=r))*1e3:.1f}')
        except Exception as e: out.append('x')
    for (M,C) in [(65536,320),(16384,640),(4096,1280)]:
        a = torch.randn((M,C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C,C))*C**-0.5, torch.randn(8*C), dev)
        try: out.append(f'{timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))*1e3:.1f}')
        except Exception as e: out.append('x')
    print('tile', t, ' '.join(out))
