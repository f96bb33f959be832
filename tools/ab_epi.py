'''Lean vs generic GEMM epilogue (FD_GEMM_FAST_EPI=1/0, read at library load: run once per value).
Linear projections with / without residual, GEGLU, conv3x3 with time-embedding bias + residual.'''
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
    return e0.elapsed_time(e1) / n * 1e3
tag = 'fast_epi=' + os.environ.get('FD_GEMM_FAST_EPI', '1')
row = []
for (M, N, K) in [(65536,320,320),(65536,640,320),(65536,320,1280),(16384,640,640),(16384,1280,640),(16384,640,2560),(4096,1280,1280),(4096,1280,5120)]:
    a = torch.randn((M, K), device=dev).half(); w = ops.prep_linear(torch.randn((N, K)) * K ** -0.5, torch.randn(N), dev)
    r = torch.randn((M, N), device=dev).half()
    row.append(f'({M},{N},{K}) {timeit(lambda: ops.gemm(a, w)):.1f} +res {timeit(lambda: ops.gemm(a, w, residual=r)):.1f}')
print(tag, 'linear us:', ' | '.join(row), flush=True)
row = []
for (M, C) in [(65536,320),(16384,640),(4096,1280)]:
    a = torch.randn((M, C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C, C)) * C ** -0.5, torch.randn(8*C), dev)
    row.append(f'({M},{C}) {timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU), 20):.1f}')
print(tag, 'geglu us:', ' | '.join(row), flush=True)
row = []
for (B, H, Cin, Cout) in [(16,64,320,320),(16,64,640,320),(16,32,640,640),(16,32,1280,640),(16,16,1280,1280),(8,128,512,512),(8,256,256,256)]:
    x = ops.Act(torch.randn((B*H*H, Cin), device=dev).half(), B, H, H)
    w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9*Cin) ** -0.5, torch.randn(Cout), dev)
    b2 = torch.randn((B, Cout), device=dev)
    r = torch.randn((B*H*H, Cout), device=dev).half()
    t0 = timeit(lambda: ops.conv2d(x, w), 10)
    t1 = timeit(lambda: ops.conv2d(x, w, bias2=b2, ld_bias2=Cout), 10)
    t2 = timeit(lambda: ops.conv2d(x, w, residual=r), 10)
    row.append(f'({B},{H},{Cin},{Cout}) {t0:.0f} b2 {t1:.0f} res {t2:.0f}')
print(tag, 'conv us:', ' | '.join(row), flush=True)
