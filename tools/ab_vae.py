import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
row = []
for (B, H, Cin, Cout) in [(8,64,512,512),(8,128,512,512),(8,256,512,256),(8,256,256,256),(8,512,256,128),(8,512,128,128)]:
    x = ops.Act(torch.randn((B * H * H, Cin), device=dev).half(), B, H, H)
    w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5, torch.randn(Cout), dev)
    ms = timeit(lambda: ops.conv2d(x, w))
    row.append(f'{ms*1e3:.0f}/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    del x
print('VAE15', os.environ.get('FD_GEMM_VAE15', '1'), ' '.join(row), flush=True)
