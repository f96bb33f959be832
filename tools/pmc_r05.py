'''Round-5 PMC evidence set (run under rocprofv3 by tools/pmc_r05.sh): the ping-pong GEMM kernels (csrc/gemm_pp.hip) beside the 2-barrier
kernels of gemm.hip on the level-0 convolution 16x64x64x320->320, the level-0 FF-out GEMM (+ folded proj_out + residual) and the GEGLU
projection with the LayerNorm fold.  argv[1]: comma list of tile:split arms (default 0:0,30:1).'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
arms = [tuple(int(v) for v in a.split(':')) for a in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0:0', '30:1'])]
g = torch.Generator().manual_seed(0)
B, H, Cin, Cout = 16, 64, 320, 320
x = ops.Act((torch.randn((B * H * H, Cin), generator=g) * 0.7).half().to(dev), B, H, H)
w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5, torch.randn(Cout, generator=g), dev)
M, N, K, K2 = 65536, 320, 1280, 320
a = torch.randn((M, K), generator=g).half().to(dev); a2 = torch.randn((M, K2), generator=g).half().to(dev)
res = torch.randn((M, N), generator=g).half().to(dev)
lw = ops.prep_linear(torch.randn((N, K + K2), generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
for tile, split in arms:
    ops.FORCE_TILE, ops.FORCE_SPLIT = tile, split
    for _ in range(3):
        ops.conv2d(x, w)
    for _ in range(3):
        ops.gemm(a, lw, a2=a2, residual=res)
ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
if len(sys.argv) > 2 and sys.argv[2] == 'all':     # the other big kernels of the forward, through the rule
    st = ops.ln_row_stats(a[:, :320].contiguous())
    gw = ops.prep_linear_ln(torch.randn((2560, 320), generator=g) * 320 ** -0.5, torch.randn(2560, generator=g), torch.ones(320), torch.zeros(320), dev, geglu=True)
    a320 = a[:, :320].contiguous()
    for _ in range(3):
        ops.gemm(a320, gw, act=ops.ACT_GEGLU, ln_stats=st)
    for (Bc, Hc, Cc) in ((16, 16, 1280), (16, 8, 1280)):
        xc = ops.Act((torch.randn((Bc * Hc * Hc, Cc), generator=g) * 0.7).half().to(dev), Bc, Hc, Hc)
        wc = ops.prep_conv(torch.randn((Cc, Cc, 3, 3), generator=g) * (9 * Cc) ** -0.5, torch.randn(Cc, generator=g), dev)
        for _ in range(3):
            ops.conv2d(xc, wc)
torch.cuda.synchronize()
