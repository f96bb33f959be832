'''Round-6 PMC evidence set (run under rocprofv3 by tools/pmc_traffic_r06.sh): the kernels that dominate the forward at the round-6 head, launched
through the rule exactly as the UNet launches them -- the level-0 convolution 16x64x64x320->320 plain and with the GroupNorm partial sums in its
epilogue (conv1 of a ResBlock), the level-0 convolution with the appended 1x1 shortcut (the 2-barrier 256x320 tile: VERDICT r5 next 7 asked for its
counters), the FF-out GEMM, the level-0 GEGLU, and the 16x16 / 8x8-level convolutions with the GroupNorm inside their split-K finish pass.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, H, C = 16, 64, 320
M = B * H * H
x = ops.Act((torch.randn((M, C), generator=g) * 0.7).half().to(dev), B, H, H)
w = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g), dev)
temb = torch.randn((B, C), generator=g).to(dev)
for _ in range(3):
    ops.conv2d(x, w)                                              # plain (EPI 1)
for _ in range(3):
    ops.conv2d(x, w, bias2=temb, ld_bias2=C, gn_parts=32)         # conv1 of a ResBlock: + per-sample bias + GroupNorm partial sums (EPI 11)
Cx = 640
wk2 = ops.prep_conv_shortcut(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g),
                             torch.randn((C, Cx), generator=g) * Cx ** -0.5, None, dev)
xk2 = (torch.randn((M, Cx), generator=g) * 0.7).half().to(dev)
for _ in range(3):
    ops.conv2d(x, wk2, a2=xk2, gn_parts=32)                       # conv2 with the appended shortcut (2-barrier 256x320 tile, EPI 11)
K, K2 = 1280, 320
a = torch.randn((M, K), generator=g).half().to(dev); a2 = torch.randn((M, K2), generator=g).half().to(dev)
res = torch.randn((M, C), generator=g).half().to(dev)
lw = ops.prep_linear(torch.randn((C, K + K2), generator=g) * K ** -0.5, torch.randn(C, generator=g), dev)
for _ in range(3):
    ops.gemm(a, lw, a2=a2, residual=res)
a320 = a[:, :320].contiguous()
st = ops.ln_row_stats(a320)
gw = ops.prep_linear_ln(torch.randn((2560, 320), generator=g) * 320 ** -0.5, torch.randn(2560, generator=g), torch.ones(320), torch.zeros(320), dev, geglu=True)
for _ in range(3):
    ops.gemm(a320, gw, act=ops.ACT_GEGLU, ln_stats=st)
for (Bc, Hc, Cc) in ((16, 16, 1280), (16, 8, 1280)):
    xc = ops.Act((torch.randn((Bc * Hc * Hc, Cc), generator=g) * 0.7).half().to(dev), Bc, Hc, Hc)
    wc = ops.prep_conv(torch.randn((Cc, Cc, 3, 3), generator=g) * (9 * Cc) ** -0.5, torch.randn(Cc, generator=g), dev)
    spec = ops.GNSpec(torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), 32, 1e-5, True)
    tb = torch.randn((Bc, Cc), generator=g).to(dev)
    for _ in range(3):
        ops.conv2d(xc, wc, bias2=tb, ld_bias2=Cc, gn=spec, keep=False)
    assert ops._last_conv_gn_fused
torch.cuda.synchronize()
