'''Round-4 PMC evidence set (run under rocprofv3 by tools/pmc_r04.sh): the two families VERDICT r3 asked counters for.
  * the 8x8-level convolution M 1024 N 1280 K 11520 as 256x160 tiles x split-K 8 (the rule's choice), 128x160 3-stage tiles x split-K 4
    (BM = 128: half the fp32 slab round trip) and 128x160 2-stage / 8-wave tiles x split-K 4 (two co-resident workgroups per CU);
  * the short-K residual linear 65536x320x320: the 256x320 tile (one workgroup per CU) against the 128x160 / 8-wave tile (two co-resident
    workgroups per CU).'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x3 = ops.Act((torch.randn((16 * 8 * 8, 1280), generator=g) * 0.7).half().to(dev), 16, 8, 8)
w3 = ops.prep_conv(torch.randn((1280, 1280, 3, 3), generator=g) * (9 * 1280) ** -0.5, torch.randn(1280, generator=g), dev)
for tile, split in ((0, 0), (20, 4), (9, 4)):
    ops.FORCE_TILE, ops.FORCE_SPLIT = tile, split
    for _ in range(3):
        ops.conv2d(x3, w3)
a = torch.randn((65536, 320), generator=g).half().to(dev)
res = torch.randn((65536, 320), generator=g).half().to(dev)
lw = ops.prep_linear(torch.randn((320, 320), generator=g) * 320 ** -0.5, torch.randn(320, generator=g), dev)
for tile in (16, 9):
    ops.FORCE_TILE, ops.FORCE_SPLIT = tile, 1
    for _ in range(3):
        ops.gemm(a, lw, residual=res)
ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
torch.cuda.synchronize()
