'''The 8x8 -> 16x16 upsample convolution of the bench forward (16 x 8 x 8 x 1280 -> 16 x 16 x 16 x 1280): fused nearest upsample (3x3 on the
upsampled map, what the UNet uses below FD_UP_PHASES_MIN_ROWS = 4096 low-resolution rows) against the parity decomposition (four 2x2
convolutions of the low-resolution map, 4/9 of the MACs) per forced tile.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, H, C = 16, (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 1280
x = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.7).half().to(dev), B, H, H)
w = torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5
b = torch.randn(C, generator=g)
cw, cp = ops.prep_conv(w, b, dev), ops.prep_conv_up_phases(w, b, dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ref = ops.conv2d(x, cw, up=True).t.float()
for rnd in range(2):
    row = [f'fused up (rule): {timeit(lambda: ops.conv2d(x, cw, up=True)):.1f}']
    for tile in (0, 9, 12, 13, 20, 32, 33):
        ops.FORCE_TILE = tile
        try:
            y = ops.conv2d_up_phases(x, cp).t.float()
            err = float((y - ref).abs().max())
            row.append(f'phases tile {tile}: {timeit(lambda: ops.conv2d_up_phases(x, cp)):.1f} (max|d| {err:.3f})')
        except Exception as e:
            row.append(f'phases tile {tile}: refused')
        ops.FORCE_TILE = 0
    print(' | '.join(row), flush=True)
