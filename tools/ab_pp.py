'''Ping-pong GEMM tiles (csrc/gemm_pp.hip, tile ids 30..33 / 40..43) against the library's rule on the UNet's big shapes:
3x3 convolutions of every level (CFG batch 16), the FF-out GEMM with its folded proj_out phase + residual, the GEGLU
projection with the LayerNorm fold.  usage: ab_pp.py [tile:split,...]   (0:0 = the rule)'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
arms = [tuple(int(v) for v in a.split(':')) for a in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0:0', '30:1', '32:1', '33:1'])]
which = sys.argv[2] if len(sys.argv) > 2 else 'conv,lin,geglu'


def timeit(fn, n=40):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


convs = [(16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 1920, 640),
         (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280)]
lins = [(65536, 320, 1280, 320), (16384, 640, 2560, 640), (4096, 1280, 5120, 1280)]      # FF-out: M, N, K, K2 (proj_out folded) + residual
geglus = [(65536, 320), (16384, 640), (4096, 1280)]
rows = {}
for rnd_ in range(2):
    for (t, sp) in arms:
        ops.FORCE_TILE, ops.FORCE_SPLIT = t, sp
        row = []
        if 'conv' in which:
            for (B, H, Cin, Cout) in convs:
                x = ops.Act(torch.randn((B * H * H, Cin), device=dev).half(), B, H, H)
                w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5, torch.randn(Cout), dev)
                try:
                    ms = timeit(lambda: ops.conv2d(x, w))
                    row.append(f'{ms * 1e3:.0f}/{2 * B * H * H * Cout * 9 * Cin / ms / 1e9:.0f}')
                except ValueError:
                    row.append('-')
        if 'ck2' in which:     # conv2 of a ResBlock with a channel change: 3x3 over Cout channels + the 1x1 shortcut over Cx appended
            for (B, H, C, Cx) in [(16, 64, 320, 640), (16, 64, 320, 960), (16, 32, 640, 1280), (16, 32, 640, 1920), (16, 32, 640, 320), (16, 16, 1280, 2560), (16, 16, 1280, 640)]:
                x = ops.Act(torch.randn((B * H * H, C), device=dev).half(), B, H, H)
                xs = torch.randn((B * H * H, Cx), device=dev).half()
                w = ops.prep_conv_shortcut(torch.randn((C, C, 3, 3)) * (9 * C) ** -0.5, torch.randn(C), torch.randn((C, Cx, 1, 1)) * Cx ** -0.5, None, dev)
                try:
                    ms = timeit(lambda: ops.conv2d(x, w, a2=xs))
                    row.append(f'{ms * 1e3:.0f}/{2 * B * H * H * C * (9 * C + Cx) / ms / 1e9:.0f}')
                except ValueError:
                    row.append('-')
        if 'lin' in which:
            for (M, N, K, K2) in lins:
                a = torch.randn((M, K), device=dev).half(); a2 = torch.randn((M, K2), device=dev).half()
                res = torch.randn((M, N), device=dev).half()
                w = ops.prep_linear(torch.randn((N, K + K2)) * K ** -0.5, torch.randn(N), dev)
                try:
                    ms = timeit(lambda: ops.gemm(a, w, a2=a2, residual=res))
                    row.append(f'{ms * 1e3:.0f}/{2 * M * N * (K + K2) / ms / 1e9:.0f}')
                except ValueError:
                    row.append('-')
        if 'geglu' in which:
            for (M, C) in geglus:
                a = torch.randn((M, C), device=dev).half()
                st = ops.ln_row_stats(a)
                w = ops.prep_linear_ln(torch.randn((8 * C, C)) * C ** -0.5, torch.randn(8 * C), torch.ones(C), torch.zeros(C), dev, geglu=True)
                try:
                    ms = timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU, ln_stats=st))
                    row.append(f'{ms * 1e3:.0f}/{2 * M * 8 * C * C / ms / 1e9:.0f}')
                except ValueError:
                    row.append('-')
        print(f'tile {t:2d} split {sp}  ' + '  '.join(f'{c:>9s}' for c in row), flush=True)
