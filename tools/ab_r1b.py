'''A/B of the 64-row wave tiles (15: 256x256, 16: 256x320; 16 waves as 4x4) against the
current choices on GEGLU, conv3x3 and FF-out shapes, with a correctness check per tile.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def run(tile, split, fn):
    ops.FORCE_TILE, ops.FORCE_SPLIT = tile, split
    try:
        return timeit(fn)
    except Exception as e:
        return float('nan')
    finally:
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
# correctness of tiles 15/16 on a conv and a GEGLU
torch.manual_seed(0)
x = torch.randn(2, 320, 32, 32); w = torch.randn(640, 320, 3, 3) * (9 * 320) ** -0.5; b = torch.randn(640)
want = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, 640)
xa = ops.nchw_to_nhwc(x.to(dev)); wp = ops.prep_conv(w, b, dev)
for t in (13, 15, 16):
    ops.FORCE_TILE, ops.FORCE_SPLIT = t, 1
    y = ops.conv2d(xa, wp).t.float().cpu()
    print('conv tile', t, 'max err', float((y - want).abs().max()))
ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
a = torch.randn(2048, 320); wg = torch.randn(2560, 320) * 320 ** -0.5; bg = torch.randn(2560)
h = a.half().float() @ wg.half().float().T + bg
wantg = h[:, :1280] * F.gelu(h[:, 1280:])
wgp = ops.prep_geglu(wg, bg, dev)
for t in (0, 15, 10):
    ops.FORCE_TILE, ops.FORCE_SPLIT = t, (1 if t else 0)
    y = ops.gemm(a.half().to(dev), wgp, act=ops.ACT_GEGLU).float().cpu()
    print('geglu tile', t, 'max err', float((y - wantg).abs().max()))
ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
print('--- GEGLU us (tile: level0 level1 level2 mid)')
for t in (0, 14, 15):
    row = []
    for (M, C) in [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280)]:
        a = torch.randn((M, C), device=dev).half(); w = ops.prep_geglu(torch.randn((8 * C, C)) * C ** -0.5, torch.randn(8 * C), dev)
        ms = run(t, 1 if t else 0, lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))
        row.append(f'{ms*1e3:.1f}us/{2*M*8*C*C/ms/1e9:.0f}TF')
    print('tile', t, ' '.join(row))
print('--- conv3x3 us')
shapes = [(16,64,320,320),(16,64,640,320),(16,64,960,320),(16,64,640,640),(16,32,640,640),(16,32,1280,640),(16,32,1920,640),(16,32,1280,1280),(16,16,1280,1280),(16,16,2560,1280)]
for (t, sp) in [(0, 0), (13, 1), (16, 1), (16, 2), (15, 1), (15, 2)]:
    row = []
    for (B, H, Cin, Cout) in shapes:
        x = ops.Act(torch.randn((B * H * H, Cin), device=dev).half(), B, H, H)
        w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3)) * (9 * Cin) ** -0.5, torch.randn(Cout), dev)
        ms = run(t, sp, lambda: ops.conv2d(x, w))
        row.append(f'{ms*1e3:.0f}/{2*B*H*H*Cout*9*Cin/ms/1e9:.0f}')
    print('tile', t, 'split', sp, ' '.join(row))
print('--- linear (FF out K=4C, proj K=C) us')
lin = [(65536,320,1280),(16384,640,2560),(4096,1280,5120),(65536,320,320),(16384,640,640)]
for t in (0, 13, 16, 15):
    row = []
    for (M, N, K) in lin:
        a = torch.randn((M, K), device=dev).half(); w = ops.prep_linear(torch.randn((N, K)) * K ** -0.5, torch.randn(N), dev)
        ms = run(t, 1 if t else 0, lambda: ops.gemm(a, w))
        row.append(f'{ms*1e3:.0f}/{2*M*N*K/ms/1e9:.0f}')
    print('tile', t, ' '.join(row))
