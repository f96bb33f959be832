'''GEGLU at the 16x16 level (M 4096, N 10240, K 1280): 640 tiles of 256x256 on 256 persistent slots = 2.5 rounds, the last
one half empty.  Does it pay to run the first 32 n-tile columns (512 tiles = 2 exact rounds) as they are and the remaining
8 as 256 tiles of 256x128 (one more exact round of half-size tiles) in a second launch?
    python tools/ab_tail_split.py [M = 4096] [C = 1280]'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
g = torch.Generator().manual_seed(0)
a = torch.randn((M, C), generator=g).half().to(dev)
st = ops.ln_row_stats(a)
w = ops.prep_linear_ln(torch.randn((8 * C, C), generator=g) * C ** -0.5, torch.randn(8 * C, generator=g), torch.ones(C), torch.zeros(C), dev, geglu=True)
out = torch.empty((M, 4 * C), dtype=torch.float16, device=dev)


def part(n0, n1):
    return ops.LinW(w.w[n0:n1], w.bias[n0:n1], n1 - n0, w.K, w.colsum[n0:n1])


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def whole():
    ops.gemm(a, w, act=ops.ACT_GEGLU, ln_stats=st, out=out)


tiles_m = M // 256
for n_main in (None, 256 // tiles_m * 256 * (8 * C // 256 * tiles_m // 256), ):
    pass
nt = 8 * C // 256                                   # n-tiles of 256 pre-activation columns
main_nt = (nt * tiles_m // 256) * 256 // tiles_m    # whole rounds' worth of n-tile columns
N1 = main_nt * 256
w1, w2 = part(0, N1), part(N1, 8 * C)


def split(tile2):
    ops.gemm(a, w1, act=ops.ACT_GEGLU, ln_stats=st, out=out[:, :N1 // 2])
    ops.FORCE_TILE = tile2
    try:
        ops.gemm(a, w2, act=ops.ACT_GEGLU, ln_stats=st, out=out[:, N1 // 2:])
    finally:
        ops.FORCE_TILE = 0


whole(); ref = out.clone()
split(14); assert torch.equal(ref[:, :N1 // 2], out[:, :N1 // 2]); err = float((ref.float() - out.float()).abs().max())
print(f'M {M} C {C}: {nt * tiles_m} tiles; main {main_nt} n-tile columns ({main_nt * tiles_m} tiles), tail {nt - main_nt} ({(nt - main_nt) * tiles_m}); max |diff| of the tail {err:.3g}')
for rnd in range(3):
    print(f'  whole {timeit(whole):6.1f} us | split, tail on tile 14 {timeit(lambda: split(14)):6.1f} | tail on tile 10 {timeit(lambda: split(10)):6.1f} | tail auto {timeit(lambda: split(0)):6.1f}', flush=True)
