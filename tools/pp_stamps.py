'''Where a ping-pong GEMM workgroup spends its time: load a -DFD_PP_STAMPS build of the library (FD_LIB_PATH), run one problem on a
ping-pong tile and print, per workgroup sample, prologue / main loop / epilogue-issue cycles (s_memtime) and the section stamps of the
last full K-tile (LOAD start, LOAD end, MFMA-issue end per phase).  Build the variant with:
  make -C flexdiffuse_amd/csrc clean; make -C flexdiffuse_amd/csrc -j8 CXXFLAGS='-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -ffp-contract=on -DFD_PP_STAMPS' LIB=../../tools/_variants/libfd_stamps.so
usage: FD_LIB_PATH=tools/_variants/libfd_stamps.so python tools/pp_stamps.py conv|lin|geglu tile'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
what, tile = sys.argv[1], int(sys.argv[2])
g = torch.Generator().manual_seed(0)
if what == 'conv':
    B, H, Cin, Cout = 16, 64, 320, 320
    x = ops.Act((torch.randn((B * H * H, Cin), generator=g) * 0.7).half().to(dev), B, H, H)
    w = ops.prep_conv(torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5, torch.randn(Cout, generator=g), dev)
    run = lambda: ops.conv2d(x, w)
elif what == 'lin':
    M, N, K, K2 = 65536, 320, 1280, 320
    a = torch.randn((M, K), generator=g).half().to(dev); a2 = torch.randn((M, K2), generator=g).half().to(dev)
    res = torch.randn((M, N), generator=g).half().to(dev)
    lw = ops.prep_linear(torch.randn((N, K + K2), generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
    run = lambda: ops.gemm(a, lw, a2=a2, residual=res)
else:
    M, C = 65536, 320
    a = torch.randn((M, C), generator=g).half().to(dev)
    st = ops.ln_row_stats(a)
    lw = ops.prep_linear_ln(torch.randn((8 * C, C), generator=g) * C ** -0.5, torch.randn(8 * C, generator=g), torch.ones(C), torch.zeros(C), dev, geglu=True)
    run = lambda: ops.gemm(a, lw, act=ops.ACT_GEGLU, ln_stats=st)
ops.FORCE_TILE, ops.FORCE_SPLIT = tile, 1
for _ in range(4):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
ws = ops._splitk_ws[(0, 0)].view(torch.int64)[:256 * 8 * 16].cpu().view(256, 8, 16)
span = (ws[:, :, 3].max(dim=1).values - ws[:, :, 0].min(dim=1).values).float()   # per workgroup (each XCD has its own counter)
print(f'{what} tile {tile}: {us:.1f} us per launch (back to back, this build); workgroup first stamp -> last stamp {span.mean():.0f} cycles '
      f'(min {span.min():.0f} max {span.max():.0f}) = {span.mean() / us / 1e3:.2f} cycles per ns if the launch were only that')
rt = ws[:, :, 15].float()   # s_memrealtime ticks (100 MHz) from the wave's first to its last instruction
print(f'  wave lifetime by the 100 MHz counter: {rt.mean() / 100:.1f} us (min {rt.min() / 100:.1f} max {rt.max() / 100:.1f}) -> s_memtime runs at '
      f'{((ws[:, :, 3] - ws[:, :, 0]).float() / (rt / 100) / 1e3).mean():.3f} ticks per ns')
pro, loop, epi = (ws[:, :, 1] - ws[:, :, 0]).float(), (ws[:, :, 2] - ws[:, :, 1]).float(), (ws[:, :, 3] - ws[:, :, 2]).float()
print(f'  prologue {pro.mean():.0f} (min {pro.min():.0f} max {pro.max():.0f})   main loop {loop.mean():.0f} (min {loop.min():.0f} max {loop.max():.0f})   '
      f'epilogue issue {epi.mean():.0f} (min {epi.min():.0f} max {epi.max():.0f})')
for blk in (0, 131):
    base = int(ws[blk, 0, 4])
    print(f'  workgroup {blk}: per phase [LOAD start, LOAD end, MFMA issue end] relative to wave 0')
    for wv in range(8):
        print('    wave %d: ' % wv + '  '.join('[%5d %5d %5d]' % tuple(int(ws[blk, wv, 4 + 3 * p + i]) - base for i in range(3)) for p in range(3)) + '  [%5d %5d]' % (int(ws[blk, wv, 13]) - base, int(ws[blk, wv, 14]) - base))
