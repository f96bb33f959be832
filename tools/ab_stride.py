'''Does the row stride of the A operand matter to a short-K GEMM?  A K-tile of the LDS-DMA loop reads 128 contiguous bytes of
each row; with lda = 320 the other four 128-byte pieces of that row are fetched one K-tile (~1 us) later, each its own DRAM
access.  Times fd_gemm_f16 at M = 65536, N = 320 for (a) K = 64 columns of a contiguous [M][64] matrix, (b) the first 64 columns
of [M][320] / [M][640] / [M][1280] matrices (same bytes read, rows 640 / 1280 / 2560 bytes apart), (c) the full K = 320, and
(d) K = 320 read from five [M][64] panels through the appended-operand path is not expressible -- see the printed ratios.
Every timing loop rotates over enough distinct buffers (> 512 MB) that no operand is served by the Infinity Cache.'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
M, N = 65536, 320


def timeit(fns, n=40):
    for f in fns[:3]:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(n):
        fns[i % len(fns)]()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


g = torch.Generator().manual_seed(1)
w64 = ops.prep_linear(torch.randn((N, 64), generator=g) * 0.1, None, dev)
w320 = ops.prep_linear(torch.randn((N, 320), generator=g) * 0.05, None, dev)
NB = 14
outs = [torch.empty((M, N), dtype=torch.float16, device=dev) for _ in range(NB)]
for ld in (64, 320, 640, 1280):
    bufs = [torch.randn((M, ld), device=dev).half() for _ in range(NB)]
    us = timeit([lambda b=b, o=o: ops.gemm(b[:, :64], w64, out=o) for b, o in zip(bufs, outs)])
    rd = M * 64 * 2 / 1e6
    print(f'K = 64 of rows {ld * 2:5d} bytes apart: {us:6.1f} us  ({rd:.1f} MB read + {M * N * 2 / 1e6:.1f} MB written -> {(rd + M * N * 2 / 1e6) / us / 1e3 * 1e3:.0f} GB/s)', flush=True)
    if ld == 320:
        us = timeit([lambda b=b, o=o: ops.gemm(b, w320, out=o) for b, o in zip(bufs, outs)])
        print(f'K = 320, lda 320: {us:6.1f} us  ({(M * 320 * 2 + M * N * 2) / 1e6 / us * 1e3 / 1e3:.2f} TB/s)', flush=True)
    del bufs
# the same bytes as K = 320 / lda 320, but as ONE K = 320 GEMM whose A rows are contiguous 640-byte runs read by a streaming kernel
x = [torch.randn((M, 320), device=dev).half() for _ in range(NB)]
from flexdiffuse_amd.ops import Act
gn_w = torch.ones(320, device=dev); gn_b = torch.zeros(320, device=dev)
us = timeit([lambda b=b: ops.ln_row_stats(b) for b in x])
print(f'read-only pass over [M][320] (fd_ln_row_stats_f16): {us:6.1f} us  ({M * 640 / 1e6 / us * 1e3 / 1e3:.2f} TB/s)')
us = timeit([lambda b=b, o=o: ops.copy_rows(o, b) for b, o in zip(x, outs)]) if hasattr(ops, 'copy_rows') else 0
print(f'copy [M][320] -> [M][320] (fd_copy2d_f16): {us:6.1f} us  ({2 * M * 640 / 1e6 / max(us, 1e-9) * 1e3 / 1e3:.2f} TB/s)')
