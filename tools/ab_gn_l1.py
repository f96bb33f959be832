'''Level-1 GroupNorm (16 x 32x32 x 640, and the decoder's concatenated inputs): the one-launch slab kernel against the apply pass alone fed with
producer-side partial sums (what extending fd_gemm_desc.gn_part_out to the 32x32 level would buy per launch).  python tools/ab_gn_l1.py'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (B, H, C) in ((16, 32, 640), (16, 32, 1280), (16, 64, 320), (16, 16, 1280)):
    x = ops.Act((torch.randn((B * H * H, C), generator=g)).half().to(dev), B, H, H)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    chunks = max(1, H * H // 128)
    xs = x.t.float().view(B, chunks, H * H // chunks, 32, C // 32)
    parts = ops.GNParts(torch.stack([xs.sum(dim=(2, 4)), (xs * xs).sum(dim=(2, 4))], -1).contiguous(), chunks, 32)
    a = t(lambda: ops.groupnorm(x, gamma, beta, 32, 1e-5, True))
    b = t(lambda: ops.groupnorm(x, gamma, beta, 32, 1e-5, True, parts=parts))
    y1, y2 = ops.groupnorm(x, gamma, beta, 32, 1e-5, True), ops.groupnorm(x, gamma, beta, 32, 1e-5, True, parts=parts)
    print(f'{B} x {H}x{H} x {C}: full GroupNorm {a:.1f} us, apply pass from partial sums {b:.1f} us ({a - b:+.1f} us); max |diff| {float((y1.t.float() - y2.t.float()).abs().max()):.3g}', flush=True)
