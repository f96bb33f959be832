'''What the split-K seam costs: start-to-start time of the chain  [split-K partial pass -> finish launch -> a dependent
consumer kernel]  repeated back to back, for the 8x8-level convolution (M 1024 N 1280 K 11520: 256x160 tiles x split-K 8 on the 2-barrier
loop) and the 16x16-level one (M 4096: 256x320 tiles x split-K 4 on the ping-pong loop), with three builds of the library:
  default                              the product
  -DFD_SPLITK_NO_STORE                 the partial pass keeps its accumulators in registers (no fp32 slabs written: no dirty bytes behind it)
  -DFD_SPLITK_NO_STORE -DFD_SPLITK_NO_FINISH   and no finish launch either (one launch + one boundary fewer)
Timing only for the two variants (their outputs are garbage).  usage: FD_LIB_PATH=... python tools/seam_probe.py'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
out = []
for (B, H, C) in ((16, 8, 1280), (16, 16, 1280)):
    x = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.7).half().to(dev), B, H, H)
    w = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g), dev)
    gg, gb = torch.ones(C, device=dev), torch.zeros(C, device=dev)

    def chain():
        y = ops.conv2d(x, w)
        return ops.groupnorm(y, gg, gb, 32, 1e-5, True)     # the consumer a ResBlock really has behind its conv

    def consumer_only(y=ops.conv2d(x, w)):
        return ops.groupnorm(y, gg, gb, 32, 1e-5, True)
    res = []
    for fn in (chain, consumer_only):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        res.append(best)
    out.append(f'{H}x{H}: chain {res[0]:.1f} us, consumer alone {res[1]:.1f} us -> conv + seam {res[0] - res[1]:.1f} us')
print(os.path.basename(os.environ.get('FD_LIB_PATH', 'default')), ' | '.join(out))
