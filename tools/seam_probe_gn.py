'''Round 6, VERDICT r5 next 1: the deep-level chain  [split-K partial pass -> finish -> GroupNorm + SiLU]  with the GroupNorm inside the finish
pass (fd_gemm_desc.gn_out, k_splitk_finish_gn: the product) against the three launches it replaces (FD knob ops.GN_FINISH_FUSE = False), same
process, interleaved samples, same method as tools/seam_probe.py: start-to-start time of the chain issued 50x back to back, best of 3, per
arm; four interleaved rounds.  Shapes: a ResBlock's conv1 (per-sample time-embedding bias, output not kept) at the 8x8 and 16x16 levels of the
bench forward (CFG batch 16), and conv2 with the appended shortcut (output kept: the conv2 -> next block's norm form).
    python tools/seam_probe_gn.py'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
cases = []
for (B, H, C, Cx) in ((16, 8, 1280, 0), (16, 16, 1280, 0), (16, 8, 1280, 2560), (16, 16, 1280, 2560)):
    M = B * H * H
    x = ops.Act((torch.randn((M, C), generator=g) * 0.7).half().to(dev), B, H, H)
    w = torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5
    b = torch.randn(C, generator=g)
    spec = ops.GNSpec(torch.ones(C, device=dev), torch.zeros(C, device=dev), 32, 1e-5, True)
    if Cx:
        cw = ops.prep_conv_shortcut(w, b, torch.randn((C, Cx), generator=g) * Cx ** -0.5, None, dev)
        kw = dict(a2=(torch.randn((M, Cx), generator=g) * 0.7).half().to(dev), keep=True)
    else:
        cw = ops.prep_conv(w, b, dev)
        kw = dict(bias2=torch.randn((B, C), generator=g).to(dev), ld_bias2=C, keep=False)
    cases.append((f'{H}x{H} {"conv2 + shortcut, kept" if Cx else "conv1, not kept"}', x, cw, spec, kw))


def chain_time(x, cw, spec, kw):
    fn = lambda: ops.conv2d(x, cw, gn=spec, **kw)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    return best


for name, x, cw, spec, kw in cases:
    rows = {True: [], False: []}
    for rnd in range(4):
        for fuse in (True, False):
            ops.GN_FINISH_FUSE = fuse
            rows[fuse].append(chain_time(x, cw, spec, kw))
            assert ops._last_conv_gn_fused == fuse
    ops.GN_FINISH_FUSE = True
    f, u = sorted(rows[True]), sorted(rows[False])
    print(f'{name}: fused {" ".join(f"{t:.1f}" for t in rows[True])} us | three launches {" ".join(f"{t:.1f}" for t in rows[False])} us '
          f'-> median {0.5 * (f[1] + f[2]):.1f} vs {0.5 * (u[1] + u[2]):.1f} us ({0.5 * (u[1] + u[2]) - 0.5 * (f[1] + f[2]):+.1f} us per chain)', flush=True)
