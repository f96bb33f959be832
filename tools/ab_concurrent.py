'''Would running INDEPENDENT small kernels of the deep UNet levels on two streams pay?  The self-attention's q|k projection and
V^T projection (both read the LayerNorm-1 statistics) at the 32x32 / 16x16 / 8x8 levels: back to back on one stream vs forked onto
a second stream and joined (events), 200 repetitions each, us per pair.'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
def run(fn_a, fn_b, fork, n=200):
    main = torch.cuda.current_stream()
    def once():
        if fork:
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn_b()
                ev2 = torch.cuda.Event(); ev2.record(side)
            fn_a()
            main.wait_event(ev2)
        else:
            fn_a(); fn_b()
    for _ in range(10): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): once()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for (B, HW, C) in ((16, 1024, 640), (16, 256, 1280), (16, 64, 1280)):
    M = B * HW
    h = torch.randn((M, C), generator=g).half().to(dev)
    st = ops.ln_row_stats(h)
    qk = ops.prep_linear_ln(torch.randn((2 * C, C), generator=g) * C ** -0.5, None, torch.ones(C), torch.zeros(C), dev)
    v = ops.prep_linear_ln(torch.randn((C, C), generator=g) * C ** -0.5, None, torch.ones(C), torch.zeros(C), dev)
    ldv = (HW + 7) // 8 * 8
    fa = lambda: ops.gemm(h, qk, ln_stats=st)
    fb = lambda: ops.gemm_vt(h, v, B, HW, ldv, ln_stats=st)
    a = run(fa, lambda: None, False); b = run(lambda: None, fb, False)
    for rep in range(2):
        s = run(fa, fb, False); f = run(fa, fb, True)
        print(f'M {M:6d} C {C:5d}: q|k alone {a:6.1f} us, V^T alone {b:6.1f} us, back to back {s:6.1f} us, forked on two streams {f:6.1f} us', flush=True)
