'''Same-box A/B: fd_xattn_q_f16 (q projection + cross-attention fused) vs the two launches it replaces,
at the level-0 shape of the headline config (CFG batch 16 x 4096 rows x 320, 77 keys) and at the shared
CFG prefix shape (8 samples, 2 context replicas).  Prints us per call (events on the launch stream).'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
heads, L = 8, 77


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for B, HW, rep, d in ((16, 4096, 1, 40), (8, 4096, 2, 40), (4, 9216, 1, 40), (16, 1024, 1, 80), (4, 2304, 1, 80)):
    C = heads * d
    M = B * HW
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((M, C), generator=g) * 1.3 + torch.randn((M, 1), generator=g)).half().to(dev)
    lw = ops.prep_linear_ln(torch.randn((C, C), generator=g) * C ** -0.5 * ops.QK_LOG2E * d ** -0.5, None,
                            1.0 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g), dev)
    kd = (torch.randn((rep * B * L, C), generator=g) * 1.2).half().to(dev)
    vtd = torch.zeros((rep * B, C, 80), dtype=torch.float16); vtd[:, :, :L] = torch.randn((rep * B, C, L), generator=g).half()
    vtd = vtd.to(dev)
    st = ops.ln_row_stats(x)
    img = ops.xattn_pack_kv(kd, vtd, rep * B, L, heads, d)
    out = torch.empty((rep * M, C), dtype=torch.float16, device=dev)
    q = torch.empty((M, C), dtype=torch.float16, device=dev)

    def unfused():
        ops.gemm(x, lw, ln_stats=st, out=q)
        for r in range(rep):
            ops.attention(q, kd[r * B * L:(r + 1) * B * L], vtd[r * B:(r + 1) * B], B, heads, HW, L, d,
                          q_prescaled=True, out=out[r * M:(r + 1) * M])
    t_q = timeit(lambda: ops.gemm(x, lw, ln_stats=st, out=q))
    t_u = timeit(unfused)
    t_f = timeit(lambda: ops.xattn_q(x, lw, st, img, HW, L, heads, d, n_rep=rep, out=out))
    t_p = timeit(lambda: ops.xattn_pack_kv(kd, vtd, rep * B, L, heads, d, out=img))
    bytes_f = 2 * M * C + 2 * rep * M * C
    print(f'd {d} B {B} HW {HW} rep {rep}: q-proj {t_q:.1f} us, q-proj + attention {t_u:.1f} us, fused {t_f:.1f} us '
          f'({bytes_f / t_f / 1e6:.2f} TB/s over x read + out write), pack (once per context) {t_p:.1f} us')
