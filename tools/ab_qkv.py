'''Self-attention q | k | v projection: one launch with a transposed tail (ops.gemm_qkv, fd_gemm_desc.trans_n0) against the two launches it replaces
(q|k GEMM through the rule + V^T GEMM), per level of the bench forward, interleaved; then the whole forward with the merge on / off per level.
    python tools/ab_qkv.py'''
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, hip, ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for (B, HW, C) in ((16, 4096, 320), (8, 4096, 320), (16, 1024, 640), (16, 256, 1280), (16, 64, 1280)):
    M = B * HW
    x = torch.randn((M, C), generator=g).half().to(dev)
    gamma, beta = torch.ones(C), torch.zeros(C)
    wqk, wv = torch.randn((2 * C, C), generator=g) * C ** -0.5, torch.randn((C, C), generator=g) * C ** -0.5
    lqk, lv = ops.prep_linear_ln(wqk, None, gamma, beta, dev), ops.prep_linear_ln(wv, None, gamma, beta, dev)
    lqkv = ops.prep_linear_ln(torch.cat([wqk, wv], 0), None, gamma, beta, dev)
    st = ops.ln_row_stats(x)
    rows = []
    for _ in range(3):
        a = t(lambda: ops.gemm_qkv(x, lqkv, B, HW, st))
        b = t(lambda: (ops.gemm(x, lqk, ln_stats=st), ops.gemm_vt(x, lv, B, HW, HW, ln_stats=st)))
        rows.append((a, b))
    a, b = sorted(r[0] for r in rows)[1], sorted(r[1] for r in rows)[1]
    print(f'M {M:6d} C {C:4d}: one launch {" ".join(f"{r[0]:.1f}" for r in rows)} us | two launches {" ".join(f"{r[1]:.1f}" for r in rows)} us -> {a:.1f} vs {b:.1f} ({b - a:+.1f} us)', flush=True)

# the whole forward, per-level choices
from flexdiffuse_amd.unet import UNet2DConditionModel
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet',))
unet = UNet2DConditionModel(sds['unet'], build.configs('sd15')[0], dev)
xl = torch.randn((8, 4, 64, 64), device=dev); ctx = torch.randn((16, 77, 768), device=dev).half()
temb = unet.time_bias(400.0, 1).expand(16, -1).contiguous()
arms = {'all levels': (0, 1 << 30), 'off': (1 << 30, 0), 'level 0 only (M >= 32768)': (32768, 1 << 30), 'levels 0-1 (M >= 16384)': (16384, 1 << 30),
        'levels 1-3 (M <= 16384)': (0, 16384)}
plans = {}
for name, (lo, hi) in arms.items():
    os.environ['FD_UNET_QKV_MIN_ROWS'], os.environ['FD_UNET_QKV_MAX_ROWS'] = str(lo), str(hi)
    unet.forward_nhwc(xl, 400, ctx, rep=2, temb=temb)
    pool = torch.cuda.MemPool(); plan = hip.Plan()
    with torch.cuda.use_mem_pool(pool, device=dev), plan.record():
        eps = unet.forward_nhwc(xl, 400, ctx, rep=2, temb=temb)
    plans[name] = (plan, pool, eps, len(plan))
torch.cuda.synchronize()
res = {a: [] for a in arms}
for r in range(6):
    for a in arms:
        plan = plans[a][0]
        for _ in range(3): plan.replay()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): plan.replay()
        torch.cuda.synchronize()
        res[a].append(1e3 * (time.time() - t0) / 20)
base = plans['off'][2]
for a in arms:
    v = sorted(res[a])
    print(f'{a}: {plans[a][3]} launches per forward; ms per forward {" ".join(f"{x:.3f}" for x in res[a])}; median {0.5 * (v[2] + v[3]):.3f}; '
          f'bit-identical to "off": {torch.equal(plans[a][2], base)}')
