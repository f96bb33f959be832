'''Error structure of fd_xattn_q_f16 against the two launches it replaces (per head, channel, 16-row block) and a
host-side decode of the packed K / V^T images against the tensors they were packed from.  (This is how the mixed
K=32 / K=16 MFMA chain problem was located: images and Q exact, denominators wrong in random 16-row blocks.)'''
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
B, HW, rep, L = 2, 1024, 1, 77
C, heads, d = 320, 8, 40
M = B * HW
g = torch.Generator().manual_seed(1)
x = (torch.randn((M, C), generator=g) * 1.3 + torch.randn((M, 1), generator=g)).half()
gamma = 1.0 + 0.3 * torch.randn(C, generator=g); beta = 0.2 * torch.randn(C, generator=g)
wq = torch.randn((C, C), generator=g) * C ** -0.5 * 1.5
lw = ops.prep_linear_ln(wq * ops.QK_LOG2E * d ** -0.5, None, gamma, beta, dev)
k = (torch.randn((rep * B * L, C), generator=g) * 1.2).half()
v = torch.randn((rep * B, L, C), generator=g).half()
ldv = (L + 7) // 8 * 8
vt = torch.zeros((rep * B, C, ldv), dtype=torch.float16); vt[:, :, :L] = v.permute(0, 2, 1)
xd, kd, vtd = x.to(dev), k.to(dev), vt.to(dev)
st = ops.ln_row_stats(xd)
img = ops.xattn_pack_kv(kd, vtd, rep * B, L, heads, d)
q = ops.gemm(xd, lw, ln_stats=st)
got = ops.xattn_q(xd, lw, st, img, HW, L, heads, d, n_rep=rep).float().cpu()
want = ops.attention(q, kd, vtd, B, heads, HW, L, d, q_prescaled=True).float().cpu()
err = (got - want).abs()
print('max err', float(err.max()), 'max want', float(want.abs().max()), 'nan', int(torch.isnan(got).sum()))
print('per head:', [round(float(err[:, h * d:(h + 1) * d].max()), 4) for h in range(heads)])
print('per channel of head 0:', [round(float(err[:, c].max()), 3) for c in range(d)])
print('per channel of head 1:', [round(float(err[:, d + c].max()), 3) for c in range(d)])
print('per 16-row block of first 256 rows:', [round(float(err[r:r + 16].max()), 3) for r in range(0, 256, 16)])
print('per 256-row tile:', [round(float(err[r:r + 256].max()), 3) for r in range(0, M, 256)])
r = int(err.max(dim=1).values.argmax())
print('worst row', r, 'got', got[r, :8].tolist(), 'want', want[r, :8].tolist())
import numpy as np
print('err table [head][16-row block of first 256 rows]:')
for h in range(heads):
    print(h, [round(float(err[r:r + 16, h * d:(h + 1) * d].max()), 2) for r in range(0, 256, 16)])
# decode the images on the host and compare with K / V^T
ki = img[0].cpu().numpy().view(np.float16).reshape(rep * B, heads, 5, 768)
vi = img[1].cpu().numpy().view(np.float16).reshape(rep * B, heads, 3, 1280)
kk = k.numpy().reshape(rep * B, L, heads, d)
vv = v.numpy().reshape(rep * B, L, heads, d)
bad_k = bad_v = 0
for b in range(rep * B):
    for h in range(heads):
        odd = h & 1
        for kb in range(5):
            for lane in range(64):
                fr, fq = lane & 15, lane >> 4
                key = kb * 16 + fr
                for i in range(8):
                    ch = (8 if odd else 0) + (fq * 4 + i if i < 4 else 16 + fq * 4 + i - 4)
                    w = kk[b, key, h, ch] if key < L else 0
                    bad_k += ki[b, h, kb, lane * 8 + i] != w
                for i in range(4):
                    w = 0
                    if key < L:
                        if not odd and fq < 2: w = kk[b, key, h, 32 + fq * 4 + i]
                        if odd and fq >= 2: w = kk[b, key, h, (fq - 2) * 4 + i]
                    bad_k += ki[b, h, kb, 512 + lane * 4 + i] != w
        for dt in range(3):
            for kg in range(3):
                for lane in range(64):
                    fr, fq = lane & 15, lane >> 4
                    dd = dt * 16 + fr
                    n = 4 if kg == 2 else 8
                    for i in range(n):
                        key = 64 + fq * 4 + i if kg == 2 else ((2 * kg) * 16 + fq * 4 + i if i < 4 else (2 * kg + 1) * 16 + fq * 4 + i - 4)
                        w = 0
                        if key < L:
                            w = vv[b, key, h, dd] if dd < d else (1 if dd == d else 0)
                        got_v = vi[b, h, dt, (0, 512, 1024)[kg] + lane * n + i]
                        bad_v += got_v != w
print('image mismatches: K', int(bad_k), 'V', int(bad_v))
