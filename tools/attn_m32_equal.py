'''k_attention_w8q2p (FD_ATTN_M32=2) must give the bits of k_attention_w8q2m (FD_ATTN_M32=1): each arm in its own process writes its output, the parent compares.
    python tools/attn_m32_equal.py'''
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for arm in ('1', '2'):
        env = dict(os.environ); env['FD_ATTN_M32'] = arm
        subprocess.run([sys.executable, os.path.abspath(__file__), arm], env=env, check=True)
    import torch
    a, b = torch.load('/tmp/attn_m32_1.pt'), torch.load('/tmp/attn_m32_2.pt')
    for (na, xa), (nb, xb) in zip(a, b):
        print(na, 'equal' if torch.equal(xa, xb) else f'DIFFER max |d| {(xa.float() - xb.float()).abs().max().item():.3e}', flush=True)
    sys.exit(0)
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
out = []
for (B, N, heads, d, Nk, causal) in [(2, 4096, 8, 40, 4096, False), (1, 2091, 3, 40, 1093, False), (1, 2500, 1, 40, 2500, True), (2, 4096, 8, 40, 77, False),
                                     (1, 2048, 2, 40, 64, False), (1, 2048, 2, 40, 128, False), (1, 2048, 2, 40, 192, False), (1, 9216, 2, 40, 9216, False)]:
    C = heads * d
    g = torch.Generator(device='cpu').manual_seed(5)
    q = (torch.randn((B * N, C), generator=g) * (d ** -0.5 * ops.QK_LOG2E)).half().to(dev)
    k = torch.randn((B * Nk, C), generator=g).half().to(dev)
    k[Nk // 2:] *= 3.0
    vt = torch.randn((B, C, (Nk + 7) // 8 * 8), generator=g).half().to(dev)
    for rep in range(3):
        o = ops.attention(q, k, vt, B, heads, N, Nk, d, causal, q_prescaled=True)
        out.append((f'{B}x{N}x{Nk} causal={causal} rep {rep}', o.cpu()))
torch.save(out, f'/tmp/attn_m32_{sys.argv[1]}.pt')
