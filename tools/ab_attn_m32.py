'''Level-0 self-attention (B 16, N 4096, 8 heads x 40, Q pre-scaled): the 32x32x16 QK^T kernel (k_attention_w8q2m, default) against the 16x16x32
form (FD_ATTN_M32=0), each arm in its own process, interleaved; then the other shapes the dispatch sends there.
    python tools/ab_attn_m32.py'''
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    arms = [('1', None), ('0', None)] + [(os.environ.get('AB_VARIANT_ARM', '1'), f) for f in sorted(__import__('glob').glob(os.path.join(ROOT, 'tools/_variants/libfd_m32_*.so')))]
    for arm, lib in arms * 3:
        env = dict(os.environ); env['FD_ATTN_M32'] = arm
        if lib: env['FD_LIB_PATH'] = lib
        subprocess.run([sys.executable, os.path.abspath(__file__), arm + (' ' + os.path.basename(lib) if lib else '')], env=env)
    sys.exit(0)
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for (B, N, heads, d, Nk) in [(16, 4096, 8, 40, 4096), (16, 4096, 8, 40, 77), (4, 9216, 8, 40, 9216), (2, 4096, 8, 40, 4096)]:
    C = heads * d
    g = torch.Generator(device='cpu').manual_seed(5)
    q = (torch.randn((B * N, C), generator=g) * (d ** -0.5 * ops.QK_LOG2E)).half().to(dev)
    k = torch.randn((B * Nk, C), generator=g).half().to(dev)
    vt = torch.randn((B, C, (Nk + 7) // 8 * 8), generator=g).half().to(dev)
    t = timeit(lambda: ops.attention(q, k, vt, B, heads, N, Nk, d, q_prescaled=True))
    o = ops.attention(q, k, vt, B, heads, N, Nk, d, q_prescaled=True)
    out.append(f'{B}x{N}x{Nk}: {t * 1e3:.1f} us ({4.0 * B * heads * N * Nk * d / t / 1e9:.0f} TF/s, sum {o.float().abs().sum().item():.6e})')
print(f'FD_ATTN_M32={sys.argv[1]}: ' + ' | '.join(out), flush=True)
