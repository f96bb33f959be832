'''Self-attention at head dim 64 (SD2.1 at 768x768: 16 x 5 heads x 9216 x 9216 at the top level, 10 heads x 2304 below; Q pre-scaled): two query blocks per wave
(k_attention_w8q2<64,4>, FD_ATTN_Q2_64=1, default) against one (k_attention_w8<64,4>, FD_ATTN_Q2_64=0); variant libraries tools/_variants/libfd_q264_*.so
are further arms.  One process per arm, interleaved.
    python tools/ab_attn_q2_64.py'''
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    arms = [('1', None), ('0', None)] + [('1', f) for f in sorted(__import__('glob').glob(os.path.join(ROOT, 'tools/_variants/libfd_q264_*.so')))]
    for arm, lib in arms * 3:
        env = dict(os.environ); env['FD_ATTN_Q2_64'] = arm
        if lib: env['FD_LIB_PATH'] = lib
        subprocess.run([sys.executable, os.path.abspath(__file__), arm + (' ' + os.path.basename(lib) if lib else '')], env=env)
    sys.exit(0)
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for (B, N, heads, d) in [(16, 9216, 5, 64), (16, 2304, 10, 64), (16, 4096, 5, 64)]:
    C = heads * d
    g = torch.Generator(device='cpu').manual_seed(5)
    q = (torch.randn((B * N, C), generator=g) * (d ** -0.5 * ops.QK_LOG2E)).half().to(dev)
    k = torch.randn((B * N, C), generator=g).half().to(dev)
    vt = torch.randn((B, C, N), generator=g).half().to(dev)
    t = timeit(lambda: ops.attention(q, k, vt, B, heads, N, N, d, q_prescaled=True))
    o = ops.attention(q, k, vt, B, heads, N, N, d, q_prescaled=True)
    out.append(f'{B}x{heads}x{N}: {t * 1e3:.1f} us ({4.0 * B * heads * N * N * d / t / 1e9:.0f} TF/s, sum {o.float().abs().sum().item():.6e})')
print(f'FD_ATTN_Q2_64={sys.argv[1]}: ' + ' | '.join(out), flush=True)
