'''Self-attention at the level-0 shape (B 16, N 4096, 8 heads x 40) across library variants
(FD_LIB_PATH): occupancy 8 / 6 / 4 waves per SIMD of k_attention_w8<64,3>.'''
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    libs = [('q2', None)] + [(os.path.basename(f)[4:-3], 'tools/_variants/' + os.path.basename(f))
                             for f in sorted(__import__('glob').glob(os.path.join(ROOT, 'tools/_variants/lib_att*.so')))]
    for name, lib in libs * 2:
        env = dict(os.environ)
        if lib and lib.startswith('ENV:'):
            k, v = lib[4:].split('='); env[k] = v
        elif lib: env['FD_LIB_PATH'] = os.path.join(ROOT, lib)
        subprocess.run([sys.executable, __file__, name], env=env)
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
for (B, N, heads, d, Nk) in [(16,4096,8,40,4096),(16,4096,8,40,77)]:
    C = heads * d
    q = torch.randn((B*N, C), device=dev).half(); k = torch.randn((B*Nk, C), device=dev).half()
    vt = torch.randn((B, C, (Nk+7)//8*8), device=dev).half()
    t = timeit(lambda: ops.attention(q, k, vt, B, heads, N, Nk, d, q_prescaled=True))
    out.append(f'{t*1e3:.1f}us({4.0*B*heads*N*Nk*d/t/1e9:.0f}TF)')
print(sys.argv[1], ' '.join(out), flush=True)
