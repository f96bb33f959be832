'''Attention kernel A/B: FD_ATTN_QT1=1 (8 waves, exact running max) vs 2 (VALU-lean) vs
2 + q_prescaled, all in one GPU session (child processes, since the mode is read once).'''
import sys, os, subprocess
sys.path.insert(0, '/root/repo')
if len(sys.argv) == 1:
    modes = (('2', '1'),) if os.environ.get('AB_QUICK') else (('1', '0'), ('2', '0'), ('2', '1'), ('1', '0'), ('2', '1'))
    for mode, pre in modes:
        env = dict(os.environ, FD_ATTN_QT1=mode)
        subprocess.run([sys.executable, __file__, pre], env=env)
    sys.exit(0)
import torch
from flexdiffuse_amd import ops
pre = sys.argv[1] == '1'
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
out=[]
for (B,N,heads,d,Nk) in [(16,4096,8,40,4096),(8,4096,8,40,4096),(16,1024,8,80,1024),(16,256,8,160,256),(16,4096,8,40,77),(16,1024,8,80,77)]:
    C=heads*d
    q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*Nk,C),device=dev).half()
    vt = torch.randn((B,C,(Nk+7)//8*8),device=dev).half()
    t = timeit(lambda: ops.attention(q,k,vt,B,heads,N,Nk,d,q_prescaled=pre))
    out.append(f'{t*1e3:.1f}us({4.0*B*heads*N*Nk*d/t/1e9:.0f}TF)')
print('mode', os.environ.get('FD_ATTN_QT1'), 'pre', int(pre), ' '.join(out))
