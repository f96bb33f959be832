import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch, numpy as np
from test_gpu_kernels import rnd, attn_ref
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
Nq,Nk,heads,d = 257,257,4,64
B, C = 2, heads*d
for (qs, ks) in [(6,6),(6,1),(1,6),(3,3)]:
    q, k, v = rnd((B, Nq, C), 1), rnd((B, Nk, C), 2), rnd((B, Nk, C), 3)
    q[0, 3] *= qs; k[0, Nk - 2] *= ks
    ld = (Nk + 7)//8*8
    vt = torch.zeros((B, C, ld), dtype=torch.float16); vt[:, :, :Nk] = v.transpose(1,2).half()
    out = ops.attention(q.half().to(dev).view(B*Nq, C), k.half().to(dev).view(B*Nk, C), vt.to(dev), B, heads, Nq, Nk, d, False).view(B,Nq,C).float().cpu()
    want = attn_ref(q,k,v,heads,False)
    err = (out-want).abs()
    idx = np.unravel_index(err.argmax().item(), err.shape)
    h = idx[2]//d
    s = (q[0,3].view(heads,d)[h] @ k[0].view(Nk,heads,d)[:,h].T) * d**-0.5
    top = s.topk(4)
    print((qs,ks), 'maxerr', err.max().item(), 'at', idx, 'want', want[idx].item(), 'got', out[idx].item())
    print('   row3 head', h, 'top scores', top.values.tolist(), top.indices.tolist(), ' fp16(q) max', q[0,3].abs().max().item())
    # reference with P rounded to fp16
    p = (s - s.max()).exp(); 
    o16 = (p.half().float() @ v[0].view(Nk,heads,d)[:,h]) / p.sum()
    o32 = (p @ v[0].view(Nk,heads,d)[:,h]) / p.sum()
    print('   ref32', o32[idx[2]-h*d].item(), 'ref p16', o16[idx[2]-h*d].item())
