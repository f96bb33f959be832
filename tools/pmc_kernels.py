import sys; sys.path.insert(0,'/root/repo')
import torch
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
only = sys.argv[1] if len(sys.argv) > 1 else ''
if only == 'conv':
    x = ops.Act(torch.randn((16*64*64,320), device=dev).half(), 16,64,64)
    w = ops.prep_conv(torch.randn((320,320,3,3))*(9*320)**-0.5, torch.randn(320), dev)
    for _ in range(3): ops.conv2d(x, w)
    torch.cuda.synchronize(); sys.exit(0)
if only == 'attn':
    for (B,N,heads,d) in [(16,4096,8,40),(16,1024,8,80)]:
        C=heads*d
        q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*N,C),device=dev).half()
        vt = torch.randn((B,C,N),device=dev).half()
        for _ in range(3): ops.attention(q,k,vt,B,heads,N,N,d,q_prescaled=True)
    torch.cuda.synchronize(); sys.exit(0)
B,N,heads,d = 16,4096,8,40
C=heads*d
q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*N,C),device=dev).half()
vt = torch.randn((B,C,N),device=dev).half()
for _ in range(3): ops.attention(q,k,vt,B,heads,N,N,d)
x = ops.Act(torch.randn((16*64*64,320), device=dev).half(), 16,64,64)
w = ops.prep_conv(torch.randn((320,320,3,3))*(9*320)**-0.5, torch.randn(320), dev)
for _ in range(3): ops.conv2d(x, w)
a = torch.randn((65536,320), device=dev).half(); lw = ops.prep_linear(torch.randn((320,320))*320**-0.5, torch.randn(320), dev)
for _ in range(3): ops.gemm(a, lw)
gx = ops.Act(torch.randn((16*4096,320), device=dev).half(), 16, 4096, 1)
gg = torch.ones(320, device=dev); gb = torch.zeros(320, device=dev)
for _ in range(3): ops.groupnorm(gx, gg, gb, 32, 1e-5, True)
B,N,heads,d = 16,1024,8,80
C=heads*d
q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*N,C),device=dev).half()
vt = torch.randn((B,C,N),device=dev).half()
for _ in range(3): ops.attention(q,k,vt,B,heads,N,N,d)
gw = ops.prep_geglu(torch.randn((2560,320))*320**-0.5, torch.randn(2560), dev)
for _ in range(3): ops.gemm(a, gw, act=ops.ACT_GEGLU)
torch.cuda.synchronize()
