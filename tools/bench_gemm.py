import sys; sys.path.insert(0,'/root/repo')
import torch, time
from flexdiffuse_amd import ops, hip
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
print('--- linear')
for (M,N,K) in [(4096,4096,4096),(8192,8192,8192),(65536,320,320),(65536,2560,320),(65536,320,1280),(16384,640,640),(16384,5120,640),(4096,1280,1280),(4096,10240,1280),(1024,1280,1280)]:
    a = torch.randn((M,K), device=dev).half(); w = ops.prep_linear(torch.randn((N,K))*K**-0.5, torch.randn(N), dev)
    ms = timeit(lambda: ops.gemm(a, w))
    print(f'M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TF')
print('--- geglu')
for (M,C) in [(65536,320),(16384,640),(4096,1280)]:
    a = torch.randn((M,C), device=dev).half(); w = ops.prep_geglu(torch.randn((8*C,C))*C**-0.5, torch.randn(8*C), dev)
    ms = timeit(lambda: ops.gemm(a, w, act=ops.ACT_GEGLU))
    print(f'M={M} N={8*C} K={C}: {ms*1e3:.1f} us  {2*M*8*C*C/ms/1e9:.1f} TF')
print('--- conv3x3')
for (B,H,Cin,Cout) in [(16,64,320,320),(16,64,640,320),(16,64,960,320),(16,32,640,640),(16,32,1280,640),(16,32,1920,640),(16,16,1280,1280),(16,16,2560,1280),(16,8,1280,1280),(16,8,2560,1280),(8,512,128,128),(8,256,256,256)]:
    x = ops.Act(torch.randn((B*H*H,Cin), device=dev).half(), B,H,H)
    w = ops.prep_conv(torch.randn((Cout,Cin,3,3))*(9*Cin)**-0.5, torch.randn(Cout), dev)
    ms = timeit(lambda: ops.conv2d(x, w), n=10)
    fl = 2*B*H*H*Cout*9*Cin
    print(f'B={B} H={H} Cin={Cin} Cout={Cout}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF')
print('--- attention')
for (B,N,heads,d,Nk) in [(16,4096,8,40,4096),(16,1024,8,80,1024),(16,256,8,160,256),(16,4096,8,40,77),(16,1024,8,80,77)]:
    C=heads*d
    q = torch.randn((B*N,C),device=dev).half(); k = torch.randn((B*Nk,C),device=dev).half()
    ld=(Nk+7)//8*8
    vt = torch.randn((B,C,ld),device=dev).half()
    ms = timeit(lambda: ops.attention(q,k,vt,B,heads,N,Nk,d), n=10)
    print(f'B={B} N={N} Nk={Nk} d={d}: {ms*1e3:.1f} us  {4*B*heads*N*Nk*d/ms/1e9:.1f} TF')
print('--- groupnorm')
for (B,HW,C) in [(16,4096,320),(16,4096,960),(16,1024,640),(16,256,1280),(16,64,2560),(8,262144,128)]:
    x = ops.Act(torch.randn((B*HW,C),device=dev).half(),B,HW,1)
    g = torch.ones(C,device=dev); b=torch.zeros(C,device=dev)
    ms = timeit(lambda: ops.groupnorm(x,g,b,32,1e-5,True), n=10)
    print(f'B={B} HW={HW} C={C}: {ms*1e3:.1f} us  {B*HW*C*4/ms/1e6:.1f} GB/s (4B/elem)')
