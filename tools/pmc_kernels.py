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
if only == 'r03':
    # round-3 evidence set: level-0 conv (256x320), level-0 GEGLU (256x256), short-K linear GEMMs with residual /
    # statistics emission, the GroupNorm pair, the fused q-projection + cross-attention, a split-K 8x8-level conv
    g = torch.Generator().manual_seed(0)
    x = ops.Act((torch.randn((16*64*64,320), generator=g)*0.7).half().to(dev), 16,64,64)
    w = ops.prep_conv(torch.randn((320,320,3,3), generator=g)*(9*320)**-0.5, torch.randn(320, generator=g), dev)
    for _ in range(3): ops.conv2d(x, w)
    a = (torch.randn((65536,320), generator=g)).half().to(dev)
    res = (torch.randn((65536,320), generator=g)).half().to(dev)
    lw = ops.prep_linear(torch.randn((320,320), generator=g)*320**-0.5, torch.randn(320, generator=g), dev)
    st = torch.empty((65536,2), dtype=torch.float32, device=dev)
    for _ in range(3): ops.gemm(a, lw, residual=res, ln_stats_out=st)      # EPI 9: o1 / o2 / proj_in
    lq = ops.prep_linear_ln(torch.randn((320,320), generator=g)*320**-0.5*0.23, None, torch.ones(320), torch.zeros(320), dev)
    for _ in range(3): ops.gemm(a, lq, ln_stats=st)                        # EPI 5: the q projection the fused kernel replaces
    gw = ops.prep_linear_ln(torch.randn((2560,320), generator=g)*320**-0.5, torch.randn(2560, generator=g), torch.ones(320), torch.zeros(320), dev, geglu=True)
    for _ in range(3): ops.gemm(a, gw, act=ops.ACT_GEGLU, ln_stats=st)     # level-0 GEGLU with the fold
    gg = torch.ones(320, device=dev); gb = torch.zeros(320, device=dev)
    for _ in range(3): ops.groupnorm(x, gg, gb, 32, 1e-5, True)
    L = 77
    kd = torch.randn((16*L,320), generator=g).half().to(dev)
    vt = torch.zeros((16,320,80), dtype=torch.float16); vt[:,:,:L] = torch.randn((16,320,L), generator=g).half()
    vt = vt.to(dev)
    img = ops.xattn_pack_kv(kd, vt, 16, L, 8, 40)
    for _ in range(3): ops.xattn_q(a, lq, st, img, 4096, L, 8, 40)
    q = ops.gemm(a, lq, ln_stats=st)
    for _ in range(3): ops.attention(q, kd, vt, 16, 8, 4096, L, 40, q_prescaled=True)   # the cross-attention launch it replaces
    x3 = ops.Act((torch.randn((16*8*8,1280), generator=g)*0.7).half().to(dev), 16,8,8)
    w3 = ops.prep_conv(torch.randn((1280,1280,3,3), generator=g)*(9*1280)**-0.5, torch.randn(1280, generator=g), dev)
    for _ in range(3): ops.conv2d(x3, w3)
    # late round 3: the 288x160 tile on a 768x768 level-0 conv (c4: CFG batch 8 x 96x96 = 73728 rows), the head-dim-80 fused
    # cross-attention (32x32 level), a 640-wide producer that emits per-n-tile partial LayerNorm sums
    x4 = ops.Act((torch.randn((8*96*96,320), generator=g)*0.7).half().to(dev), 8,96,96)
    for _ in range(3): ops.conv2d(x4, w)
    a1 = torch.randn((16384,640), generator=g).half().to(dev)
    r1 = torch.randn((16384,640), generator=g).half().to(dev)
    l1 = ops.prep_linear(torch.randn((640,640), generator=g)*640**-0.5, torch.randn(640, generator=g), dev)
    parts = torch.empty((ops.can_emit_row_stats(16384, 640, 640), 16384, 2), dtype=torch.float32, device=dev)
    for _ in range(3): ops.gemm(a1, l1, residual=r1, ln_stats_out=parts)
    st1 = ops.ln_finalize_stats(parts, 640)
    lq1 = ops.prep_linear_ln(torch.randn((640,640), generator=g)*640**-0.5*0.23, None, torch.ones(640), torch.zeros(640), dev)
    kd1 = torch.randn((16*L,640), generator=g).half().to(dev)
    vt1 = torch.zeros((16,640,80), dtype=torch.float16); vt1[:,:,:L] = torch.randn((16,640,L), generator=g).half()
    img1 = ops.xattn_pack_kv(kd1, vt1.to(dev), 16, L, 8, 80)
    for _ in range(3): ops.xattn_q(a1, lq1, st1, img1, 1024, L, 8, 80)
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
