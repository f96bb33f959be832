'''Record every fd_gemm_f16 shape of one SD1.5 UNet forward (CFG batch 16, 64x64 latents) and
one VAE decode (B=8), then time each (tile, split_k) candidate per unique shape.'''
import sys, ctypes, collections, json
sys.path.insert(0, '/root/repo')
import torch
from flexdiffuse_amd import hip, ops, build
dev = torch.device('cuda:0')
rec = collections.OrderedDict()
orig_call = hip.call
def spy(name, *args):
    if name == 'fd_gemm_f16':
        d = args[0]._obj
        key = (d.M, d.N, d.K, d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.stride, d.upsample2x,
               d.act, d.trans_out, d.batch, d.out_f32, bool(d.residual), bool(d.bias2), d.lda, d.ldc, d.pad_t)
        rec[key] = rec.get(key, 0) + 1
    return orig_call(name, *args)
hip.call = spy; ops.hip.call = spy
sds = build.synthetic_state_dicts('sd15', seed=0, parts=('unet','vae'))
from flexdiffuse_amd.unet import UNet2DConditionModel
from flexdiffuse_amd.vae import AutoencoderKL
unet = UNet2DConditionModel(sds['unet'], device=dev)
vae = AutoencoderKL(sds['vae'], device=dev, encoder=False)
x = torch.randn((8,4,64,64), device=dev); ctx = torch.randn((16,77,768), device=dev)
unet.forward_nhwc(x, 500, ctx, rep=2)
n_unet = dict(rec)
vae.decode_nhwc(x)
torch.cuda.synchronize()
hip.call = orig_call; ops.hip.call = orig_call
del unet, vae, sds
print('unique shapes', len(rec))
def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ws = torch.empty(256<<20, dtype=torch.uint8, device=dev)
results = []
tot_auto = tot_best = 0.0
for key, cnt in rec.items():
    (M,N,K,conv,ih,iw,ic,oh,ow,kh,stride,up,act,trans,batch,of32,hasres,hasb2,lda,ldc,pad_t) = key
    if trans or batch > 1: continue
    d = ops.fd_gemm_desc()
    if conv:
        B = M // (oh*ow)
        A = torch.randn((B*ih*iw, ic), device=dev).half()
    else:
        A = torch.randn((M, lda), device=dev).half()
    W = (torch.randn((N, K), device=dev) * K**-0.5).half()
    nout = N//2 if act == 4 else N
    C = torch.empty((M, max(ldc, (nout+3)//4*4)), device=dev, dtype=torch.float32 if of32 else torch.float16)
    bias = torch.randn(((N+3)//4*4,), device=dev)
    res = torch.randn((M, C.shape[1]), device=dev).half() if hasres else None
    b2 = torch.randn((max(1, M // max(1,(oh*ow if conv else M))), N), device=dev) if hasb2 else None
    d.A, d.W, d.C, d.bias = A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr()
    d.residual = res.data_ptr() if res is not None else None
    d.bias2 = b2.data_ptr() if b2 is not None else None
    d.M, d.N, d.K, d.lda, d.ldw, d.ldc = M, N, K, lda, K, C.shape[1]
    d.ldr = C.shape[1] if hasres else 0
    d.ld_bias2 = N; d.rows_per_sample = oh*ow if conv else 0
    d.act, d.out_f32, d.alpha, d.batch = act, of32, 1.0, 1
    d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.kw = conv, ih, iw, ic, oh, ow, kh, kh
    d.stride, d.pad_t, d.pad_l, d.upsample2x = stride, pad_t, pad_t, up
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    st = hip.stream()
    def run(tile, sk):
        d.tile, d.split_k = tile, sk
        hip.call('fd_gemm_f16', ctypes.byref(d), st)
    t_auto = timeit(lambda: run(0, 0))
    best = (1e9, None)
    row = {}
    for tile in (1,2,3,4,6,9,10,11,12,13,14,15,16,20):
        if act == 4 and tile in (2,5,7,9,12,13,16,20): continue
        if tile in (15,16) and N % (256 if tile == 15 else 320): continue
        for sk in (1,2,4,8,16):
            if sk > 1 and (act == 4 or (K//64)//sk < 4 or sk*M*N*4 > ws.numel()): continue
            try:
                t = timeit(lambda: run(tile, sk), n=6)
            except Exception as e:
                continue
            row[(tile,sk)] = t
            if t < best[0]: best = (t, (tile, sk))
    fl = 2.0*M*N*K
    tot_auto += t_auto*cnt; tot_best += best[0]*cnt
    top = sorted(row.items(), key=lambda kv: kv[1])[:4]
    print(f'M={M:6d} N={N:5d} K={K:5d} conv={conv} act={act} x{cnt:3d}: auto {t_auto*1e3:7.1f}us ({fl/t_auto/1e9:5.0f}TF) best {best[1]} {best[0]*1e3:7.1f}us ({fl/best[0]/1e9:5.0f}TF)  top: ' + ' '.join(f'{k}:{v*1e3:.0f}' for k,v in top))
print(f'TOTAL per (unet fwd + vae decode): auto {tot_auto:.2f} ms, best {tot_best:.2f} ms')
