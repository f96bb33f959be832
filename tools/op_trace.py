'''Per-launch timing of ONE CFG UNet forward (headline shape) through the eager front: every C-ABI call is bracketed by two
events on the launch stream; prints the launches grouped by (entry point, shape) with count, mean us and share.'''
import sys, os, ctypes, collections; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexdiffuse_amd import build, hip, ops
from flexdiffuse_amd.unet import UNet2DConditionModel
dev = torch.device('cuda:0')
# python tools/op_trace.py [preset = sd15] [latent size = 64] [batch = 8]   (c4: sd15 96 4; c5: sd21 96 8)
_a = [v for v in sys.argv[1:] if not v.startswith('--')]
PRESET = _a[0] if len(_a) > 0 else 'sd15'
LAT = int(_a[1]) if len(_a) > 1 else 64
B = int(_a[2]) if len(_a) > 2 else 8
sds = build.synthetic_state_dicts(PRESET, seed=0, parts=('unet',))
ucfg = build.configs(PRESET)[0]
unet = UNet2DConditionModel(sds['unet'], ucfg, dev)
x = torch.randn((B, 4, LAT, LAT), device=dev)
ctx = torch.randn((2 * B, 77, ucfg.cross_attention_dim), device=dev).half()
for i in range(3):
    unet.forward_nhwc(x, 400 - i, ctx, rep=2)
torch.cuda.synchronize()
recs = []
orig = hip.call


def describe(name, args):
    a0 = args[0]
    obj = getattr(a0, '_obj', None)
    if isinstance(obj, ops.fd_gemm_desc):
        d = obj
        kind = 'conv' if d.conv else 'gemm'
        extra = ''
        if d.conv:
            extra = f' {d.kh}x{d.kw} s{d.stride}' + (' up' if d.upsample2x == 1 else ' phases' if d.upsample2x == 2 else '')
        if d.K2: extra += f' +K2 {d.K2}'
        if d.trans_out: extra += ' V^T'
        if d.act == 4: extra += ' GEGLU'
        if d.residual: extra += ' +res'
        if d.ln_stats: extra += ' LNfold'
        if d.ln_stats_out: extra += ' stats'
        fl = 2.0 * d.M * d.N * (d.K + d.K2) * max(d.batch, 1)
        return f'{kind} M{d.M} N{d.N} K{d.K}{extra}' + (f' x{d.batch}' if d.batch > 1 else '') + f' #GF{fl / 1e9:.2f}'
    if isinstance(obj, ops.fd_attention_desc):
        d = obj
        return f'attention B{d.batch} h{d.heads} nq{d.n_q} nk{d.n_k} d{d.head_dim}'
    if isinstance(obj, ops.fd_xattn_desc):
        d = obj
        return f'xattn_q M{d.M} rep{d.n_rep}'
    if name == 'fd_groupnorm_fold_linear_f16':
        return f'groupnorm_fold B{args[3]} HW{args[4]} C{args[5]} N{args[11]}'
    if name.startswith('fd_groupnorm'):
        return f'groupnorm B{args[6]} HW{args[7]} C{args[8]} silu{args[11]}'
    if name == 'fd_ln_row_stats_f16':
        return f'ln_row_stats rows{args[2]} C{args[3]}'
    if name == 'fd_copy2d_f16':
        return f'copy2d rows{args[4]} cols{args[5]}'
    return name


def traced(name, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(name, *args)
    e1.record()
    recs.append((describe(name, args), e0, e1))


hip.call = traced
if '--vae' in sys.argv:   # the VAE decode of B images instead of the UNet forward:  python tools/op_trace.py sd15 64 8 --vae
    from flexdiffuse_amd.vae import AutoencoderKL
    hip.call = orig
    vae = AutoencoderKL(build.synthetic_state_dicts(PRESET, seed=0, parts=('vae',))['vae'], build.configs(PRESET)[1], dev, encoder=False)
    for _ in range(2):
        vae.decode_nhwc(x)
    torch.cuda.synchronize()
    hip.call = traced
    vae.decode_nhwc(x)
else:
    unet.forward_nhwc(x, 390, ctx, rep=2)
torch.cuda.synchronize()
hip.call = orig
agg = collections.OrderedDict()
for k, e0, e1 in recs:
    us = 1e3 * e0.elapsed_time(e1)
    n, t = agg.get(k, (0, 0.0))
    agg[k] = (n + 1, t + us)
tot = sum(t for _, t in agg.values())
print(f'{len(recs)} launches, {tot / 1e3:.2f} ms bracketed (each bracket adds ~3-6 us)')
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    gf = float(k.split('#GF')[1]) if '#GF' in k else 0.0   # MFMA work issued (the parity upsample: 4/9 of the algorithmic count)
    tf = f'  {gf / (t / n) * 1e3:6.0f} TF/s, {t - n * gf * 1e3 / 1100:7.0f} us above 1100 TF/s' if gf else ''
    print(f'{100 * t / tot:5.1f} %  {n:3d} x {t / n:8.1f} us  {k.split(" #GF")[0]}{tf}')
