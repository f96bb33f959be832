'''Record every fd_gemm_f16 launch of one CFG UNet forward (and optionally one VAE decode) BY VALUE -- the descriptor with all of its
fields: appended operand (A2 / K2), LayerNorm fold, statistics emission, parity upsample (batch 4) -- and keep every buffer it points to
alive, so that each unique launch can be re-issued with a forced (tile, split_k).  Used by tools/sweep_gemm.py (exhaustive sweep) and by
tests/test_gpu_gemm_rule.py (performance guard of the tile rule).'''
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def record(preset='sd15', lat=64, batch=8, vae=True, dev=None, unet=True):
    '''-> (OrderedDict key -> [fd_gemm_desc copy, launches per forward], list of tensors that must stay alive)'''
    import torch
    from flexdiffuse_amd import build, hip, ops
    from flexdiffuse_amd.unet import UNet2DConditionModel
    from flexdiffuse_amd.vae import AutoencoderKL
    dev = dev or torch.device('cuda:0')
    rec = collections.OrderedDict()
    keep = []
    orig_call, orig_empty = hip.call, ops._empty

    def spy(name, *a):
        if name == 'fd_gemm_f16':
            d = a[0]._obj
            key = (d.M, d.N, d.K, d.K2, d.conv, d.in_h, d.in_w, d.in_c, d.kh, d.stride, d.upsample2x, d.act, d.trans_out, d.batch,
                   d.out_f32, bool(d.residual), bool(d.bias2), bool(d.ln_stats), bool(d.ln_stats_out), d.lda, d.ldc)
            if key not in rec:
                c = ops.fd_gemm_desc()
                ctypes.memmove(ctypes.byref(c), ctypes.byref(d), ctypes.sizeof(d))
                rec[key] = [c, 0]
            rec[key][1] += 1
        return orig_call(name, *a)

    def keep_empty(*a, **k):
        t = orig_empty(*a, **k)
        keep.append(t)
        return t

    te = torch.empty
    hip.call = spy
    ops.hip.call = spy
    ops._empty = keep_empty
    torch.empty = lambda *a, **k: (keep.append(te(*a, **k)) or keep[-1])
    try:
        sds = build.synthetic_state_dicts(preset, seed=0, parts=(('unet',) if unet else ()) + (('vae',) if vae else ()))
        ucfg, vcfg, _ = build.configs(preset)
        x = torch.randn((batch, 4, lat, lat), device=dev)
        keep.append(x)
        if unet:
            net = UNet2DConditionModel(sds['unet'], ucfg, dev)
            ctx = torch.randn((2 * batch, 77, ucfg.cross_attention_dim), device=dev).half()
            keep += [ctx, net]
            net.forward_nhwc(x, 500, ctx, rep=2)
        if vae:
            v = AutoencoderKL(sds['vae'], vcfg, device=dev, encoder=False)
            keep.append(v)
            v.decode_nhwc(x)
        torch.cuda.synchronize()
    finally:
        hip.call = orig_call
        ops.hip.call = orig_call
        ops._empty = orig_empty
        torch.empty = te
    return rec, keep


KEY_FIELDS = ('M', 'N', 'K', 'K2', 'conv', 'ih', 'iw', 'ic', 'kh', 'stride', 'up', 'act', 'trans', 'batch', 'of32', 'hasres', 'hasb2', 'lnf',
              'lno', 'lda', 'ldc')


def describe(key) -> str:
    k = dict(zip(KEY_FIELDS, key))
    return (('conv ' if k['conv'] else 'gemm ') + ('up%d ' % k['up'] if k['up'] else '') + (f"+K2 {k['K2']} " if k['K2'] else '') +
            ('LNfold ' if k['lnf'] else '') + ('stats ' if k['lno'] else '') + ('V^T ' if k['trans'] else '') + ('res ' if k['hasres'] else '') +
            (f"x{k['batch']} " if k['batch'] > 1 else '') + ('GEGLU ' if k['act'] == 4 else '')).strip()
