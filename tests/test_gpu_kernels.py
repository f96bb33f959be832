'''Per-kernel numerics of the HIP ops (through the C ABI) against plain PyTorch fp32
references of the same op on the same fp16-rounded inputs.  Needs an MI355X.

Tolerances: fp16 storage of outputs => relative 2^-10; accumulations are fp32.  Each test
states its bound.'''
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().float()


def close(got, want, rtol=2e-3, atol=2e-3):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    assert bool((err <= bound).all()), f'max err {err.max().item():.4g} (max |want| {want.abs().max().item():.4g})'


@pytest.mark.parametrize('M,N,K', [(256, 320, 320), (1000, 640, 1280), (77, 768, 768),
                                   (4096, 1280, 640), (130, 36, 72), (64, 8, 8), (16, 20160, 1280)])
def test_gemm_linear(dev, M, N, K):
    from flexdiffuse_amd import ops
    a, w, b = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3)
    res = rnd((M, N), 4)
    lw = ops.prep_linear(w, b, dev)
    ad = a.half().to(dev)
    if lw.K != K:
        ad = F.pad(ad, (0, lw.K - K))
    out = ops.gemm(ad, lw)
    close(out[:, :N], a @ w.T + b)
    out = ops.gemm(ad, lw, act=ops.ACT_SILU, residual=F.pad(res, (0, out.shape[1] - N)).half().to(dev))
    close(out[:, :N], F.silu(a @ w.T + b) + res)
    out = ops.gemm(ad, lw, act=ops.ACT_QUICK_GELU, out_f32=True)
    y = a @ w.T + b
    close(out[:, :N], y * torch.sigmoid(1.702 * y), rtol=1e-3, atol=1e-3)


def test_gemm_rule_cascade_on_shapes_outside_the_presets(dev):
    '''fd_gemm_f16's tile / split-K rules were fitted to the GEMM shapes of four model presets; every OTHER shape must
    still come out right (or be refused with FD_ESHAPE, never silently wrong).  48 seeded random (M, N, K) -- odd row
    counts, N not a multiple of any tile, K with a tail, tiny and skinny problems, rows of 9 x 2^k -- through the
    library's own choice (tile = split_k = 0), with bias, with and without a residual / activation, against fp32 torch;
    plus 16 random convolutions (odd maps, stride 1 / 2, fused nearest-2x upsample, channel counts outside the UNet's).'''
    from flexdiffuse_amd import ops
    rng = np.random.default_rng(20240)
    picks_m = [1, 3, 17, 64, 77, 130, 255, 256, 257, 1000, 1152, 2304, 4097, 9216, 20000]
    picks_n = [8, 20, 36, 64, 96, 160, 200, 320, 328, 480, 640, 1000, 1280, 2048]
    picks_k = [8, 24, 64, 72, 128, 200, 320, 520, 768, 1024, 1280, 2560, 4104]
    refused = 0
    for case in range(48):
        M, N, K = int(rng.choice(picks_m)), int(rng.choice(picks_n)), int(rng.choice(picks_k))
        a, w, b = rnd((M, K), 100 + case), rnd((N, K), 200 + case, K ** -0.5), rnd((N,), 300 + case)
        lw = ops.prep_linear(w, b, dev)
        ad = a.half().to(dev)
        if lw.K != K:
            ad = F.pad(ad, (0, lw.K - K))
        mode = case % 3
        try:
            if mode == 0:
                out = ops.gemm(ad, lw)
                want = a @ w.T + b
            elif mode == 1:
                res = rnd((M, N), 400 + case)
                out = ops.gemm(ad, lw, residual=F.pad(res, (0, (-N) % 4)).half().to(dev))
                want = a @ w.T + b + res
            else:
                out = ops.gemm(ad, lw, act=ops.ACT_SILU)
                want = F.silu(a @ w.T + b)
        except ValueError as ex:      # FD_EINVAL / FD_ESHAPE: a loud refusal is allowed, a wrong result is not
            print('refused', (M, N, K, mode), str(ex)[:120])
            refused += 1
            continue
        got = out[:, :N].float().cpu()
        err = (got - want).abs()
        assert bool((err <= 3e-3 + 3e-3 * want.abs()).all()), ('linear', M, N, K, mode, float(err.max()))
    assert refused <= 4, f'{refused} of 48 linear shapes refused'
    for case in range(16):
        B = int(rng.choice([1, 2, 3, 5]))
        cin, cout = int(rng.choice([64, 96, 128, 192, 320, 448, 640])), int(rng.choice([32, 64, 100, 128, 320, 384]))
        H, W = int(rng.choice([5, 8, 13, 16, 24, 33])), int(rng.choice([6, 8, 16, 20, 31]))
        stride, up = (2, False) if case % 4 == 1 else (1, case % 4 == 2)
        x, w, b = rnd((B, cin, H, W), 500 + case), rnd((cout, cin, 3, 3), 600 + case, (9 * cin) ** -0.5), rnd((cout,), 700 + case)
        cw = ops.prep_conv(w, b, dev)
        if up and cin % 64:
            with pytest.raises(ValueError):       # no upsample form on the explicit-im2col path: refused, not wrong
                ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), cw, stride=stride, up=up)
            up = False
        y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), cw, stride=stride, up=up)
        xin = F.interpolate(x, scale_factor=2.0, mode='nearest') if up else x
        want = F.conv2d(xin, w, b, stride=stride, padding=1)
        assert (y.H, y.W) == tuple(want.shape[-2:]), (case, y.H, y.W, want.shape)
        got = y.t.float().cpu().view(B, y.H, y.W, -1)[..., :cout].permute(0, 3, 1, 2)
        err = (got - want).abs()
        assert bool((err <= 3e-3 + 3e-3 * want.abs()).all()), ('conv', B, cin, cout, H, W, stride, up, float(err.max()))


def test_gemm_bias2_and_geglu(dev):
    from flexdiffuse_amd import ops
    B, HW, K, C = 4, 96, 320, 320
    a, w, b = rnd((B * HW, K), 1), rnd((C, K), 2, K ** -0.5), rnd((C,), 3)
    b2 = rnd((B, 3 * C), 5)
    lw = ops.prep_linear(w, b, dev)
    b2d = b2.to(dev)
    out = ops.gemm(a.half().to(dev), lw, bias2=b2d[:, C:2 * C], ld_bias2=3 * C, rows_per_sample=HW)
    want = (a @ w.T + b).view(B, HW, C) + b2[:, None, C:2 * C]
    close(out, want.view(B * HW, C))
    # GEGLU: proj [8C][C] -> value * gelu(gate)
    wg, bg = rnd((8 * C, K), 6, K ** -0.5), rnd((8 * C,), 7)
    lg = ops.prep_geglu(wg, bg, dev)
    out = ops.gemm(a.half().to(dev), lg, act=ops.ACT_GEGLU)
    y = a @ wg.T + bg
    val, gate = y.chunk(2, dim=-1)
    assert out.shape == (B * HW, 4 * C)
    close(out, val * F.gelu(gate), rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize('cin,cout,H,W,stride,up', [(64, 128, 16, 16, 1, False),
                                                   (320, 320, 32, 32, 1, False),
                                                   (640, 320, 8, 8, 1, True),
                                                   (128, 64, 16, 24, 2, False),
                                                   (960, 640, 16, 16, 1, False)])
def test_conv3x3_implicit(dev, cin, cout, H, W, stride, up):
    from flexdiffuse_amd import ops
    B = 3
    x, w, b = rnd((B, cin, H, W), 1), rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5), rnd((cout,), 3)
    cw = ops.prep_conv(w, b, dev)
    assert not cw.im2col
    xa = ops.nchw_to_nhwc(x.to(dev))
    y = ops.conv2d(xa, cw, stride=stride, up=up)
    xin = F.interpolate(x, scale_factor=2.0, mode='nearest') if up else x
    want = F.conv2d(xin, w, b, stride=stride, padding=1)
    assert (y.H, y.W) == tuple(want.shape[-2:])
    got = y.t.float().view(B, y.H, y.W, cout).permute(0, 3, 1, 2)
    close(got, want, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize('tile,stride,up', [(t, s, u) for t in (0, 2, 9, 13, 16, 20, 30, 32, 33) for s, u in ((1, False), (2, False), (1, True))
                                            if not (u and t >= 30)])     # (the ping-pong tiles have no fused nearest upsample)
def test_conv_input_with_a_pixel_stride(dev, tile, stride, up):
    '''The input of a convolution may be a column slice of a wider NHWC matrix (fd_gemm_desc.lda = pixel stride: a skip tensor
    living in the right-hand columns of its concat buffer feeds the UNet's downsample convolution).  Same kernels, same
    arithmetic, other addresses: bit-identical to the contiguous input, for every loader (register-staged, LDS-DMA one-tile /
    persistent, ping-pong), stride 1 / 2 and the fused nearest upsample; the neighbouring columns hold NaN.'''
    from flexdiffuse_amd import ops
    B, H, W, cin, cout, left = 2, 32, 32, 320, 320, 192
    x, w, b = rnd((B * H * W, cin), 1), rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5), rnd((cout,), 3)
    cw = ops.prep_conv(w, b, dev)
    wide = torch.full((B * H * W, left + cin + 64), float('nan'), dtype=torch.float16, device=dev)
    wide[:, left:left + cin] = x.half().to(dev)
    xc = ops.Act(x.half().to(dev), B, H, W)
    xs = ops.Act(wide[:, left:left + cin], B, H, W)
    assert not xs.t.is_contiguous()
    old = ops.FORCE_TILE
    ops.FORCE_TILE = tile
    try:
        want = ops.conv2d(xc, cw, stride=stride, up=up).t.clone()
        got = ops.conv2d(xs, cw, stride=stride, up=up).t
    finally:
        ops.FORCE_TILE = old
    assert torch.isfinite(got.float()).all()
    assert torch.equal(got, want)


def test_conv_pixel_stride_on_the_register_staged_loader(dev):
    '''...and the 4-wave register-staged kernel (tensors >= 2 GiB; FD_GEMM_NO_DMA=1 selects it for everything) in a child process:
    strided input == contiguous input bit for bit, and both agree with torch.'''
    import os
    import subprocess
    import sys
    code = '''
import torch, torch.nn.functional as F
from flexdiffuse_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
B, H, W, cin, cout, left = 2, 16, 16, 128, 64, 64
x = torch.randn((B * H * W, cin), generator=g)
w = torch.randn((cout, cin, 3, 3), generator=g) * (9 * cin) ** -0.5
b = torch.randn((cout,), generator=g)
cw = ops.prep_conv(w, b, dev)
wide = torch.full((B * H * W, left + cin + 8), float('nan'), dtype=torch.float16, device=dev)
wide[:, left:left + cin] = x.half().to(dev)
for stride in (1, 2):
    want = ops.conv2d(ops.Act(x.half().to(dev), B, H, W), cw, stride=stride)
    got = ops.conv2d(ops.Act(wide[:, left:left + cin], B, H, W), cw, stride=stride)
    assert torch.equal(got.t, want.t)
    ref = F.conv2d(x.half().float().view(B, H, W, cin).permute(0, 3, 1, 2), w.half().float(), b, stride=stride, padding=1)
    err = (got.t.float().view(B, got.H, got.W, cout).permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
    assert err < 2e-2, err
print('ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FD_GEMM_NO_DMA='1', PYTHONPATH=root)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('tile', [9, 12, 13, 14, 15, 16, 20, 23])
def test_lds_dma_tiles_race_screen_with_weights_streaming_from_hbm(dev, tile):
    '''The race screen of tests/test_gpu_gemm_pp.py for the other LDS-DMA kernels (one-tile 2-stage, 3-stage with counted vmcnt, persistent):
    a 52 MB weight matrix of which every tile reads its own rows once (every W DMA is an HBM miss), short K loop; the result must agree
    with torch and repeat bit for bit.'''
    from flexdiffuse_amd import ops
    M, N, K = 576, 81920, 320
    g = torch.Generator().manual_seed(19)
    a = torch.randn((M, K), generator=g).half().to(dev)
    w = (torch.randn((N, K), generator=g) * K ** -0.5).half()
    lw = ops.prep_linear(w.float(), torch.zeros(N), dev)
    want = a.float() @ w.to(dev).float().t()
    old = ops.FORCE_TILE
    ops.FORCE_TILE = tile
    try:
        ref = ops.gemm(a, lw).clone()
        torch.cuda.synchronize()
        err = float((ref.float() - want).abs().max())
        assert err < 2e-2 * max(1.0, float(want.abs().max())), err
        out = torch.empty_like(ref)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(40):
            ops.gemm(a, lw, out=out)
            bad += (out != ref).sum()
        torch.cuda.synchronize()
        assert int(bad) == 0, int(bad)
    finally:
        ops.FORCE_TILE = old


def test_conv_small_cin_and_asym_pad(dev):
    from flexdiffuse_amd import ops
    B, H, W = 2, 16, 16
    x, w, b = rnd((B, 4, H, W), 1), rnd((320, 4, 3, 3), 2, 1 / 6), rnd((320,), 3)
    cw = ops.prep_conv(w, b, dev)
    assert cw.im2col
    y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), cw)
    close(y.t.float().view(B, H, W, 320).permute(0, 3, 1, 2), F.conv2d(x, w, b, padding=1))
    # VAE-encoder style: pad (0,1,0,1) then stride-2 conv without padding
    x, w, b = rnd((B, 128, H, W), 4), rnd((128, 128, 3, 3), 5, (9 * 128) ** -0.5), rnd((128,), 6)
    cw = ops.prep_conv(w, b, dev)
    y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), cw, stride=2, pad=(0, 0), out_hw=(H // 2, W // 2))
    want = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    close(y.t.float().view(B, H // 2, W // 2, 128).permute(0, 3, 1, 2), want, rtol=3e-3, atol=3e-3)
    # conv_out style: N = 4, fp32 output
    x, w, b = rnd((B, 320, H, W), 7), rnd((4, 320, 3, 3), 8, (9 * 320) ** -0.5), rnd((4,), 9)
    cw = ops.prep_conv(w, b, dev)
    y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), cw, out_f32=True)
    got = ops.nhwc_to_nchw(y.t, B, 4, H, W)
    close(got, F.conv2d(x, w, b, padding=1), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('C,HW,silu,eps', [(320, 1024, True, 1e-5), (960, 256, True, 1e-5),
                                           (1280, 64, False, 1e-6), (2560, 64, True, 1e-5),
                                           (128, 4096, True, 1e-6), (64, 256, False, 1e-6),
                                           (640, 1024, True, 1e-5), (1920, 1024, True, 1e-5),
                                           (640, 4096, False, 1e-5), (96, 300, True, 1e-5)])
def test_groupnorm(dev, C, HW, silu, eps):
    from flexdiffuse_amd import ops
    B = 3
    x = (rnd((B, C, HW), 1) * 1.5 + 0.7).half().float()
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    xa = ops.Act(x.permute(0, 2, 1).reshape(B * HW, C).half().to(dev).contiguous(), B, HW, 1)
    y = ops.groupnorm(xa, g.to(dev), b.to(dev), 32, eps, silu)
    want = F.group_norm(x, 32, g, b, eps)
    if silu:
        want = F.silu(want)
    close(y.t.float().view(B, HW, C).permute(0, 2, 1), want, rtol=3e-3, atol=3e-3)


@pytest.mark.parametrize('B,HW,C,silu', [(16, 4096, 320, True), (16, 4096, 640, True), (8, 4096, 320, False),
                                        (2, 4096, 320, True), (16, 3600, 320, True), (5, 4100, 640, False),
                                        (16, 4096, 960, True), (1, 16384, 128, True)])
def test_groupnorm_large_maps(dev, B, HW, C, silu):
    '''Large maps (the streaming statistics + apply pair, partial sums through the workspace): against
    torch fp32, then 60 repeats while another stream keeps the chip unevenly busy -- every repeat must
    reproduce the first bit for bit (fixed-order reductions, no atomics).'''
    from flexdiffuse_amd import ops
    gen = torch.Generator().manual_seed(B * 7 + C)
    x = (torch.randn((B, HW, C), generator=gen) * 1.5 + 0.7 + torch.randn((B, 1, C), generator=gen)).half()
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    xa = ops.Act(x.reshape(B * HW, C).to(dev).contiguous(), B, HW, 1)
    y = ops.groupnorm(xa, g.to(dev), b.to(dev), 32, 1e-5, silu)
    want = F.group_norm(x.float().permute(0, 2, 1), 32, g, b, 1e-5)
    if silu:
        want = F.silu(want)
    close(y.t.float().view(B, HW, C).permute(0, 2, 1), want, rtol=3e-3, atol=3e-3)
    first = y.t.clone()
    side = torch.cuda.Stream()
    a = torch.randn((4096, 1280), device=dev).half()
    w = ops.prep_linear(torch.randn((1280, 1280)) * 1280 ** -0.5, None, dev)
    for it in range(60):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                ops.WS_SLOT = 1
                for _ in range(2 + it % 5):
                    ops.gemm(a, w)
                ops.WS_SLOT = 0
        y = ops.groupnorm(xa, g.to(dev), b.to(dev), 32, 1e-5, silu)
        assert torch.equal(y.t, first), it
    torch.cuda.synchronize()


@pytest.mark.parametrize('C', [128, 320, 768, 1280])
def test_layernorm(dev, C):
    from flexdiffuse_amd import ops
    x = (rnd((77, C), 1) * 2 + 0.3).half().float()
    g, b = 1 + 0.1 * rnd((C,), 2), 0.1 * rnd((C,), 3)
    y = ops.layernorm(x.half().to(dev), g.to(dev), b.to(dev), 1e-5, out_f32=True)
    close(y, F.layer_norm(x, (C,), g, b, 1e-5), rtol=1e-4, atol=1e-4)
    y = ops.layernorm(x.half().to(dev), g.to(dev), b.to(dev), 1e-5)
    close(y, F.layer_norm(x, (C,), g, b, 1e-5))


def attn_ref(q, k, v, heads, causal=False):
    B, Nq, C = q.shape
    d = C // heads
    qh, kh, vh = (t.view(B, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * d ** -0.5
    if causal:
        s = s + torch.full((Nq, k.shape[1]), float('-inf')).triu(1)
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize('Nq,Nk,heads,d,causal', [(256, 256, 8, 40, False), (1024, 1024, 2, 40, False),
                                                  (256, 77, 8, 40, False), (64, 64, 8, 160, False),
                                                  (256, 256, 8, 80, False), (100, 77, 8, 160, False),
                                                  (77, 77, 12, 64, True), (257, 257, 4, 64, False),
                                                  (2048, 1024, 2, 40, False), (2200, 1100, 1, 48, False),
                                                  (2048, 2048, 1, 40, True),
                                                  # long query rows x short text context: two query blocks per wave
                                                  (4096, 77, 8, 40, False), (2100, 64, 2, 48, False), (2048, 200, 2, 40, False),
                                                  # the exact level-0 self-attention shape of the SD1.5 UNet (64x64 latents, 8 heads x 40)
                                                  (4096, 4096, 8, 40, False)])
def test_attention(dev, Nq, Nk, heads, d, causal):
    from flexdiffuse_amd import ops
    B, C = 2, heads * d
    q, k, v = rnd((B, Nq, C), 1), rnd((B, Nk, C), 2), rnd((B, Nk, C), 3)
    q[0, 3] *= 6.0      # a spiky row: forces large running-max updates across key tiles
    k[0, Nk - 2] *= 6.0
    q, k = q.half().float(), k.half().float()     # keep the inputs fp16-exact
    ld = (Nk + 7) // 8 * 8
    vt = torch.zeros((B, C, ld), dtype=torch.float16)
    vt[:, :, :Nk] = v.transpose(1, 2).half()
    out = ops.attention(q.half().to(dev).view(B * Nq, C), k.half().to(dev).view(B * Nk, C),
                        vt.to(dev), B, heads, Nq, Nk, d, causal)
    close(out.view(B, Nq, C), attn_ref(q, k, v, heads, causal), rtol=4e-3, atol=4e-3)


@pytest.mark.parametrize('Nq,Nk,heads,d,causal', [(256, 256, 8, 40, False), (1024, 1024, 2, 40, False),
                                                  (256, 77, 8, 40, False), (64, 64, 8, 160, False),
                                                  (256, 250, 8, 80, False), (100, 77, 8, 160, False),
                                                  (77, 77, 12, 64, True), (130, 200, 2, 48, False),
                                                  (192, 320, 2, 32, False),
                                                  # >= 2048 queries x >= 1024 keys: two query blocks per wave
                                                  (2048, 1024, 2, 40, False), (4096, 4096, 1, 40, False),
                                                  (2100, 1030, 2, 48, False), (2048, 1024, 1, 32, True),
                                                  (4096, 77, 8, 40, False), (2100, 64, 2, 48, False), (2304, 200, 2, 40, False),
                                                  # head_dim 64 with two query blocks per wave (k_attention_w8q2<64,4>: SD2.1 at 768x768)
                                                  (2304, 2304, 2, 64, False), (2100, 1030, 2, 56, False), (2048, 2048, 1, 64, True),
                                                  # head_dim 40 on the 32x32x16 QK^T form (k_attention_w8q2m): causal, ragged queries and keys
                                                  (2048, 2048, 1, 40, True), (2091, 1093, 3, 40, False), (2500, 2500, 1, 40, True),
                                                  # the exact level-0 self-attention launch (k_attention_w8q2m: 12 % of the pass)
                                                  (4096, 4096, 8, 40, False)])
def test_attention_prescaled_q(dev, Nq, Nk, heads, d, causal):
    '''q_prescaled: Q carries head_dim^-0.5 * log2(e); the kernel feeds the running max into
    the QK^T MFMA accumulator and (head_dim <= 40) takes the denominator from the PV MFMA.
    The max is advanced lazily, so keys are ordered to make later tiles exceed it by far.'''
    from flexdiffuse_amd import ops
    assert ops.attention_accepts_prescaled(d)
    B, C = 2, heads * d
    q, k, v = rnd((B, Nq, C), 11), rnd((B, Nk, C), 12), rnd((B, Nk, C), 13)
    q[0, 3] *= 6.0
    k[0, Nk - 2] *= 6.0
    k[1, Nk // 2:] *= 3.0    # second half of the keys scores much higher than the first
    qs = (q * (d ** -0.5 * ops.QK_LOG2E)).half()      # what the scaled q projection would emit
    q = qs.float() / (d ** -0.5 * ops.QK_LOG2E)       # the reference sees the same rounded q
    k = k.half().float()
    ld = (Nk + 7) // 8 * 8 + 8
    vt = torch.full((B, C, ld), 7.0, dtype=torch.float16)   # finite junk beyond n_k rounded to 8
    vt[:, :, :(Nk + 7) // 8 * 8] = 0
    vt[:, :, :Nk] = v.transpose(1, 2).half()
    out = ops.attention(qs.to(dev).view(B * Nq, C), k.half().to(dev).view(B * Nk, C),
                        vt.to(dev), B, heads, Nq, Nk, d, causal, q_prescaled=True)
    close(out.view(B, Nq, C), attn_ref(q, k, v, heads, causal), rtol=4e-3, atol=4e-3)


def test_gemm_transposed_store_feeds_attention(dev):
    '''V projection written as V^T by the GEMM epilogue, consumed by the attention kernel.'''
    from flexdiffuse_amd import ops
    B, N, Cin, heads, d = 2, 77, 768, 8, 40
    C = heads * d
    ctx, wv = rnd((B * N, Cin), 1), rnd((C, Cin), 2, Cin ** -0.5)
    vt = ops.gemm_vt(ctx.half().to(dev), ops.prep_linear(wv, None, dev), B, N, 80)
    want = (ctx @ wv.T).view(B, N, C).transpose(1, 2)
    close(vt[:, :, :N], want)
    assert float(vt[:, :, N:].abs().max()) == 0.0


@pytest.mark.parametrize('B,N,C', [(2, 4100, 320), (4, 4096, 640), (16, 1024, 640)])
def test_gemm_transposed_store_wide_tile(dev, B, N, C):
    '''M = B*N >= 8192 with 160 | C takes the 128x160 / 8-wave transposed tile (ragged last row
    block, padded columns stay zero); FD_GEMM_VT_TILE=0 is the 128x64 tile of the small case above.'''
    from flexdiffuse_amd import ops
    x, wv, bv = rnd((B * N, C), 3), rnd((C, C), 4, C ** -0.5), rnd((C,), 5, 0.1)
    ldv = (N + 7) // 8 * 8 + 8
    vt = ops.gemm_vt(x.half().to(dev), ops.prep_linear(wv, bv, dev), B, N, ldv)
    want = (x.half().float() @ wv.half().float().T + bv).view(B, N, C).transpose(1, 2)
    close(vt[:, :, :N], want, rtol=3e-3, atol=3e-3)
    assert float(vt[:, :, N:].abs().max()) == 0.0


def test_vae_attention_pieces(dev):
    from flexdiffuse_amd import ops
    B, N, C = 2, 256, 512
    q, k = rnd((B, N, C), 1), rnd((B, N, C), 2)
    s = ops.bgemm(q.half().to(dev), k.half().to(dev), alpha=C ** -0.5)
    close(s, (q @ k.transpose(1, 2)) * C ** -0.5, rtol=3e-3, atol=3e-3)
    p = ops.softmax_rows_(s.clone())
    close(p, ((q @ k.transpose(1, 2)) * C ** -0.5).softmax(-1), rtol=5e-3, atol=2e-4)


def test_cfg_ddim_and_layout(dev):
    from flexdiffuse_amd import ops
    from oracle import ddim_ref as D
    B, C, H, W = 3, 4, 8, 8
    x, eps = rnd((B, C, H, W), 1), rnd((2 * B, C, H, W), 2)
    acp = D.alphas_cumprod()
    t, g = 500, 8.0
    e = eps[:B] + g * (eps[B:] - eps[:B])
    want = D.ddim_step(e, t, x, acp, 50)
    a_t, a_p = acp[t], acp[t - 20]
    coef = (float((1 - a_t).sqrt()), float(a_t.sqrt()), float(a_p.sqrt()), float((1 - a_p).sqrt()))
    xd = x.to(dev).clone()
    eps_nhwc = eps.permute(0, 2, 3, 1).reshape(2 * B * H * W, C).contiguous().to(dev)
    ops.cfg_ddim_step(xd, eps_nhwc, B, C, H * W, True, g, coef)
    close(xd, want, rtol=1e-6, atol=1e-6)
    a = ops.nchw_to_nhwc(x.to(dev), rep=2, c_pad=8)
    assert a.t.shape == (2 * B * H * W, 8)
    want_nhwc = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
    close(a.t[:B * H * W, :C], want_nhwc, rtol=1e-3, atol=1e-3)
    close(a.t[B * H * W:, :C], want_nhwc, rtol=1e-3, atol=1e-3)
    assert float(a.t[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize('tile,split', [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (1, 2), (2, 4),
                                        (5, 2), (6, 4), (3, 8), (7, 1), (8, 1), (7, 2), (9, 1), (10, 2), (12, 1), (13, 1), (13, 4),
                                        (14, 1), (15, 1), (15, 2), (16, 1), (16, 2), (20, 1), (23, 1), (23, 2)])
def test_gemm_every_tile_and_split(dev, tile, split):
    """Each block tile (incl. the 8-wave 256-row ones) and split-K factor gives the same
    result as torch on a conv and on a ragged linear problem."""
    from flexdiffuse_amd import ops
    try:
        ops.FORCE_TILE, ops.FORCE_SPLIT = tile, split
        B, cin, cout, H = 2, 320, 320, 24
        x, w, b = rnd((B, cin, H, H), 1), rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5), rnd((cout,), 3)
        res = rnd((B * H * H, cout), 4)
        y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), ops.prep_conv(w, b, dev),
                       residual=res.half().to(dev))
        want = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(B * H * H, cout) + res
        close(y.t, want, rtol=3e-3, atol=3e-3)
        M, N, K = 1000, 328, 1288
        a, wl, bl = rnd((M, K), 5), rnd((N, K), 6, K ** -0.5), rnd((N,), 7)
        out = ops.gemm(a.half().to(dev), ops.prep_linear(wl, bl, dev), act=ops.ACT_SILU)
        close(out[:, :N], F.silu(a @ wl.T + bl), rtol=3e-3, atol=3e-3)
    finally:
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0


def test_gemm_bias_tiles_staged_in_lds(dev):
    """The DMA kernels stage each tile's bias (and the per-sample bias of single-sample tiles)
    in LDS.  Tile widths that are not a multiple of 64 (160) must not let the last 64-column
    slice spill into the neighbouring buffer: distinct bias values per column, several n-tiles,
    a per-sample bias, persistent and one-shot launches, bit-identical reruns."""
    from flexdiffuse_amd import ops
    B, cin, cout, H = 3, 64, 480, 16                      # 3 n-tiles of 160; tiles lie inside one sample
    x, w = rnd((B, cin, H, H), 1), rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5)
    b = torch.arange(cout, dtype=torch.float32) * 0.25 - 40.0
    b2 = rnd((B, cout), 3) * 8.0
    xa, wp = ops.nchw_to_nhwc(x.to(dev)), ops.prep_conv(w, b, dev)
    b2d = b2.to(dev).contiguous()
    want = (F.conv2d(x, w, b, padding=1) + b2[:, :, None, None]).permute(0, 2, 3, 1).reshape(B * H * H, cout)
    outs = [ops.conv2d(xa, wp, bias2=b2d, ld_bias2=cout).t.float().cpu() for _ in range(4)]
    close(outs[0], want, rtol=3e-3, atol=2e-2)
    assert all(torch.equal(outs[0], o) for o in outs)
    # short-K linear with many tiles (persistent kernel: the next tile's bias arrives while the
    # current epilogue reads its own)
    M, N, K = 8192, 800, 320                               # 5 n-tiles of 160
    a, wl = rnd((M, K), 5), rnd((N, K), 6, K ** -0.5)
    bl = torch.arange(N, dtype=torch.float32) * 0.125 - 50.0
    wlp = ops.prep_linear(wl, bl, dev)
    ad = a.half().to(dev)
    outs = [ops.gemm(ad, wlp).float().cpu() for _ in range(4)]
    close(outs[0][:, :N], a.half().float() @ wl.half().float().T + bl, rtol=3e-3, atol=3e-2)
    assert all(torch.equal(outs[0], o) for o in outs)


def _ln_ref(x, gamma, beta, w, b, eps=1e-5):
    n = torch.nn.functional.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps)
    return n @ w.float().t() + (b.float() if b is not None else 0.0)


@pytest.mark.parametrize('M,N,K', [(65536, 320, 320), (2048, 640, 320), (16384, 640, 640), (4096, 1280, 1280),
                                   (1024, 1280, 1280), (1000, 328, 320), (200, 320, 640)])
def test_gemm_layernorm_fold(dev, M, N, K):
    '''fd_gemm_desc.ln_stats: LayerNorm folded into the GEMM (gain folded into the weights, per-row
    (rstd, -mean rstd) applied in the epilogue) == LayerNorm followed by the linear layer.  Rows get
    a non-zero mean and spread so that the mean term matters; shapes cover the lean persistent /
    non-persistent kernels (EPI 5), the 3-stage tile, the 64x64 tile and ragged generic tiles.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn((M, K), generator=g) * (0.5 + torch.rand((M, 1), generator=g) * 2) +
         torch.randn((M, 1), generator=g) * 1.5).half()
    gamma = 1.0 + 0.3 * torch.randn(K, generator=g)
    beta = 0.2 * torch.randn(K, generator=g)
    w = torch.randn((N, K), generator=g) * K ** -0.5
    b = torch.randn(N, generator=g) * 0.1
    lw = ops.prep_linear_ln(w, b, gamma, beta, dev)
    xd = x.to(dev)
    st = ops.ln_row_stats(xd)
    mean, var = x.float().mean(1), x.float().var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0].cpu(), rstd, rtol=1e-5, atol=1e-6)
    assert torch.allclose(st[:, 1].cpu(), -mean * rstd, rtol=1e-5, atol=1e-5)
    got = ops.gemm(xd, lw, ln_stats=st)[:, :N].float().cpu()
    want = _ln_ref(x, gamma, beta, w, b)
    assert float((got - want).abs().max()) < 2e-2 * float(want.abs().max()), (M, N, K)
    # and against the unfused device path (LayerNorm kernel + plain GEMM): same tolerance class
    n16 = ops.layernorm(xd, gamma.to(dev), beta.to(dev))
    plain = ops.gemm(n16, ops.prep_linear(w, b, dev))[:, :N].float().cpu()
    assert float((got - plain).abs().max()) < 2e-2 * float(want.abs().max())
    assert float((got - want).abs().mean()) <= 1.5 * float((plain - want).abs().mean()) + 1e-4


@pytest.mark.parametrize('M,C', [(16384, 320), (4096, 320), (1024, 640), (264, 320)])
def test_gemm_layernorm_fold_geglu_and_transposed(dev, M, C):
    '''The fold through the GEGLU epilogue (EPI 6 and the generic path) and through the transposed
    (V^T) store.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn((M, C), generator=g) * 1.3 + torch.randn((M, 1), generator=g)).half()
    gamma = 1.0 + 0.3 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    w = torch.randn((8 * C, C), generator=g) * C ** -0.5
    b = torch.randn(8 * C, generator=g) * 0.1
    xd = x.to(dev)
    st = ops.ln_row_stats(xd)
    lw = ops.prep_linear_ln(w, b, gamma, beta, dev, geglu=True)
    got = ops.gemm(xd, lw, act=ops.ACT_GEGLU, ln_stats=st).float().cpu()
    h = _ln_ref(x, gamma, beta, w, b)
    want = h[:, :4 * C] * torch.nn.functional.gelu(h[:, 4 * C:])
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 2e-2 * float(want.abs().max()), (M, C)
    # transposed store: V^T [B][C][HW] of LN(x) Wv^T
    B = 4 if M % 4 == 0 and (M // 4) % 8 == 0 else 1
    HW = M // B
    if HW % 8 == 0:
        wv = torch.randn((C, C), generator=g) * C ** -0.5
        lv = ops.prep_linear_ln(wv, None, gamma, beta, dev)
        vt = ops.gemm_vt(xd, lv, B, HW, HW, ln_stats=st).float().cpu()
        want_v = _ln_ref(x, gamma, beta, wv, None).view(B, HW, C).permute(0, 2, 1)
        assert float((vt - want_v).abs().max()) < 2e-2 * float(want_v.abs().max())


@pytest.mark.parametrize('M,res', [(65536, True), (2048, False), (256, True)])
def test_gemm_emits_row_stats_of_its_output(dev, M, res):
    '''fd_gemm_desc.ln_stats_out: the 256x320 tile spans the whole row (N = 320) and writes
    (rstd, -mean rstd) of the fp16 rows it stores -- the LayerNorm statistics its consumer GEMM folds
    in -- identical in meaning to fd_ln_row_stats_f16 on the stored output.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M)
    N = K = 320
    a = torch.randn((M, K), generator=g).half().to(dev)
    w = ops.prep_linear(torch.randn((N, K), generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.5, dev)
    r = (torch.randn((M, N), generator=g) * 2 + torch.randn((M, 1), generator=g) * 3).half().to(dev) if res else None
    st = torch.empty((M, 2), dtype=torch.float32, device=dev)
    out = ops.gemm(a, w, residual=r, ln_stats_out=st)
    plain = ops.gemm(a, w, residual=r)
    # same values as the launch without statistics (which may pick another tile: last-bit differences only)
    assert float((out.float() - plain.float()).abs().max()) <= 2e-3 * float(plain.float().abs().max())
    ref = ops.ln_row_stats(out)                          # exact two-pass statistics of the stored rows
    x = out.float()
    assert torch.allclose(ref[:, 0], (x.var(1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    # one-pass E[x^2] - mean^2 in fp32 vs the exact two-pass: relative 1e-4 on rstd, absolute 1e-4 on -mean rstd
    assert float(((st[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < 2e-4
    assert float((st[:, 1] - ref[:, 1]).abs().max()) < 2e-4 * max(1.0, float(ref[:, 1].abs().max()))
    assert ops.can_emit_row_stats(M, 640) == 4 and not ops.can_emit_row_stats(M, 328) and not ops.can_emit_row_stats(M + 8, 640)
    with pytest.raises(ValueError):
        ops.gemm(a[:, :K], ops.prep_linear(torch.zeros((328, K)), None, dev), ln_stats_out=st)


@pytest.mark.parametrize('B,HW,rep,L,d', [(2, 4096, 1, 77, 40), (1, 1024, 2, 77, 40), (3, 256, 1, 65, 40), (2, 512, 2, 80, 40),
                                            (16, 1024, 1, 77, 80), (2, 128, 2, 66, 80), (3, 2304, 1, 77, 80)])
def test_fused_q_projection_cross_attention(dev, B, HW, rep, L, d):
    '''fd_xattn_q_f16: LayerNorm-fold q projection + softmax(Q K^T) V over a packed 65..80-key context in
    ONE launch (8 heads x 40: 256-row tiles, a head pair per wave; 8 x 80: 128-row tiles, two n-tiles, a head per wave) vs (a) the two launches it replaces -- fd_gemm_f16 with ln_stats, then
    fd_attention_f16 with q_prescaled -- and (b) a torch fp32 reference of LayerNorm -> to_q ->
    attention.  Q is rounded to fp16 identically in both device paths; the attention proper differs in
    MFMA summation order (channel / key permutations inside the fragments) and in the softmax
    reference point (exact row max here, first-tile lazy max there), so (a) is a few fp16 ulps, not
    bit-level.  `rep` > 1: context replicas sharing the queries (the CFG fan-out of the shared prefix).'''
    from flexdiffuse_amd import ops
    heads = 8
    C = heads * d
    M = B * HW
    g = torch.Generator().manual_seed(B * HW + rep + L)
    x = (torch.randn((M, C), generator=g) * (0.5 + torch.rand((M, 1), generator=g) * 2) +
         torch.randn((M, 1), generator=g) * 1.5).half()
    gamma = 1.0 + 0.3 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    wq = torch.randn((C, C), generator=g) * C ** -0.5 * 1.5
    qs = ops.QK_LOG2E * d ** -0.5
    lw = ops.prep_linear_ln(wq * qs, None, gamma, beta, dev)
    k = (torch.randn((rep * B * L, C), generator=g) * 1.2).half()
    v = torch.randn((rep * B, L, C), generator=g).half()
    ldv = (L + 7) // 8 * 8
    vt = torch.zeros((rep * B, C, ldv), dtype=torch.float16)
    vt[:, :, :L] = v.permute(0, 2, 1)
    xd, kd, vtd = x.to(dev), k.to(dev), vt.to(dev)
    st = ops.ln_row_stats(xd)
    assert ops.xattn_supported(heads, d, L, HW)
    img = ops.xattn_pack_kv(kd, vtd, rep * B, L, heads, d)
    got = ops.xattn_q(xd, lw, st, img, HW, L, heads, d, n_rep=rep).float().cpu()
    assert got.shape == (rep * M, C) and bool(torch.isfinite(got).all())
    # (a) the unfused device path
    q = ops.gemm(xd, lw, ln_stats=st)
    for r in range(rep):
        o = ops.attention(q, kd[r * B * L:(r + 1) * B * L], vtd[r * B:(r + 1) * B], B, heads, HW, L, d,
                          q_prescaled=True).float().cpu()
        err = (got[r * M:(r + 1) * M] - o).abs()
        assert float(err.max()) <= 4e-3 * float(o.abs().max()) + 1e-3, (r, float(err.max()), float(o.abs().max()))
    # (b) torch fp32: LayerNorm -> to_q -> softmax(q k^T / sqrt d) v per head
    xn = torch.nn.functional.layer_norm(x.float(), (C,), gamma, beta, 1e-5)
    qf = (xn @ wq.T).view(B, HW, heads, d).permute(0, 2, 1, 3)
    for r in range(rep):
        kf = k[r * B * L:(r + 1) * B * L].float().view(B, L, heads, d).permute(0, 2, 1, 3)
        vf = v[r * B:(r + 1) * B].float().view(B, L, heads, d).permute(0, 2, 1, 3)
        p = torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, dim=-1)
        want = (p @ vf).permute(0, 2, 1, 3).reshape(M, C)
        e = float((got[r * M:(r + 1) * M] - want).abs().max())
        assert e <= 2e-2 * float(want.abs().max()), (r, e, float(want.abs().max()))
    if rep == 2:   # the replicas really saw different contexts
        assert float((got[:M] - got[M:]).abs().max()) > 0.05
    assert not ops.xattn_supported(8, 64, L, HW) and not ops.xattn_supported(heads, d, 64, HW)
    # the producer's PARTIAL sums instead of finished statistics (fd_xattn_desc.ln_stats_parts): every tile finalises its own rows -- the bits of
    # fd_ln_finalize_stats_f32 + the launch above
    for kparts in (2, 4):
        xs = xd.float().view(M, kparts, C // kparts)
        parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).permute(1, 0, 2).contiguous()      # [k][M][2]
        fin = ops.ln_finalize_stats(parts, C)
        a = ops.xattn_q(xd, lw, fin, img, HW, L, heads, d, n_rep=rep)
        b = ops.xattn_q(xd, lw, parts, img, HW, L, heads, d, n_rep=rep)
        assert torch.equal(a, b) and bool(torch.isfinite(b.float()).all())
        assert float((b.float().cpu() - got).abs().max()) <= 2e-2 * float(got.abs().max())
    assert not ops.xattn_supported(heads, d, L, 64) and not ops.xattn_supported(5, 64, L, HW)


@pytest.mark.parametrize('B,H,Cin,Cx', [(16, 64, 320, 640), (4, 32, 640, 960), (16, 16, 1280, 2560), (16, 8, 1280, 2560),
                                        (2, 16, 64, 128)])
def test_conv_with_appended_shortcut(dev, B, H, Cin, Cx):
    '''fd_gemm_desc.A2 / K2: a ResBlock's 1x1 shortcut accumulated by conv2's own K loop,
    conv3x3(h) + x Ws^T + (b + bs), vs (a) a torch fp32 reference and (b) the two launches it replaces (shortcut GEMM,
    then the conv with that tensor as residual: there the shortcut is rounded to fp16 once more, so (b) is a few
    ulps).  Shapes: the 256x320 tap-fastest tile (level 0), 256x160 (level 1), split-K 2/4 (16x16) and 8 (8x8), a
    small generic one.'''
    from flexdiffuse_amd import ops
    Cout = Cin
    g = torch.Generator().manual_seed(B * H + Cx)
    hx = (torch.randn((B, Cin, H, H), generator=g) * 0.7).half()
    x = (torch.randn((B, Cx, H, H), generator=g) * 0.7).half()
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ws = torch.randn((Cout, Cx), generator=g) * Cx ** -0.5
    bs = torch.randn(Cout, generator=g) * 0.1
    M = B * H * H
    hd = ops.Act(hx.permute(0, 2, 3, 1).reshape(M, Cin).contiguous().to(dev), B, H, H)
    # x as a column slice of a wider buffer (the decoder's concat buffers): row stride != Cx
    xbuf = torch.zeros((M, Cx + 64), dtype=torch.float16, device=dev)
    xbuf[:, :Cx] = x.permute(0, 2, 3, 1).reshape(M, Cx).to(dev)
    xd = xbuf[:, :Cx]
    fused = ops.conv2d(hd, ops.prep_conv_shortcut(w, b, ws, bs, dev), a2=xd).t.float().cpu()
    want = F.conv2d(hx.float(), w.half().float(), b + bs, padding=1) + \
        torch.einsum('bchw,oc->bohw', x.float(), ws.half().float())
    want = want.permute(0, 2, 3, 1).reshape(M, Cout)
    assert float((fused - want).abs().max()) <= 4e-3 * float(want.abs().max()) + 2e-3
    sc = ops.gemm(xd, ops.prep_linear(ws, bs, dev))
    two = ops.conv2d(hd, ops.prep_conv(w, b, dev), residual=sc).t.float().cpu()
    assert float((fused - two).abs().max()) <= 4e-3 * float(want.abs().max()) + 2e-3
    assert float((fused - want).abs().mean()) <= float((two - want).abs().mean()) * 1.05 + 1e-5   # one rounding fewer
    with pytest.raises(AssertionError):
        ops.conv2d(hd, ops.prep_conv_shortcut(w, b, ws, bs, dev))          # a2 missing


@pytest.mark.parametrize('M,C', [(65536, 320), (16384, 640), (4096, 1280), (1024, 1280), (256, 64)])
def test_gemm_with_appended_operand_folds_two_linears(dev, M, C):
    '''fd_gemm_desc.A2 / K2 on a linear GEMM: out = f Wc^T + h Wp^T + b + residual in ONE K loop, the form the UNet
    uses to fold proj_out through the feed-forward output layer (Wc = Wp W2): vs a torch fp32 reference of
    proj_out(ff2(f) + h) + x and vs the two device launches it replaces.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + C)
    f = (torch.randn((M, 4 * C), generator=g) * 0.5).half()
    h = torch.randn((M, C), generator=g).half()
    xr = torch.randn((M, C), generator=g).half()
    w2, b2 = torch.randn((C, 4 * C), generator=g) * (4 * C) ** -0.5, torch.randn(C, generator=g) * 0.1
    wp, bp = torch.randn((C, C), generator=g) * C ** -0.5, torch.randn(C, generator=g) * 0.1
    want = ((f.float() @ w2.T + b2 + h.float()) @ wp.T + bp + xr.float())
    lw = ops.prep_linear(torch.cat([wp @ w2, wp], 1), bp + wp @ b2, dev)
    fd, hd, xd = f.to(dev), h.to(dev), xr.to(dev)
    got = ops.gemm(fd, lw, a2=hd, residual=xd).float().cpu()
    assert float((got - want).abs().max()) <= 6e-3 * float(want.abs().max()) + 2e-3
    h2 = ops.gemm(fd, ops.prep_linear(w2, b2, dev), residual=hd)
    two = ops.gemm(h2, ops.prep_linear(wp, bp, dev), residual=xd).float().cpu()
    assert float((got - two).abs().max()) <= 8e-3 * float(want.abs().max()) + 2e-3
    assert float((got - want).abs().mean()) <= float((two - want).abs().mean()) * 1.1 + 1e-5
    with pytest.raises(AssertionError):
        ops.gemm(fd, lw, residual=xd)      # K mismatch without the second operand


@pytest.mark.parametrize('B,H,Cin,Cout', [(16, 32, 640, 640), (16, 16, 1280, 1280), (2, 64, 512, 512), (1, 128, 256, 256),
                                          (4, 32, 128, 128), (16, 8, 1280, 1280), (16, 12, 1280, 1280)])   # (the last two: 1024 / 2304 rows, round 5)
def test_upsample_conv_phase_decomposition(dev, B, H, Cin, Cout):
    '''fd_gemm_desc.upsample2x == 2: nearest-2x upsample + conv3x3 (diffusers Upsample2D) as four 2x2 parity convolutions
    of the low-resolution input in one launch (4/9 of the MACs) vs (a) torch fp32 F.interpolate + conv2d and (b) the
    fused-upsample implicit GEMM it replaces.  The parity filters are the 3x3 taps summed in fp32 and rounded once, so
    (b) differs by weight rounding only.  Shapes: the UNet's two large upsamplers, VAE-like widths, into a strided
    output (the decoder's concat buffer).'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(B + H + Cin)
    x = (torch.randn((B, Cin, H, H), generator=g) * 0.7).half()
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, generator=g) * 0.1
    want = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode='nearest'), w.float(), b, padding=1)
    want = want.permute(0, 2, 3, 1).reshape(B * 4 * H * H, Cout)
    xd = ops.Act(x.permute(0, 2, 3, 1).reshape(B * H * H, Cin).contiguous().to(dev), B, H, H)
    assert ops.up_phases_supported(B * H * H, Cout, Cin)
    buf = torch.zeros((B * 4 * H * H, Cout + 64), dtype=torch.float16, device=dev)     # strided destination
    got = ops.conv2d_up_phases(xd, ops.prep_conv_up_phases(w, b, dev), out=buf[:, :Cout])
    assert (got.B, got.H, got.W) == (B, 2 * H, 2 * H)
    gotf = got.t.float().cpu()
    old = ops.conv2d(xd, ops.prep_conv(w, b, dev), up=True).t.float().cpu()
    tol = 4e-3 * float(want.abs().max()) + 2e-3
    assert float((gotf - want).abs().max()) <= tol, float((gotf - want).abs().max())
    assert float((gotf - old).abs().max()) <= tol
    assert float((gotf - want).abs().mean()) <= 1.2 * float((old - want).abs().mean()) + 1e-5
    assert float(buf[:, Cout:].abs().max()) == 0.0           # nothing written past the view
    assert not ops.up_phases_supported(512, Cout, Cin)       # too few rows: stays on the fused-upsample conv
    assert not ops.up_phases_supported(1024, 512, 512) and ops.up_phases_supported(1024, 1280, 1280)   # (VAE widths: from 4096 rows)


@pytest.mark.parametrize('M,N,K,res', [(16384, 640, 640, True), (4096, 1280, 1280, True), (4096, 1280, 1280, False),
                                       (1024, 1280, 1280, True), (16384, 640, 640, False), (128, 480, 320, True),
                                       (36864, 320, 320, True), (2304, 640, 640, True), (18432, 640, 640, False)])
def test_gemm_emits_partial_row_stats_for_wide_rows(dev, M, N, K, res):
    '''fd_gemm_desc.ln_stats_out with N > 320: the 160-wide tiles write raw (sum, sum of squares) per n-tile,
    fd_ln_finalize_stats_f32 combines them -> the same (rstd, -mean rstd) as fd_ln_row_stats_f16 on the stored rows
    (one-pass E[x^2] - mean^2 in fp32 vs exact two-pass: 2e-4), and the output itself equals the launch without
    statistics up to tile-choice rounding.  Shapes: the UNet's 32x32 / 16x16 / 8x8-level producers (tiles 13, 20, 12) and the
    9 x 2^k row counts of 768x768 images, which take the 288-row tile (N = 320 then emits two slabs instead of finished pairs).'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn((M, K), generator=g).half().to(dev)
    w = ops.prep_linear(torch.randn((N, K), generator=g) * K ** -0.5, torch.randn(N, generator=g) * 0.5, dev)
    r = (torch.randn((M, N), generator=g) * 2 + torch.randn((M, 1), generator=g) * 3).half().to(dev) if res else None
    k = ops.can_emit_row_stats(M, N, K)
    assert k == N // 160
    parts = torch.full((k, M, 2), float('nan'), dtype=torch.float32, device=dev)
    out = ops.gemm(a, w, residual=r, ln_stats_out=parts)
    assert bool(torch.isfinite(parts).all())
    st = ops.ln_finalize_stats(parts, N)
    plain = ops.gemm(a, w, residual=r)
    assert float((out.float() - plain.float()).abs().max()) <= 2e-3 * float(plain.float().abs().max())
    ref = ops.ln_row_stats(out)
    assert float(((st[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < 2e-4
    assert float((st[:, 1] - ref[:, 1]).abs().max()) < 2e-4 * max(1.0, float(ref[:, 1].abs().max()))
    # the slabs really are per-tile sums of the stored values
    x = out.float()
    assert torch.allclose(parts[1, :, 0], x[:, 160:320].sum(1), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize('B,H,C,N,ld', [(16, 64, 320, 320, 320), (8, 32, 640, 640, 640), (4, 16, 320, 320, 960), (2, 48, 320, 320, 320)])
def test_groupnorm_folded_into_proj_in(dev, B, H, C, N, ld):
    '''fd_groupnorm_fold_linear_f16 + fd_gemm_f16 with per-batch weights and bias (batch_stride_w / batch_stride_bias):
    proj_in(GroupNorm(x)) from ONE statistics pass over x, no normalised activation.  vs fp32 torch (group_norm + linear)
    and vs the unfused device path (fd_groupnorm + fd_gemm): the fold rounds W gamma rstd to fp16 instead of the normalised
    activation -- same error class (mean error <= 1.5x the unfused path's).  Also: the statistics the per-sample GEMM
    emits for the NEXT LayerNorm equal fd_ln_row_stats_f16 on its output (row index = batch * HW + m), x may be a column
    slice of a wider matrix, 9 x 2^k rows per sample.'''
    from flexdiffuse_amd import ops
    HW, G = H * H, 32
    g = torch.Generator().manual_seed(B * 1000 + C + H)
    xw = (torch.randn((B * HW, ld), generator=g) * 1.5 + torch.randn((B, 1, ld), generator=g).repeat(1, HW, 1).reshape(B * HW, ld) * 0.7)
    xw = xw.half()
    x32 = xw[:, :C].float().reshape(B, HW, C)
    w = torch.randn((N, C), generator=g) * C ** -0.5
    b = torch.randn(N, generator=g) * 0.3
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    want = F.linear(F.group_norm(x32.permute(0, 2, 1), G, gamma, beta, 1e-6).permute(0, 2, 1), w, b).reshape(B * HW, N)
    xd = xw.to(dev)
    xa = ops.Act(xd[:, :C], B, H, H)
    gf = ops.prep_gn_fold(w, b, gamma, beta, G, 1e-6, dev)
    wb, bb = ops.gn_fold_linear(xa, gf)
    k = ops.can_emit_row_stats(HW, N, C)
    st = torch.full((B * HW, 2) if k == 1 else (k, B * HW, 2), float('nan'), dtype=torch.float32, device=dev) if k else None
    got = ops.gemm_per_sample(xa.t, wb, bb, B, HW, ln_stats_out=st)
    unf = ops.gemm(ops.groupnorm(xa, ops.f32(gamma, dev), ops.f32(beta, dev), G, 1e-6, False).t, ops.prep_linear(w, b, dev))
    e_fold = (got.float().cpu() - want).abs()
    e_unf = (unf.float().cpu() - want).abs()
    scale = float(want.abs().max())
    assert float(e_fold.max()) < 1e-2 * scale, (float(e_fold.max()), scale)
    assert float(e_fold.mean()) <= 1.5 * float(e_unf.mean()) + 1e-6, (float(e_fold.mean()), float(e_unf.mean()))
    if st is not None:
        assert bool(torch.isfinite(st).all())
        s2 = st if st.dim() == 2 else ops.ln_finalize_stats(st, N)
        ref = ops.ln_row_stats(got)
        assert float(((s2[:, 0] - ref[:, 0]).abs() / ref[:, 0]).max()) < 2e-4
        assert float((s2[:, 1] - ref[:, 1]).abs().max()) < 2e-4 * max(1.0, float(ref[:, 1].abs().max()))
    # without statistics: same values up to the tile choice
    plain = ops.gemm_per_sample(xa.t, wb, bb, B, HW)
    assert float((plain.float() - got.float()).abs().max()) <= 2e-3 * scale


def test_groupnorm_fold_with_large_group_means(dev):
    '''Real UNet residual streams carry group means several times their spread.  The fold subtracts the mean through
    the bias, so it only cancels if that bias is summed over the SAME rounded fp16 weights the GEMM multiplies with
    (ABI 9); with means of 10 sigma the error must stay in the class of the unfused path, not grow 10x.'''
    from flexdiffuse_amd import ops
    B, H, C, N, G = 4, 32, 320, 320, 32
    HW = H * H
    g = torch.Generator().manual_seed(77)
    means = torch.randn((B, 1, G), generator=g).repeat_interleave(C // G, dim=2) * 10.0
    x32 = (torch.randn((B, HW, C), generator=g) + means).half().float()
    w = torch.randn((N, C), generator=g) * C ** -0.5
    b = torch.randn(N, generator=g) * 0.3
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
    want = F.linear(F.group_norm(x32.permute(0, 2, 1), G, gamma, beta, 1e-6).permute(0, 2, 1), w, b).reshape(B * HW, N)
    xa = ops.Act(x32.reshape(B * HW, C).half().to(dev), B, H, H)
    gf = ops.prep_gn_fold(w, b, gamma, beta, G, 1e-6, dev)
    wb, bb = ops.gn_fold_linear(xa, gf)
    got = ops.gemm_per_sample(xa.t, wb, bb, B, HW)
    unf = ops.gemm(ops.groupnorm(xa, ops.f32(gamma, dev), ops.f32(beta, dev), G, 1e-6, False).t, ops.prep_linear(w, b, dev))
    e_fold, e_unf = (got.float().cpu() - want).abs(), (unf.float().cpu() - want).abs()
    print(f'GN fold at |mean| = 10 sigma: mean err {float(e_fold.mean()):.2e} (unfused {float(e_unf.mean()):.2e}), '
          f'max {float(e_fold.max()):.2e} (unfused {float(e_unf.max()):.2e})')
    assert float(e_fold.mean()) <= 2.0 * float(e_unf.mean()) + 1e-6
    assert float(e_fold.max()) <= 3.0 * float(e_unf.max()) + 1e-6


def test_per_batch_bias_is_refused_where_it_cannot_be_staged(dev):
    from flexdiffuse_amd import ops
    a = torch.zeros((512, 64), dtype=torch.float16, device=dev)
    wb = torch.zeros((1, 64, 64), dtype=torch.float16, device=dev)
    bb = torch.zeros((1, 64), dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        ops.gemm_per_sample(a, wb, bb, 1, 512)      # batch_stride_bias needs batch > 1


@pytest.mark.parametrize('B,H,C,Cx,keep', [(16, 8, 1280, 0, False), (16, 16, 1280, 0, False), (16, 16, 1280, 0, True), (16, 8, 1280, 2560, True),
                                          (16, 16, 1280, 2560, True), (2, 24, 1280, 0, False), (4, 12, 1280, 0, True)])
def test_groupnorm_fused_into_the_splitk_finish(dev, B, H, C, Cx, keep):
    '''fd_gemm_desc.gn_out: the pass that sums a split-K convolution's fp32 slabs also normalises them (ResBlock conv1 -> norm2 +
    SiLU at the 16x16 / 8x8 levels; with Cx the convolution carries an appended shortcut and the un-normalised output is kept, the
    conv2 -> next block's norm form).  Against (a) the two launches it replaces -- same bits, for the output and for the normalised
    tensor --, (b) a torch fp32 reference of conv + per-sample bias + GroupNorm + SiLU on the same fp16-rounded output, (c) itself,
    20 times.'''
    from flexdiffuse_amd import ops
    G, eps, silu = 32, 1e-5, True
    g = torch.Generator().manual_seed(B * H + C + Cx)
    M = B * H * H
    hx = (torch.randn((B, C, H, H), generator=g) * 0.7).half()
    w = torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5
    b = torch.randn(C, generator=g) * 0.1
    temb = (torch.randn((B, C + 64), generator=g) * 0.5).to(dev)      # per-sample bias rows inside a wider matrix
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    hd = ops.Act(hx.permute(0, 2, 3, 1).reshape(M, C).contiguous().to(dev), B, H, H)
    spec = ops.GNSpec(gamma, beta, G, eps, silu)
    kw = dict(bias2=temb[:, 32:32 + C], ld_bias2=C + 64)
    if Cx:
        x = (torch.randn((M, Cx), generator=g) * 0.7).half().to(dev)
        ws, bs = torch.randn((C, Cx), generator=g) * Cx ** -0.5, torch.randn(C, generator=g) * 0.1
        cw = ops.prep_conv_shortcut(w, b, ws, bs, dev)
        kw = dict(a2=x)
    else:
        cw = ops.prep_conv(w, b, dev)
    assert ops.GN_FINISH_FUSE
    out_f, y_f = ops.conv2d(hd, cw, gn=spec, keep=keep, **kw)
    # the launch really is a split-K one (otherwise both paths here are the same two launches)
    assert ops._last_conv_gn_fused is True and (out_f is not None) == keep
    ops.GN_FINISH_FUSE = False
    try:
        out_u, y_u = ops.conv2d(hd, cw, gn=spec, keep=keep, **kw)
    finally:
        ops.GN_FINISH_FUSE = True
    assert ops._last_conv_gn_fused is False
    assert torch.equal(y_f.t, y_u.t), float((y_f.t.float() - y_u.t.float()).abs().max())
    if keep:
        assert torch.equal(out_f.t, out_u.t)
    # (b) torch fp32 on the fp16-rounded convolution output
    h16 = out_u.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H)
    want = F.silu(F.group_norm(h16, G, gamma.float(), beta.float(), eps))
    close(y_f.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H), want, rtol=4e-3, atol=4e-3)
    # (c) repeats bit for bit
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(20):
        _, y2 = ops.conv2d(hd, cw, gn=spec, keep=keep, **kw)
        bad += (y2.t != y_f.t).sum()
    assert int(bad) == 0


def test_groupnorm_finish_fusion_is_refused_where_it_cannot_run(dev):
    '''gn_out on a launch the rule does not split is FD_ESHAPE (the Python front asks fd_gemm_plan first and never gets there);
    fd_gemm_can_fuse_groupnorm answers for shapes and split factors.'''
    import ctypes
    from flexdiffuse_amd import hip, ops
    lib = hip.lib()
    assert lib.fd_gemm_can_fuse_groupnorm(1024, 1280, 64, 32, 8) == 1 and lib.fd_gemm_can_fuse_groupnorm(4096, 1280, 256, 32, 4) == 1
    assert lib.fd_gemm_can_fuse_groupnorm(4096, 1280, 256, 32, 1) == 0 and lib.fd_gemm_can_fuse_groupnorm(4096, 1280, 256, 32, 3) == 0
    assert lib.fd_gemm_can_fuse_groupnorm(65536, 320, 4096, 32, 2) == 0          # a 64x64 slab does not fit the finish kernel's registers
    assert lib.fd_gemm_can_fuse_groupnorm(1024, 1280, 60, 32, 8) == 0             # rows_per_sample does not divide M
    B, H, C = 2, 64, 320                                                           # level-0 shape: never split
    g = torch.Generator().manual_seed(5)
    hd = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.5).half().to(dev), B, H, H)
    cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * 0.02, torch.zeros(C), dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out, y = ops.conv2d(hd, cw, gn=ops.GNSpec(gamma, beta, 32, 1e-5, True))       # falls back to two launches
    want = F.silu(F.group_norm(out.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H), 32, gamma, beta, 1e-5))
    close(y.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H), want, rtol=4e-3, atol=4e-3)
    # a shape whose slab fits the finish kernel but that the rule does not split (64 x 16 x 16 x 1280: 256 full tiles)
    B, H, C = 64, 16, 1280
    assert lib.fd_gemm_can_fuse_groupnorm(B * H * H, C, H * H, 32, 2) == 1
    hd = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.5).half().to(dev), B, H, H)
    cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * 0.01, torch.zeros(C), dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    out, y = ops.conv2d(hd, cw, gn=ops.GNSpec(gamma, beta, 32, 1e-5, True))
    d = ops.fd_gemm_desc()
    d.A, d.W, d.C, d.bias = hd.t.data_ptr(), cw.w.data_ptr(), out.t.data_ptr(), cw.bias.data_ptr()
    d.M, d.N, d.K, d.ldw, d.ldc, d.lda = B * H * H, C, cw.kpad, cw.w.stride(0), C, C
    d.rows_per_sample, d.alpha, d.batch = H * H, 1.0, 1
    d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.kw, d.stride, d.pad_t, d.pad_l = 1, H, H, C, H, H, 3, 3, 1, 1, 1
    ops._sched(d, dev)
    tile, split = ctypes.c_int32(0), ctypes.c_int32(0)
    hip.check(lib.fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile), ctypes.byref(split)), 'fd_gemm_plan')
    assert split.value == 1
    d.gn_out, d.gn_gamma, d.gn_beta, d.gn_groups, d.gn_silu, d.gn_eps = y.t.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 32, 1, 1e-5
    with pytest.raises(ValueError, match='split-K'):
        hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())


def test_persistent_geglu_tile_race_screen_with_weights_streaming_from_hbm(dev):
    '''The race screen of test_lds_dma_tiles_race_screen_with_weights_streaming_from_hbm for the kernel that serves the GEGLU
    launches of the forward: the PERSISTENT 256x256 / 16-wave tile with the LayerNorm-fold + GEGLU epilogue
    (`k_gemm_f16_dmap<256, 256, ..., 6>`): the next tile's first K-tile is in flight under the epilogue, its counted waits are the
    only thing between an LDS-DMA and the ds_read of its piece.  52 MB of weights, every tile reads its own rows once (every W DMA is
    an HBM miss), K = 320 as at level 0; vs torch, then 400 launches bit for bit.'''
    from flexdiffuse_amd import ops
    M, N, K = 512, 81920, 320
    g = torch.Generator().manual_seed(23)
    x = (torch.randn((M, K), generator=g) * 0.8 + 0.3).half()
    w = torch.randn((N, K), generator=g) * K ** -0.5
    bias = torch.randn(N, generator=g) * 0.1
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    lw = ops.prep_linear_ln(w, bias, gamma, beta, dev, geglu=True)
    xd = x.to(dev)
    st = ops.ln_row_stats(xd)
    import ctypes
    from flexdiffuse_amd import hip
    ref = ops.gemm(xd, lw, act=ops.ACT_GEGLU, ln_stats=st).clone()
    torch.cuda.synchronize()
    hdn = F.layer_norm(x.float().to(dev), (K,), gamma.to(dev), beta.to(dev), 1e-5) @ w.to(dev).t() + bias.to(dev)
    want = hdn[:, :N // 2] * F.gelu(hdn[:, N // 2:])
    err = float((ref.float() - want).abs().max())
    assert err < 2e-2 * max(1.0, float(want.abs().max())), err
    out = torch.empty_like(ref)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(400):
        ops.gemm(xd, lw, act=ops.ACT_GEGLU, ln_stats=st, out=out)
        bad += (out != ref).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, int(bad)


@pytest.mark.parametrize('B,HW,rep,d', [(16, 4096, 2, 40), (16, 1024, 2, 80)])
def test_fused_cross_attention_race_screen(dev, B, HW, rep, d):
    '''400 launches of fd_xattn_q_f16 (`k_xattn<40>` / `k_xattn<80>`) at the shapes of the bench forward -- 16 x 64 x 64 x 320
    and 16 x 32 x 32 x 640 hidden states streaming from HBM through its LDS-DMA ring, two context replicas -- must repeat the first
    launch bit for bit (the numerics of the kernel are test_fused_q_projection_cross_attention's subject).'''
    from flexdiffuse_amd import ops
    heads, L = 8, 77
    C, M = heads * d, B * HW
    g = torch.Generator().manual_seed(B + HW + d)
    xd = (torch.randn((M, C), generator=g) * 1.5 + 0.5).half().to(dev)
    lw = ops.prep_linear_ln(torch.randn((C, C), generator=g) * C ** -0.5 * ops.QK_LOG2E * d ** -0.5, None,
                            1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g), dev)
    kd = (torch.randn((rep * B * L, C), generator=g) * 1.2).half().to(dev)
    ldv = (L + 7) // 8 * 8
    vt = torch.zeros((rep * B, C, ldv), dtype=torch.float16)
    vt[:, :, :L] = torch.randn((rep * B, C, L), generator=g).half()
    st = ops.ln_row_stats(xd)
    assert ops.xattn_supported(heads, d, L, HW)
    img = ops.xattn_pack_kv(kd, vt.to(dev), rep * B, L, heads, d)
    ref = ops.xattn_q(xd, lw, st, img, HW, L, heads, d, n_rep=rep).clone()
    assert bool(torch.isfinite(ref.float()).all()) and float(ref.float().abs().max()) > 0.1
    out = torch.empty_like(ref)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(400):
        ops.xattn_q(xd, lw, st, img, HW, L, heads, d, n_rep=rep, out=out)
        bad += (out != ref).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, int(bad)


@pytest.mark.parametrize('B,H,Cx,res', [(16, 64, 0, False), (8, 64, 0, False), (16, 64, 0, True), (16, 64, 640, False), (16, 64, 960, False)])
def test_groupnorm_partial_sums_from_the_producing_convolution(dev, B, H, Cx, res):
    '''fd_gemm_desc.gn_part_out: the lean epilogue of a 320-wide row-spanning tile (ping-pong 256x320 / 128x320, the 2-barrier 256x320
    tile of the convolutions with an appended shortcut) writes per-(sample, row chunk, group) partial (sum, sum of squares) of its
    fp16-rounded output, so the GroupNorm behind it -- a ResBlock's norm2 behind conv1, the transformer block's folded input norm
    behind conv2 -- runs without its statistics pass.  Checks: the partial sums against torch on the stored output; the apply pass
    from them against torch GroupNorm + SiLU and against the two-pass kernel (the statistics differ in summation order only: a few
    fp16 ulps); the folded per-sample weights against the statistics-pass form; 20 repeats bit for bit.'''
    from flexdiffuse_amd import ops
    C, G, eps = 320, 32, 1e-5
    g = torch.Generator().manual_seed(B * H + Cx + int(res))
    M = B * H * H
    hx = (torch.randn((M, C), generator=g) * 0.7 + 0.2).half().to(dev)
    w = torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5 * 1.5
    b = torch.randn(C, generator=g) * 0.3
    temb = (torch.randn((B, C), generator=g) * 0.5).to(dev)
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(dev), (0.3 * torch.randn(C, generator=g)).to(dev)
    hd = ops.Act(hx, B, H, H)
    kw = {}
    if Cx:
        cw = ops.prep_conv_shortcut(w, b, torch.randn((C, Cx), generator=g) * Cx ** -0.5, None, dev)
        kw['a2'] = (torch.randn((M, Cx), generator=g) * 0.7).half().to(dev)
    else:
        cw = ops.prep_conv(w, b, dev)
        kw.update(bias2=temb, ld_bias2=C)
    if res:
        kw['residual'] = (torch.randn((M, C), generator=g) * 2.0 + 1.0).half().to(dev)     # group means far from zero
    out, parts = ops.conv2d(hd, cw, gn_parts=G, **kw)
    assert parts is not None and parts.t.shape == (B, parts.chunks, G, 2) and parts.chunks in (16, 32), 'the rule no longer emits partial sums here'
    plain = ops.conv2d(hd, cw, **kw)
    # the output itself: the same values -- up to the last fp16 bit of a few elements in 10^5: with the statistics code next to it hipcc
    # rounds some accumulators fp32 -> fp16 in one step (v_fma_mix*_f16) where the plain epilogue rounds fma -> fp32 -> fp16
    dd = (out.t.float() - plain.t.float()).abs()
    assert float((dd > 0).float().mean()) < 1e-3 and float((dd / plain.t.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -9
    o32 = out.t.float().view(B, parts.chunks, H * H // parts.chunks, G, C // G)
    want_s, want_q = o32.sum(dim=(2, 4)), (o32 * o32).sum(dim=(2, 4))
    assert float((parts.t[..., 0] - want_s).abs().max()) <= 2e-4 * float(want_s.abs().max()) + 1e-2
    assert float((parts.t[..., 1] - want_q).abs().max()) <= 2e-4 * float(want_q.abs().max())
    # apply from the partial sums
    y = ops.groupnorm(out, gamma, beta, G, eps, True, parts=parts)
    y2 = ops.groupnorm(out, gamma, beta, G, eps, True)
    o4 = out.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H)
    want = F.silu(F.group_norm(o4, G, gamma, beta, eps))
    close(y.t.float().view(B, H * H, C).permute(0, 2, 1).reshape(B, C, H, H), want, rtol=4e-3, atol=4e-3)
    d = (y.t.float() - y2.t.float()).abs()
    assert float(d.max()) <= 4e-3 * float(y2.t.float().abs().max()) and float((d > 0).float().mean()) < 0.2
    # conv2d(..., gn=) takes the same route when the launch is not split over K
    _, y3 = ops.conv2d(hd, cw, gn=ops.GNSpec(gamma, beta, G, eps, True), **kw)
    assert torch.equal(y3.t, y.t) and ops._last_conv_gn_fused is False
    # the folded form: per-sample weights / bias from the partial sums vs from the statistics pass
    wl = torch.randn((C, C), generator=g) * C ** -0.5
    gf = ops.prep_gn_fold(wl, torch.randn(C, generator=g) * 0.1, gamma.cpu(), beta.cpu(), G, 1e-6, dev)
    wb, bb = ops.gn_fold_linear(out, gf, parts=parts)
    wb2, bb2 = ops.gn_fold_linear(out, gf)
    assert float((wb.float() - wb2.float()).abs().max()) <= 2e-3 * float(wb2.float().abs().max())
    assert float((bb - bb2).abs().max()) <= 2e-3 * max(1.0, float(bb2.abs().max()))
    # repeats bit for bit
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(20):
        _, p2 = ops.conv2d(hd, cw, gn_parts=G, **kw)
        bad += (p2.t != parts.t).sum()
    assert int(bad) == 0


def test_groupnorm_partial_sums_are_refused_where_no_tile_spans_the_row(dev):
    '''gn_part_out on a launch whose tile does not span the row / whose launch is split answers FD_ESHAPE; fd_gemm_gn_parts_chunks says
    so beforehand (the Python front asks it and falls back to the statistics pass).'''
    import ctypes
    from flexdiffuse_amd import hip, ops
    g = torch.Generator().manual_seed(9)
    for (B, H, C) in ((16, 16, 1280), (16, 32, 640)):
        hd = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.5).half().to(dev), B, H, H)
        cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * 0.01, torch.zeros(C), dev)
        out, parts = ops.conv2d(hd, cw, gn_parts=32)
        assert parts is None
        d = ops.fd_gemm_desc()
        d.A, d.W, d.C, d.bias = hd.t.data_ptr(), cw.w.data_ptr(), out.t.data_ptr(), cw.bias.data_ptr()
        d.M, d.N, d.K, d.ldw, d.ldc, d.lda = B * H * H, C, cw.kpad, cw.w.stride(0), C, C
        d.rows_per_sample, d.alpha, d.batch = H * H, 1.0, 1
        d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.kw, d.stride, d.pad_t, d.pad_l = 1, H, H, C, H, H, 3, 3, 1, 1, 1
        ops._sched(d, dev)
        d.gn_groups = 32
        assert hip.lib().fd_gemm_gn_parts_chunks(ctypes.byref(d)) == 0
        buf = torch.empty((B, 64, 32, 2), dtype=torch.float32, device=dev)
        d.gn_part_out, d.gn_part_chunks = buf.data_ptr(), 64
        with pytest.raises(ValueError, match='gn_part_out'):
            hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    # a buffer sized for the rule's tile is not written by a forced tile with another row count (256-row tile: 16 chunks per sample,
    # 128-row tile: 32): FD_ESHAPE instead of a write past the end
    B, H, C = 16, 64, 320
    hd = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.5).half().to(dev), B, H, H)
    cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * 0.02, torch.zeros(C), dev)
    out, parts = ops.conv2d(hd, cw, gn_parts=32)
    assert parts is not None and parts.chunks == 16
    old = ops.FORCE_TILE
    ops.FORCE_TILE = 32
    try:
        with pytest.raises(ValueError, match='gn_part_out'):
            d = ops.fd_gemm_desc()
            d.A, d.W, d.C, d.bias = hd.t.data_ptr(), cw.w.data_ptr(), out.t.data_ptr(), cw.bias.data_ptr()
            d.M, d.N, d.K, d.ldw, d.ldc, d.lda = B * H * H, C, cw.kpad, cw.w.stride(0), C, C
            d.rows_per_sample, d.alpha, d.batch = H * H, 1.0, 1
            d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.kw, d.stride, d.pad_t, d.pad_l = 1, H, H, C, H, H, 3, 3, 1, 1, 1
            ops._sched(d, dev)
            d.gn_groups, d.gn_part_out, d.gn_part_chunks = 32, parts.t.data_ptr(), parts.chunks
            hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    finally:
        ops.FORCE_TILE = old


@pytest.mark.parametrize('B,Cin,H,W,Cout,rep2', [(8, 4, 64, 64, 320, 2), (2, 4, 96, 96, 320, 0), (1, 3, 17, 23, 64, 3), (2, 4, 8, 8, 1280, 2)])
def test_conv3x3_narrow_input_in_one_launch(dev, B, Cin, H, W, Cout, rep2):
    '''fd_conv3x3_narrow_f16 (the UNet's conv_in: 4 latent channels -> 320 from the fp32 NCHW latents, replicas of the output for the CFG
    fan-out included) against the three launches it replaces (layout change, explicit im2col, GEMM: the same fp16 products in another
    summation order -> within fp16 rounding of each other) and against torch's fp32 convolution; ragged widths, borders, a column-slice
    destination; repeats bit for bit.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Cout)
    x = torch.randn((B, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * (9 * Cin) ** -0.5
    b = torch.randn(Cout, generator=g)
    nw = ops.prep_conv_narrow(w, b, dev)
    assert nw is not None and ops.prep_conv_narrow(torch.zeros((64, 8, 3, 3)), None, dev) is None
    xd = x.to(dev)
    M = B * H * W
    buf = torch.full((max(rep2, 1) * M, Cout + 64), 3.0, dtype=torch.float16, device=dev)
    view = buf[:, 64:]
    got = ops.conv3x3_narrow(xd, nw, out2=view if rep2 else None, rep2=rep2, scale=0.5)
    again = ops.conv3x3_narrow(xd, nw, scale=0.5)
    assert torch.equal(got.t, again.t) and bool(torch.isfinite(got.t.float()).all())
    if rep2:
        assert torch.equal(view, torch.cat([got.t] * rep2, 0)) and bool((buf[:, :64] == 3.0).all())
    old = ops.conv2d(ops.nchw_to_nhwc(xd, c_pad=8, scale=0.5), ops.prep_conv(w, b, dev, cin_pad=8))
    d = (got.t.float() - old.t.float()).abs()
    assert float(d.max()) <= 2.0 ** -9 * max(1.0, float(old.t.float().abs().max())), float(d.max())
    ref = torch.nn.functional.conv2d((x * 0.5).half().float(), w.half().float(), b, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    close(got.t.float().cpu(), ref, rtol=2e-3, atol=2e-3)
    with pytest.raises(ValueError):
        ops.conv3x3_narrow(xd, nw, out=torch.empty((M, Cout + 4), dtype=torch.float16, device=dev)[:, :Cout])     # row stride not a multiple of 8


def test_repeat_rows_is_one_launch_and_equals_the_copies(dev):
    '''fd_repeat_rows_f16: the CFG fan-out (B samples -> rep * B) as one launch, into a contiguous tensor or into a column slice of a wider
    buffer (the skip tensors' concat buffers); equals rep separate fd_copy2d_f16 calls.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((1000, 320), generator=g).half().to(dev)
    xs = torch.zeros((1000, 384), dtype=torch.float16, device=dev)
    xs[:, 32:352] = x
    for src in (x, xs[:, 32:352]):
        out = ops.repeat_rows(src, 3)
        assert torch.equal(out, torch.cat([x, x, x], 0))
        buf = torch.full((2000, 640), 7.0, dtype=torch.float16, device=dev)
        ops.repeat_rows(src, 2, out=buf[:, 320:])
        assert torch.equal(buf[:, 320:], torch.cat([x, x], 0)) and bool((buf[:, :320] == 7.0).all())
    with pytest.raises(ValueError):
        ops.repeat_rows(x[:, :36], 2)                  # 36 columns: not a multiple of 8


@pytest.mark.parametrize('M,rows,N,K,K2,stats', [(65536, 32768, 320, 320, 0, True),     # the out-projection behind the CFG fan-out (256x320 tile, statistics epilogue)
                                                  (65536, 32768, 320, 1280, 320, False),  # FF-out + folded proj_out, appended operand (ping-pong tile)
                                                  (4096, 1024, 320, 320, 0, False), (2048, 512, 1280, 640, 0, False), (1536, 768, 640, 320, 0, True),
                                                  (3072, 1024, 200, 96, 0, False),         # generic epilogue (ragged N)
                                                  (18432, 9216, 320, 320, 0, False)])      # c4 at batch 1: the rule splits this one over K (the finish pass wraps)
def test_gemm_residual_read_modulo_a_row_count(dev, M, rows, N, K, K2, stats):
    '''fd_gemm_desc.residual_rows (ops.gemm with a residual of fewer rows): output row m adds residual row m % rows -- the residual of the
    replicated rows of a shared prefix without the replicas.  Same bits as the GEMM fed with the materialised replicas, on the lean
    statistics epilogue, the ping-pong tile with an appended operand, small tiles and the generic epilogue.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn((M, K), generator=g).half().to(dev)
    a2 = torch.randn((M, K2), generator=g).half().to(dev) if K2 else None
    w = ops.prep_linear(torch.randn((N, K + K2), generator=g) * (K + K2) ** -0.5, torch.randn(N, generator=g), dev)
    res = torch.randn((rows, N), generator=g).half().to(dev)
    full = ops.repeat_rows(res, M // rows) if N % 8 == 0 else torch.cat([res] * (M // rows), 0)
    kw = {}
    if stats and ops.can_emit_row_stats(M, N, K, N, N) == 1:
        kw = dict(ln_stats_out=torch.zeros((M, 2), dtype=torch.float32, device=dev))
    ops.FORCE_SPLIT = 2 if M == 18432 else 0      # (the split path -- its finish pass adds the residual -- must wrap too, whatever the rule picks)
    try:
        got = ops.gemm(a, w, a2=a2, residual=res, **kw).clone()
        st_got = kw['ln_stats_out'].clone() if kw else None
        want = ops.gemm(a, w, a2=a2, residual=full, **kw)
    finally:
        ops.FORCE_SPLIT = 0
    assert torch.equal(got, want) and bool(torch.isfinite(got.float()).all())
    if kw:
        assert torch.equal(st_got, kw['ln_stats_out'])
    ref = a.float().cpu() @ w.w.float().cpu()[:, :K].T + (a2.float().cpu() @ w.w.float().cpu()[:, K:].T if K2 else 0) + w.bias.cpu() + full.float().cpu()
    close(got[:, :N].float().cpu(), ref, rtol=4e-3, atol=2e-2)


def test_gemm_residual_rows_is_refused_where_it_cannot_run(dev):
    from flexdiffuse_amd import hip, ops
    a = torch.randn((1024, 320)).half().to(dev)
    w = ops.prep_linear(torch.randn((320, 320)) * 0.05, None, dev)
    with pytest.raises(AssertionError):
        ops.gemm(a, w, residual=torch.zeros((384, 320), dtype=torch.float16, device=dev))      # 1024 % 384 != 0
    with pytest.raises((hip.FDError, ValueError)):
        ops.gemm(a, w, residual=torch.zeros((128, 320), dtype=torch.float16, device=dev))      # a 256-row tile would straddle the wrap
    assert ops.residual_wrap_supported(32768, 2) and ops.residual_wrap_supported(36864, 2) and not ops.residual_wrap_supported(1000, 2)
    assert not ops.residual_wrap_supported(768, 3)          # 2304 output rows could take the 288-row tile, 768 is not a multiple of it


@pytest.mark.parametrize('B,HW,C', [(16, 4096, 320), (8, 4096, 320), (16, 1024, 640), (16, 256, 1280), (16, 64, 1280), (2, 576, 640)])
def test_qkv_projection_with_a_transposed_tail(dev, B, HW, C):
    '''fd_gemm_desc.trans_n0 / C2 (ops.gemm_qkv): the self-attention's q | k | v projection from LayerNorm-folded weights in ONE launch --
    q|k row-major, V transposed into the layout fd_attention_f16 takes -- against the two launches it replaces (same bits: no split over K,
    the same MFMA summation order whatever the tile) and against a torch fp32 LayerNorm + linear; repeats bit for bit.'''
    from flexdiffuse_amd import ops
    M = B * HW
    g = torch.Generator().manual_seed(B + HW + C)
    x = (torch.randn((M, C), generator=g) * (0.5 + torch.rand((M, 1), generator=g)) + torch.randn((M, 1), generator=g)).half()
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    wqk, wv = torch.randn((2 * C, C), generator=g) * C ** -0.5, torch.randn((C, C), generator=g) * C ** -0.5
    lqk, lv = ops.prep_linear_ln(wqk, None, gamma, beta, dev), ops.prep_linear_ln(wv, None, gamma, beta, dev)
    lqkv = ops.prep_linear_ln(torch.cat([wqk, wv], 0), None, gamma, beta, dev)
    xd = x.to(dev)
    st = ops.ln_row_stats(xd)
    assert ops.qkv_merge_supported(M, C, HW)
    qk, vt = ops.gemm_qkv(xd, lqkv, B, HW, st)
    qk2 = ops.gemm(xd, lqk, ln_stats=st)
    vt2 = ops.gemm_vt(xd, lv, B, HW, HW, ln_stats=st)
    assert qk.shape == (M, 2 * C) and vt.shape == (B, C, HW)
    assert torch.equal(qk, qk2), float((qk.float() - qk2.float()).abs().max())
    assert torch.equal(vt, vt2), float((vt.float() - vt2.float()).abs().max())
    xn = F.layer_norm(x.float().to(dev), (C,), gamma.to(dev), beta.to(dev), 1e-5)
    close(qk, xn @ wqk.to(dev).t(), rtol=6e-3, atol=6e-3)
    close(vt, (xn @ wv.to(dev).t()).view(B, HW, C).permute(0, 2, 1), rtol=6e-3, atol=6e-3)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(20):
        q3, v3 = ops.gemm_qkv(xd, lqkv, B, HW, st)
        bad += (q3 != qk).sum() + (v3 != vt).sum()
    assert int(bad) == 0


def test_transposed_tail_is_refused_outside_its_shapes(dev):
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(2)
    C = 320
    lqkv = ops.prep_linear_ln(torch.randn((3 * C, C), generator=g) * 0.05, None, torch.ones(C), torch.zeros(C), dev)
    x = torch.randn((200, C), generator=g).half().to(dev)          # 200 rows: not a multiple of 128
    assert not ops.qkv_merge_supported(200, C, 100)
    with pytest.raises(ValueError, match='trans_n0'):
        ops.gemm_qkv(x, lqkv, 2, 100, ops.ln_row_stats(x))


@pytest.mark.parametrize('B,H', [(16, 8), (16, 16)])
def test_experimental_inlaunch_splitk_reduction_gives_the_finish_launch_bits(dev, B, H):
    '''fd_gemm_desc.sk_sync (EXPERIMENTAL, ops.SPLITK_INLAUNCH; measured slower than the finish launch and left off: profiles/r06_seam_probe.txt):
    every K-slice workgroup of a ping-pong split-K launch publishes its slab, arrives on its tile's counter (agent-scope release / acquire, bounded
    spin) and finishes 1/S of the tile in slice order.  Same bits as the partial pass + k_splitk_finish, with a per-sample bias and a residual;
    the counters are left re-armed (100 launches in a row).'''
    from flexdiffuse_amd import ops
    C = 1280
    g = torch.Generator().manual_seed(B + H)
    M = B * H * H
    x = ops.Act((torch.randn((M, C), generator=g) * 0.7).half().to(dev), B, H, H)
    cw = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g), dev)
    kw = dict(bias2=torch.randn((B, C), generator=g).to(dev), ld_bias2=C, residual=torch.randn((M, C), generator=g).half().to(dev))
    ref = ops.conv2d(x, cw, **kw).t.clone()
    old = ops.SPLITK_INLAUNCH
    ops.SPLITK_INLAUNCH = True
    try:
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(100):
            bad += (ops.conv2d(x, cw, **kw).t != ref).sum()
        sync = ops._sk_sync[(0, ops.WS_SLOT)]
        assert int(bad) == 0 and int(sync.abs().sum()) == 0          # ... and it really ran (the counters exist) and left them at zero
    finally:
        ops.SPLITK_INLAUNCH = old


@pytest.mark.parametrize('M,C,HW', [(16384, 640, 1024), (4096, 1280, 256), (1024, 1280, 64), (2304, 640, 576)])
def test_layernorm_partial_sums_finalised_by_the_consumer_tiles(dev, M, C, HW):
    '''fd_gemm_desc.ln_stats_parts: the LayerNorm-fold consumers of a wide row (q|k, V^T, the merged q|k|v, the cross-attention's q, GEGLU at
    K >= 1280) take the producer's k partial slabs as they are -- every tile finalises its own rows into LDS with k_ln_finalize's arithmetic --
    instead of the finished pairs of a fd_ln_finalize_stats_f32 launch.  Same bits, consumer by consumer; the producer is a real statistics-
    emitting GEMM (o-projection + residual).'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(M + C)
    B = M // HW
    x = (torch.randn((M, C), generator=g) * 0.8).half().to(dev)
    res = (torch.randn((M, C), generator=g) * (0.5 + torch.rand((M, 1), generator=g)) + torch.randn((M, 1), generator=g)).half().to(dev)
    wo = ops.prep_linear(torch.randn((C, C), generator=g) * C ** -0.5, torch.randn(C, generator=g) * 0.1, dev)
    k = ops.can_emit_row_stats(M, C, C)
    assert k in (2, 4, 8), k
    parts = torch.empty((k, M, 2), dtype=torch.float32, device=dev)
    h = ops.gemm(x, wo, residual=res, ln_stats_out=parts)
    fin = ops.ln_finalize_stats(parts, C)
    close(fin, ops.ln_row_stats(h), rtol=2e-3, atol=2e-3)            # the slabs really are this tensor's statistics
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    wqk, wv = torch.randn((2 * C, C), generator=g) * C ** -0.5, torch.randn((C, C), generator=g) * C ** -0.5
    lqk, lv = ops.prep_linear_ln(wqk, None, gamma, beta, dev), ops.prep_linear_ln(wv, None, gamma, beta, dev)
    lq = ops.prep_linear_ln(wv * 0.7, None, gamma, beta, dev)
    lqkv = ops.prep_linear_ln(torch.cat([wqk, wv], 0), None, gamma, beta, dev)
    consumers = {
        'q|k': lambda st: ops.gemm(h, lqk, ln_stats=st),
        'q (cross-attention)': lambda st: ops.gemm(h, lq, ln_stats=st),
        'V^T': lambda st: ops.gemm_vt(h, lv, B, HW, HW, ln_stats=st),
    }
    if ops.qkv_merge_supported(M, C, HW):
        consumers['q|k|v, transposed tail (q|k)'] = lambda st: ops.gemm_qkv(h, lqkv, B, HW, st)[0]
        consumers['q|k|v, transposed tail (V^T)'] = lambda st: ops.gemm_qkv(h, lqkv, B, HW, st)[1]
    if C >= 1280:
        lff = ops.prep_linear_ln(torch.randn((8 * C, C), generator=g) * C ** -0.5, torch.randn(8 * C, generator=g) * 0.1, gamma, beta, dev, geglu=True)
        consumers['GEGLU'] = lambda st: ops.gemm(h, lff, act=ops.ACT_GEGLU, ln_stats=st)
    for name, fn in consumers.items():
        want = fn(fin).clone()
        got = fn(parts).clone()
        assert torch.equal(got, want), (name, float((got.float() - want.float()).abs().max()))
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for _ in range(10):
            bad += (fn(parts) != want).sum()
        assert int(bad) == 0, name
    # ... and against torch: LayerNorm(h) @ Wqk^T
    xn = F.layer_norm(h.float(), (C,), gamma.to(dev), beta.to(dev), 1e-5)
    close(consumers['q|k'](parts), xn @ wqk.to(dev).t(), rtol=8e-3, atol=8e-3)
