'''The deep-pipelined ("ping-pong") GEMM / implicit-GEMM convolution kernels of csrc/gemm_pp.hip (tile ids 30..33),
forced through fd_gemm_desc.tile, against plain PyTorch fp32 references on the same fp16-rounded inputs -- every epilogue
the kernels are instantiated with, split-K partial slabs, the appended A2 phase, odd
K-tile counts (tail phase) and a single K-tile.  Needs an MI355X.

Tolerance: fp16 output rounding (2^-11 relative) + fp32 accumulation order => 3e-3 relative + 3e-3 absolute.'''
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TILES = {30: (256, 320), 31: (256, 256), 32: (128, 320), 33: (256, 160)}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture(autouse=True)
def _reset_force():
    from flexdiffuse_amd import ops
    yield
    ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).half().float()


def close(got, want, rtol=3e-3, atol=3e-3):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    excess = err - (atol + rtol * want.abs())
    if bool((excess > 0).any()):
        i = int(excess.argmax())
        raise AssertionError(f'{int((excess > 0).sum())} of {err.numel()} elements out of tolerance; worst at flat index {i}: got '
                             f'{got.flatten()[i].item():.6g} want {want.flatten()[i].item():.6g} (max err {err.max().item():.4g}, max |want| {want.abs().max().item():.4g})')


@pytest.mark.parametrize('tile', sorted(TILES))
@pytest.mark.parametrize('K', [64, 128, 192, 256, 320, 1280])
def test_pp_linear_epilogues(dev, tile, K):
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    M, N = 2 * bm, 2 * bn
    a, w, b, res = rnd((M, K), 1), rnd((N, K), 2, K ** -0.5), rnd((N,), 3), rnd((M, N), 4)
    lw = ops.prep_linear(w, b, dev)
    ad = a.half().to(dev)
    y = a @ w.T + b
    ops.FORCE_TILE = tile
    close(ops.gemm(ad, lw), y)                                                        # lean plain
    close(ops.gemm(ad, lw, residual=res.half().to(dev)), y + res)                     # lean + residual
    close(ops.gemm(ad, lw, act=ops.ACT_SILU, residual=res.half().to(dev)), F.silu(y) + res)   # generic
    close(ops.gemm(ad, lw, act=ops.ACT_QUICK_GELU, out_f32=True), y * torch.sigmoid(1.702 * y), rtol=1e-3, atol=1e-3)
    b2 = rnd((2, N), 5)
    out = ops.gemm(ad, lw, bias2=b2.to(dev), ld_bias2=N, rows_per_sample=bm)          # per-sample bias, one sample per m-tile
    close(out, y + b2.repeat_interleave(bm, dim=0))


@pytest.mark.parametrize('tile', sorted(TILES))
def test_pp_split_k_and_appended_phase(dev, tile):
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    M, N, K, K2 = bm, bn, 1152, 128
    a, a2 = rnd((M, K), 1), rnd((M, K2), 6)
    w, b = rnd((N, K + K2), 2, (K + K2) ** -0.5), rnd((N,), 3)
    lw = ops.prep_linear(w, b, dev)
    ops.FORCE_TILE = tile
    for split in (1, 2, 4):
        ops.FORCE_SPLIT = split
        out = ops.gemm(a.half().to(dev), lw, a2=a2.half().to(dev))
        close(out, torch.cat([a, a2], 1) @ w.T + b)
    # split-K of a plain GEMM with an odd K-tile count per slice (18 K-tiles over 4 slices: 5, 5, 5, 3)
    lw1 = ops.prep_linear(w[:, :K], b, dev)
    ops.FORCE_SPLIT = 4
    close(ops.gemm(a.half().to(dev), lw1), a @ w[:, :K].T + b)


@pytest.mark.parametrize('tile', [30, 32])
def test_pp_split_k_slices_pinned_to_xcds(dev, tile):
    '''Tile counts that are multiples of 8 / split_k take the flat split-K grid (K slices pinned to XCDs: slice = (block % 8) / (8 / split_k)):
    every (tile, slice) pair must be visited exactly once -- linear GEMM and convolution, split 2 / 4 / 8.'''
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    M, N, K = 8 * bm, 2 * bn, 1536
    a, w, b = rnd((M, K), 21), rnd((N, K), 22, K ** -0.5), rnd((N,), 23)
    lw = ops.prep_linear(w, b, dev)
    ops.FORCE_TILE = tile
    for split in (2, 4, 8):
        ops.FORCE_SPLIT = split
        close(ops.gemm(a.half().to(dev), lw), a @ w.T + b)
    B, H, W, cin, cout = 8, 16, 16 * bm // 256, 128, bn          # 8 m-tiles
    x, cw, cb = rnd((B, cin, H, W), 24), rnd((cout, cin, 3, 3), 25, (9 * cin) ** -0.5), rnd((cout,), 26)
    want = F.conv2d(x, cw, cb, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    for split in (2, 4):
        ops.FORCE_SPLIT = split
        y = ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), ops.prep_conv(cw, cb, dev))
        close(y.t[:, :cout], want)


def test_pp_geglu_and_layernorm_fold(dev):
    from flexdiffuse_amd import ops
    M, C = 512, 320
    x = (rnd((M, C), 1, 2.0) + 0.5).half().float()      # rows with a mean, exactly representable in fp16
    wg, bg = rnd((8 * C, C), 6, C ** -0.5), rnd((8 * C,), 7)
    gamma, beta = rnd((C,), 8, 0.3) + 1.0, rnd((C,), 9, 0.2)
    ops.FORCE_TILE = 31
    out = ops.gemm(x.half().to(dev), ops.prep_geglu(wg, bg, dev), act=ops.ACT_GEGLU)
    val, gate = (x @ wg.T + bg).chunk(2, dim=-1)
    close(out, val * F.gelu(gate))
    # LayerNorm folded into the GEGLU projection (EPI 6) and into a plain projection (EPI 5)
    xd = x.half().to(dev)
    stats = ops.ln_row_stats(xd)
    xn = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    out = ops.gemm(xd, ops.prep_linear_ln(wg, bg, gamma, beta, dev, geglu=True), act=ops.ACT_GEGLU, ln_stats=stats)
    val, gate = (xn @ wg.T + bg).chunk(2, dim=-1)
    close(out, val * F.gelu(gate), rtol=6e-3, atol=6e-3)
    wq = rnd((1280, C), 10, C ** -0.5)
    for tile in (30, 31, 32, 33):
        ops.FORCE_TILE = tile
        out = ops.gemm(xd, ops.prep_linear_ln(wq, None, gamma, beta, dev), ln_stats=stats)
        close(out, xn @ wq.T, rtol=6e-3, atol=6e-3)


@pytest.mark.parametrize('tile', sorted(TILES))
def test_pp_conv3x3(dev, tile):
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    B, H, W = 2, 16, 24            # Wo % 8 == 0; M = 768 -> pad rows up to the tile by batch
    cin, cout = 128, bn
    while (B * H * W) % bm:
        B += 1
    x, w, b = rnd((B, cin, H, W), 1), rnd((cout, cin, 3, 3), 2, (9 * cin) ** -0.5), rnd((cout,), 3)
    cw = ops.prep_conv(w, b, dev)
    xa = ops.nchw_to_nhwc(x.to(dev))
    want = F.conv2d(x, w, b, padding=1)
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    ops.FORCE_TILE = tile
    y = ops.conv2d(xa, cw)
    close(y.t[:, :cout], nhwc(want))
    # per-sample bias (ResBlock time embedding) + residual + split-K
    b2 = rnd((B, cout), 4)
    res = rnd((B * H * W, cout), 5)
    if (H * W) % bm == 0:
        y = ops.conv2d(xa, cw, bias2=b2.to(dev), ld_bias2=cout)
        close(y.t[:, :cout], nhwc(want) + b2.repeat_interleave(H * W, dim=0))
    y = ops.conv2d(xa, cw, residual=res.half().to(dev))
    close(y.t[:, :cout], nhwc(want) + res)
    ops.FORCE_SPLIT = 2
    y = ops.conv2d(xa, cw, residual=res.half().to(dev))
    close(y.t[:, :cout], nhwc(want) + res)
    ops.FORCE_SPLIT = 0
    # appended 1x1 shortcut (conv2 of a ResBlock with a channel change)
    cx = 64
    xs, ws, bs = rnd((B * H * W, cx), 6), rnd((cout, cx, 1, 1), 7, cx ** -0.5), rnd((cout,), 8)
    cws = ops.prep_conv_shortcut(w, b, ws, bs, dev)
    y = ops.conv2d(xa, cws, a2=xs.half().to(dev))
    close(y.t[:, :cout], nhwc(want) + xs @ ws.view(cout, cx).T + bs)


@pytest.mark.parametrize('tile', [30, 32])
def test_pp_conv_stride2_and_asymmetric_padding(dev, tile):
    '''The rule can hand a stride-2 (Downsample2D) convolution to a ping-pong tile at larger batches: stride 2 with symmetric padding 1, and
    the UNet's downsampler form (no top / left padding, one row / column of zeros at the bottom / right: pad (0, 0) with out = in / 2).'''
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    B, H, W, cin, cout = 4, 32, 32, 64, bn            # out 16 x 16: M = 1024 = 4 or 8 tiles
    x, w, b = rnd((B, cin, H, W), 31), rnd((cout, cin, 3, 3), 32, (9 * cin) ** -0.5), rnd((cout,), 33)
    cw = ops.prep_conv(w, b, dev)
    xa = ops.nchw_to_nhwc(x.to(dev))
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    ops.FORCE_TILE = tile
    y = ops.conv2d(xa, cw, stride=2)
    close(y.t[:, :cout], nhwc(F.conv2d(x, w, b, stride=2, padding=1)))
    y = ops.conv2d(xa, cw, stride=2, pad=(0, 0), out_hw=(H // 2, W // 2))
    close(y.t[:, :cout], nhwc(F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)))


def test_pp_agrees_with_the_2barrier_kernels_at_full_size(dev):
    '''BASELINE configs[1] sizes (CFG batch 16): the ping-pong kernels against the 2-barrier kernels of gemm.hip on the SAME device inputs -- two
    independent main loops, DMA address paths and K orders computing the same fp32 sums: the level-0 convolution (tile 30 vs 16), the
    16x16-level convolution with split-K (256x320 x 4 slices pinned to XCDs vs 256x160 x 2), FF-out with the folded proj_out operand and a
    residual (tile 30 vs 16).  fp16 outputs of the two paths may differ by one rounding of a sum accumulated in another order.'''
    from flexdiffuse_amd import ops
    g = torch.Generator().manual_seed(3)

    def both(fn, arms):
        outs = []
        for tile, split in arms:
            ops.FORCE_TILE, ops.FORCE_SPLIT = tile, split
            outs.append(fn().float())
        ops.FORCE_TILE, ops.FORCE_SPLIT = 0, 0
        ref = outs[-1]
        for o in outs[:-1]:
            err = (o - ref).abs()
            assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), float(err.max())
        assert float(ref.abs().max()) > 1.0
    for (B, H, C, arms) in ((16, 64, 320, ((30, 1), (16, 1))), (16, 16, 1280, ((30, 4), (13, 2)))):
        x = ops.Act((torch.randn((B * H * H, C), generator=g) * 0.7).half().to(dev), B, H, H)
        w = ops.prep_conv(torch.randn((C, C, 3, 3), generator=g) * (9 * C) ** -0.5, torch.randn(C, generator=g), dev)
        b2 = torch.randn((B, C), generator=g).to(dev)
        both(lambda: ops.conv2d(x, w, bias2=b2, ld_bias2=C).t, arms)
    M, N, K, K2 = 65536, 320, 1280, 320
    a, a2 = torch.randn((M, K), generator=g).half().to(dev), torch.randn((M, K2), generator=g).half().to(dev)
    res = torch.randn((M, N), generator=g).half().to(dev)
    lw = ops.prep_linear(torch.randn((N, K + K2), generator=g) * K ** -0.5, torch.randn(N, generator=g), dev)
    both(lambda: ops.gemm(a, lw, a2=a2, residual=res), ((30, 1), (16, 1)))


def test_pp_refuses_what_it_cannot_run(dev):
    from flexdiffuse_amd import ops
    ops.FORCE_TILE = 30
    a, w = rnd((300, 64), 1), rnd((320, 64), 2)
    with pytest.raises(ValueError):          # ragged M
        ops.gemm(a.half().to(dev), ops.prep_linear(w, None, dev))
    x, cw = rnd((1, 64, 16, 20), 3), rnd((320, 64, 3, 3), 4, 0.05)
    with pytest.raises(ValueError):          # Wo % 8 != 0
        ops.conv2d(ops.nchw_to_nhwc(x.to(dev)), ops.prep_conv(cw, None, dev))


@pytest.mark.parametrize('tile', sorted(TILES))
@pytest.mark.parametrize('M,N,K', [(256, 81920, 320), (512, 40960, 2560)])
def test_pp_race_screen_with_weights_streaming_from_hbm(dev, tile, M, N, K):
    '''Race screen of the LDS ring.  The counted vmcnt of EVERY phase is what orders a piece's LDS-DMA before its ds_read; a missing wait
    does not show up in the refchecks above, whose weights sit in L2: the piece was issued several phases earlier and has landed.
    (Round 5: phase 3 did not wait, so the B0 piece of the next K-tile was read unretired -- about one launch in 25,000 of a forward
    went wrong, found as a sporadic graph-vs-plan mismatch of the 50-step loop.)  Here W is 50-200 MB and every tile reads its own rows of
    it exactly once, so every B piece is an HBM miss, and the short MFMA sections of the 64-row wave tiles leave the least time between
    issue and read: a build without phase 3's wait fails this test in every launch on tiles 32 / 33 (checked against such a build);
    the result must agree with torch and repeat bit for bit.'''
    from flexdiffuse_amd import ops
    bm, bn = TILES[tile]
    assert M % bm == 0 and N % bn == 0
    g = torch.Generator().manual_seed(17)
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
        for _ in range(60):
            ops.gemm(a, lw, out=out)
            bad += (out != ref).sum()
        torch.cuda.synchronize()
        assert int(bad) == 0, int(bad)
    finally:
        ops.FORCE_TILE = old
