'''Thin Python front of the C ABI: packs descriptors, allocates outputs with torch (device
memory + streams only) and launches the HIP kernels.  No arithmetic happens here.'''
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int32, c_int64, c_void_p
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import hip

ACT_NONE, ACT_SILU, ACT_QUICK_GELU, ACT_GELU, ACT_GEGLU = 0, 1, 2, 3, 4
FAMILY_GEMM, FAMILY_ATTENTION, FAMILY_GROUPNORM, FAMILY_OTHER = 0, 1, 2, 3


class fd_gemm_desc(ctypes.Structure):
    _fields_ = [('A', c_void_p), ('W', c_void_p), ('C', c_void_p), ('bias', c_void_p),
                ('bias2', c_void_p), ('residual', c_void_p),
                ('M', c_int32), ('N', c_int32), ('K', c_int32),
                ('lda', c_int32), ('ldw', c_int32), ('ldc', c_int32), ('ldr', c_int32),
                ('ld_bias2', c_int32), ('rows_per_sample', c_int32), ('act', c_int32),
                ('out_f32', c_int32), ('alpha', c_float),
                ('conv', c_int32), ('in_h', c_int32), ('in_w', c_int32), ('in_c', c_int32),
                ('out_h', c_int32), ('out_w', c_int32), ('kh', c_int32), ('kw', c_int32),
                ('stride', c_int32), ('pad_t', c_int32), ('pad_l', c_int32),
                ('upsample2x', c_int32), ('trans_out', c_int32), ('trans_ld', c_int32),
                ('trans_sample_stride', c_int64), ('batch', c_int32),
                ('batch_stride_a', c_int64), ('batch_stride_w', c_int64),
                ('batch_stride_c', c_int64), ('batch_stride_res', c_int64),
                ('tile', c_int32), ('split_k', c_int32), ('workspace', c_void_p),
                ('workspace_bytes', c_int64), ('ln_stats', c_void_p), ('ln_colsum', c_void_p),
                ('ln_stats_out', c_void_p), ('ln_eps', c_float), ('A2', c_void_p), ('lda2', c_int32), ('K2', c_int32),
                ('batch_stride_bias', c_int64),
                ('gn_out', c_void_p), ('gn_gamma', c_void_p), ('gn_beta', c_void_p),
                ('gn_groups', c_int32), ('gn_silu', c_int32), ('gn_eps', c_float), ('gn_skip_c', c_int32),
                ('gn_part_out', c_void_p), ('gn_part_chunks', c_int32), ('trans_n0', c_int32), ('C2', c_void_p),
                ('sk_sync', c_void_p), ('ln_stats_parts', c_int32), ('ln_stats_rows', c_int32), ('ln_fold_eps', c_float),
                ('residual_rows', c_int32)]


class fd_attention_desc(ctypes.Structure):
    _fields_ = [('Q', c_void_p), ('K', c_void_p), ('Vt', c_void_p), ('O', c_void_p),
                ('q_sample_stride', c_int64), ('k_sample_stride', c_int64),
                ('vt_sample_stride', c_int64), ('o_sample_stride', c_int64),
                ('ldq', c_int32), ('ldk', c_int32), ('ldvt', c_int32), ('ldo', c_int32),
                ('batch', c_int32), ('heads', c_int32), ('n_q', c_int32), ('n_k', c_int32),
                ('head_dim', c_int32), ('causal', c_int32), ('scale', c_float),
                ('q_prescaled', c_int32)]


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------ weights
@dataclass
class LinW:
    '''[N][K] fp16 weight (K contiguous) + fp32 bias padded to a multiple of 4.'''
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    K: int
    colsum: Optional[torch.Tensor] = None   # LayerNorm fold: fp32 [N] row sums of the gain-folded fp16 weights


@dataclass
class ConvW:
    w: torch.Tensor               # [Cout][kh*kw*Cin (padded)] fp16
    bias: Optional[torch.Tensor]
    cout: int
    cin: int
    kh: int
    kw: int
    im2col: bool                  # Cin < 64: explicit im2col + GEMM
    kpad: int
    k2: int = 0                   # columns of an appended 1x1 shortcut (prep_conv_shortcut)


def _bias(b: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if b is None:
        return None
    n = b.numel()
    out = torch.zeros(_round_up(n, 4), dtype=torch.float32, device=dev)
    out[:n] = b.to(dev, torch.float32)
    return out


def prep_linear(w: torch.Tensor, b: Optional[torch.Tensor], dev, k_pad: int = 0) -> LinW:
    N, K = w.shape
    kp = max(_round_up(K, 8), k_pad)
    wd = torch.zeros((N, kp), dtype=torch.float16, device=dev)
    wd[:, :K] = w.to(dev, torch.float16)
    return LinW(wd, _bias(b, dev), N, kp)


def prep_geglu(w: torch.Tensor, b: torch.Tensor, dev) -> LinW:
    '''ff.net.0.proj [8C][C]: interleave 16 value rows / 16 gate rows for the fused epilogue.'''
    n2, K = w.shape
    half = n2 // 2
    assert half % 16 == 0
    wv, wg = w[:half].reshape(half // 16, 16, K), w[half:].reshape(half // 16, 16, K)
    wi = torch.stack([wv, wg], dim=1).reshape(n2, K)
    bv, bg = b[:half].reshape(half // 16, 16), b[half:].reshape(half // 16, 16)
    bi = torch.stack([bv, bg], dim=1).reshape(n2)
    return prep_linear(wi, bi, dev)


def prep_linear_ln(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, dev,
                   geglu: bool = False) -> LinW:
    '''Linear layer that consumes LayerNorm(x; gamma, beta), with the LayerNorm FOLDED into the GEMM
    (fd_gemm_desc.ln_stats):  LN(x) W^T + b = rstd (x W'^T) - rstd mean colsum(W') + (b + W beta),
    W' = W diag(gamma).  Host-side weight preparation only (done once at model construction).'''
    w32, g32, be32 = w.float(), gamma.float(), beta.float()
    wf = (w32 * g32[None, :]).half()                      # the fp16 weights the MFMAs will see
    bias = w32 @ be32 + (b.float() if b is not None else 0.0)
    colsum = wf.float().sum(dim=1)                        # of the ROUNDED weights: exact cancellation of the mean
    if geglu:
        n2, K = wf.shape
        half = n2 // 2
        il = lambda t: torch.stack([t[:half].reshape(half // 16, 16, -1), t[half:].reshape(half // 16, 16, -1)],
                                   dim=1).reshape(n2, -1)
        wf, bias, colsum = il(wf), il(bias[:, None])[:, 0], il(colsum[:, None])[:, 0]
    lw = prep_linear(wf, bias, dev)
    lw.colsum = _bias(colsum, dev)
    return lw


def can_emit_row_stats(M: int, N: int, K: int = 320, ldc: int = 0, ldr: int = 0) -> int:
    '''How fd_gemm_f16 can write the LayerNorm statistics of its output rows itself (fd_gemm_desc.ln_stats_out):
    0 not at all (run ln_row_stats on the output), 1 the finished pairs (N == 320: one 256x320 tile spans the row),
    k > 1: k slabs of raw partial sums [k][M][2] that ln_finalize_stats combines.  The library answers, so its A/B
    switches (FD_GEMM_FAST_EPI=0, FD_GEMM_BIAS_LDS=0, FD_GEMM_NO_DMA) degrade to the separate statistics pass.'''
    if N != 320 and os.environ.get('FD_UNET_LN_PARTS', '1') == '0':
        return 0
    return int(hip.lib().fd_gemm_can_emit_row_stats(M, N, K, ldc or N, ldr))


def ln_finalize_stats(parts: torch.Tensor, N: int, eps: float = 1e-5) -> torch.Tensor:
    '''[k][M][2] partial sums of a stats-emitting GEMM -> [M][2] (rstd, -mean rstd).'''
    k, M, _ = parts.shape
    out = _empty((M, 2), torch.float32, parts)
    hip.call('fd_ln_finalize_stats_f32', parts.data_ptr(), out.data_ptr(), M, N, k, eps, hip.stream())
    return out


def ln_row_stats(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    '''fp16 [rows][C] -> fp32 [rows][2] = (rstd, -mean * rstd) per row (one read of x).'''
    rows, C = x.shape
    out = _empty((rows, 2), torch.float32, x)
    hip.call('fd_ln_row_stats_f16', x.data_ptr(), out.data_ptr(), rows, C, x.stride(0), eps, hip.stream())
    return out


def prep_conv(w: torch.Tensor, b: Optional[torch.Tensor], dev, cin_pad: int = 0) -> ConvW:
    '''[Cout][Cin][kh][kw] -> [Cout][kh][kw][Cin] fp16.  Cin < 64 -> im2col layout (K padded).'''
    cout, cin, kh, kw = w.shape
    cin_eff = max(cin, cin_pad)
    wp = torch.zeros((cout, kh, kw, cin_eff), dtype=torch.float32)
    wp[..., :cin] = w.permute(0, 2, 3, 1).float()
    k = kh * kw * cin_eff
    im2col = cin_eff % 64 != 0
    kpad = _round_up(k, 64) if im2col else k
    wd = torch.zeros((cout, kpad), dtype=torch.float16, device=dev)
    wd[:, :k] = wp.reshape(cout, k).to(dev, torch.float16)
    return ConvW(wd, _bias(b, dev), cout, cin_eff, kh, kw, im2col, kpad)


@dataclass
class NarrowConvW:
    w: torch.Tensor               # [Cout][3][3][4] fp16 (input channels >= Cin zero)
    bias: Optional[torch.Tensor]
    cout: int
    cin: int


# FD_UNET_CONV_IN=0: the UNet's conv_in through layout change + explicit im2col + GEMM (+ the fan-out copy) instead of fd_conv3x3_narrow_f16 (A/B)
CONV_IN_DIRECT = os.environ.get('FD_UNET_CONV_IN', '1') != '0'


def prep_conv_narrow(w: torch.Tensor, b: Optional[torch.Tensor], dev) -> Optional[NarrowConvW]:
    '''[Cout][Cin <= 4][3][3] -> the operand of fd_conv3x3_narrow_f16 (None when the layer has another shape).'''
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3) or cin > 4 or cout % 8 or cout > 2048:
        return None
    wp = torch.zeros((cout, 3, 3, 4), dtype=torch.float32)
    wp[..., :cin] = w.permute(0, 2, 3, 1).float()
    return NarrowConvW(wp.to(dev, torch.float16).contiguous(), _bias(b, dev), cout, cin)


def conv3x3_narrow(x: torch.Tensor, w: NarrowConvW, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
                   rep2: int = 0, scale: float = 1.0) -> Act:
    '''fp32 (B, Cin <= 4, H, W) -> 3x3 / stride 1 / pad 1 convolution as an fp16 NHWC Act in ONE launch (fd_conv3x3_narrow_f16);
    `out2` [rep2 * B*H*W][Cout] (row stride free) also receives `rep2` replicas of the output.'''
    B, C, H, W = x.shape
    assert C == w.cin and x.dtype == torch.float32
    x = x.contiguous()
    if out is None:
        out = _empty((B * H * W, w.cout), torch.float16, x)
    assert out.shape == (B * H * W, w.cout) and out.stride(1) == 1 and out.dtype == torch.float16
    if rep2:
        assert out2 is not None and out2.shape == (rep2 * B * H * W, w.cout) and out2.stride(1) == 1 and out2.dtype == torch.float16
    hip.call('fd_conv3x3_narrow_f16', x.data_ptr(), w.w.data_ptr(), _p(w.bias), out.data_ptr(), out.stride(0),
             _p(out2) if rep2 else None, out2.stride(0) if rep2 else 0, rep2, B, C, H, W, w.cout, scale, hip.stream())
    return Act(out, B, H, W)


def prep_conv_shortcut(w: torch.Tensor, b: Optional[torch.Tensor], ws: torch.Tensor, bs: Optional[torch.Tensor],
                       dev) -> ConvW:
    '''conv (kh x kw) weights [Cout][Cin][kh][kw] with a 1x1 shortcut [Cout][Cx] appended along K:
    [Cout][kh*kw*Cin | Cx] fp16, biases summed -- for conv2d(..., a2=x): the shortcut is accumulated by the
    convolution's own K loop (fd_gemm_desc.A2 / K2).  Cin and Cx multiples of 64.'''
    cout, cin, kh, kw = w.shape
    cx = ws.shape[1]
    assert cin % 64 == 0 and cx % 64 == 0 and ws.shape[0] == cout
    k = kh * kw * cin
    wd = torch.zeros((cout, k + cx), dtype=torch.float16, device=dev)
    wd[:, :k] = w.permute(0, 2, 3, 1).reshape(cout, k).to(dev, torch.float16)
    wd[:, k:] = ws.reshape(cout, cx).to(dev, torch.float16)
    bias = (b.float() if b is not None else 0.0) + (bs.float() if bs is not None else 0.0)
    return ConvW(wd, _bias(bias if torch.is_tensor(bias) else None, dev), cout, cin, kh, kw, False, k, cx)


def prep_conv_up_phases(w: torch.Tensor, b: Optional[torch.Tensor], dev) -> ConvW:
    '''3x3 weights of an Upsample2D conv [Cout][Cin][3][3] -> the four 2x2 parity filters of the phase-decomposed form
    (fd_gemm_desc.upsample2x == 2): out[2y+py][2x+px] = sum over a 2x2 window of the LOW-resolution input, with the
    3x3 taps that land on the same source pixel summed (fp32, rounded to fp16 once): rows {0 | 1+2} for py = 0,
    {0+1 | 2} for py = 1, same for columns.  Layout [4][Cout][2*2*Cin], K contiguous.'''
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cin % 64 == 0
    w32 = w.float()
    rows = {0: (w32[:, :, 0:1].sum(2), w32[:, :, 1:3].sum(2)), 1: (w32[:, :, 0:2].sum(2), w32[:, :, 2:3].sum(2))}   # [Cout][Cin][3]
    out = torch.zeros((4, cout, 2, 2, cin), dtype=torch.float32)
    for py in (0, 1):
        for a in (0, 1):
            r = rows[py][a]                                   # [Cout][Cin][kw]
            cols = {0: (r[:, :, 0:1].sum(2), r[:, :, 1:3].sum(2)), 1: (r[:, :, 0:2].sum(2), r[:, :, 2:3].sum(2))}
            for px in (0, 1):
                for bb in (0, 1):
                    out[py * 2 + px, :, a, bb, :] = cols[px][bb]
    wd = out.reshape(4, cout, 4 * cin).to(dev, torch.float16).contiguous()
    cw = ConvW(wd, _bias(b, dev), cout, cin, 2, 2, False, 4 * cin)
    return cw


def conv2d_up_phases(x: Act, w: ConvW, out: Optional[torch.Tensor] = None) -> Act:
    '''nearest-2x upsample + conv3x3 (diffusers Upsample2D) as four 2x2 parity convolutions of the low-resolution
    input in ONE launch (4/9 of the MACs of the fused-upsample form conv2d(..., up=True)); `w` from prep_conv_up_phases.'''
    assert x.C == w.cin and w.w.dim() == 3 and w.w.shape[0] == 4 and x.t.is_contiguous()
    M = x.B * x.H * x.W
    Ho, Wo = 2 * x.H, 2 * x.W
    if out is None:
        out = _empty((x.B * Ho * Wo, w.cout), torch.float16, x.t)
    assert out.shape == (x.B * Ho * Wo, w.cout) and out.stride(1) == 1
    d = fd_gemm_desc()
    d.A, d.W, d.C = x.t.data_ptr(), w.w.data_ptr(), out.data_ptr()
    d.bias = _p(w.bias)
    d.M, d.N, d.K = M, w.cout, w.kpad
    d.ldw, d.ldc = w.w.stride(1), out.stride(0)
    d.rows_per_sample, d.alpha = x.H * x.W, 1.0
    d.conv, d.in_h, d.in_w, d.in_c, d.out_h, d.out_w, d.kh, d.kw = 1, x.H, x.W, w.cin, x.H, x.W, 2, 2
    d.stride, d.pad_t, d.pad_l, d.upsample2x = 1, 1, 1, 2
    d.batch, d.batch_stride_w = 4, w.w.stride(0)
    _sched(d, x.t.device)
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return Act(out, x.B, Ho, Wo)


def up_phases_supported(rows_low: int, cout: int, cin: int) -> bool:
    '''Shapes for which conv2d_up_phases is used instead of conv2d(..., up=True): full 256-row tiles, UNet / VAE widths,
    and enough low-resolution rows that four parity slices fill the chip without split-K (UNet widths: from 1024 rows -- the 8x8 -> 16x16
    convolution of the bench forward: 68 us on the 3-stage 128x160 tile against 133 us for the fused-upsample form, tools/ab_up8.py -- the
    1024-row threshold holds for that measured class only, N >= 1280 and Cin >= 1280; narrower layers and the VAE widths, which have no
    128-row tile with the parity row map: from 4096 rows).'''
    return (os.environ.get('FD_UP_PHASES', '1') != '0' and rows_low % 256 == 0
            and rows_low >= int(os.environ.get('FD_UP_PHASES_MIN_ROWS', '1024' if cout % 160 == 0 and cout >= 1280 and cin >= 1280 else '4096'))
            and (cout % 160 == 0 or cout % 128 == 0) and cin % 64 == 0)


def f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(dev, torch.float32).contiguous()


def f16(t: torch.Tensor, dev) -> torch.Tensor:
    return t.to(dev, torch.float16).contiguous()


# --------------------------------------------------------------------------- activations
@dataclass
class Act:
    '''NHWC fp16 activation stored as a [B*H*W][C] matrix.'''
    t: torch.Tensor
    B: int
    H: int
    W: int

    @property
    def C(self) -> int:
        return self.t.shape[1]

    @property
    def HW(self) -> int:
        return self.H * self.W


def _empty(shape, dtype, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=like.device)


# ----------------------------------------------------------------------------------- gemm
_splitk_ws = {}
SPLITK_WS_BYTES = 96 << 20
GN_FINISH_FUSE = os.environ.get('FD_GN_FINISH_FUSE', '1') != '0'   # GroupNorm inside the split-K finish (A/B: 0 = separate launches)
_last_conv_gn_fused = False
gn_fused_launches = 0       # conv2d(..., gn=) calls that took the fused form so far (tests, tools)
FORCE_TILE = 0      # debugging / tuning knobs (0 = library cost model)
FORCE_SPLIT = 0
WS_SLOT = 0         # scratch-buffer set; work enqueued concurrently on another stream must use another slot


SPLITK_INLAUNCH = os.environ.get('FD_SPLITK_INLAUNCH', '0') == '1'    # EXPERIMENTAL in-launch split-K reduction (fd_gemm_desc.sk_sync); see DESIGN sec. 9 item 1b
_sk_sync = {}


def _sched(d: 'fd_gemm_desc', dev: torch.device):
    key = (dev.index if dev.index is not None else 0, WS_SLOT)
    ws = _splitk_ws.get(key)
    if ws is None:
        ws = torch.empty(SPLITK_WS_BYTES, dtype=torch.uint8, device=dev)
        _splitk_ws[key] = ws
    d.workspace, d.workspace_bytes = ws.data_ptr(), SPLITK_WS_BYTES
    d.tile, d.split_k = FORCE_TILE, FORCE_SPLIT


def _inlaunch_request(d: 'fd_gemm_desc', dev: torch.device, tile: int, split: int) -> bool:
    '''Experimental: hand the launch the tile counters of the in-launch split-K reduction when it would run on a 320-wide ping-pong tile with
    every workgroup resident (tiles x split <= 256).'''
    if not SPLITK_INLAUNCH or split <= 1 or tile not in (30, 32):
        return False
    bm = 256 if tile == 30 else 128
    if d.N % 320 or d.M % bm or (d.M // bm) * (d.N // 320) * split > 256 or bm % split:
        return False
    key = (dev.index if dev.index is not None else 0, WS_SLOT)
    t = _sk_sync.get(key)
    if t is None:
        t = torch.zeros(8192, dtype=torch.int32, device=dev)
        _sk_sync[key] = t
    d.sk_sync = t.data_ptr()
    return True


LN_PARTS = os.environ.get('FD_UNET_LN_FINALIZE', '0') != '1'    # LayerNorm partial sums finalised by the CONSUMER GEMM's tiles (A/B: FD_UNET_LN_FINALIZE=1 = the finalise launch)


def _set_ln_stats(d: 'fd_gemm_desc', ln_stats: torch.Tensor, w: LinW, M: int, eps: float = 1e-5):
    '''ln_stats [M][2]: finished (rstd, -mean rstd) pairs; [k][M][2] (k in 2, 4, 8): the raw partial sums a producer GEMM wrote through
    ln_stats_out -- the consumer's tiles finalise their own rows (fd_gemm_desc.ln_stats_parts), no fd_ln_finalize_stats_f32 launch.'''
    assert w.colsum is not None and ln_stats.dtype == torch.float32 and ln_stats.is_contiguous()
    if ln_stats.dim() == 3:
        assert ln_stats.shape[1:] == (M, 2) and ln_stats.shape[0] in (2, 4, 8), tuple(ln_stats.shape)
        d.ln_stats_parts, d.ln_stats_rows, d.ln_fold_eps = ln_stats.shape[0], M, eps
    else:
        assert ln_stats.shape == (M, 2)
    d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), w.colsum.data_ptr()


def gemm(a: torch.Tensor, w: LinW, *, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
         bias2: Optional[torch.Tensor] = None, ld_bias2: int = 0, rows_per_sample: int = 0,
         out_f32: bool = False, out: Optional[torch.Tensor] = None, alpha: float = 1.0,
         use_bias: bool = True, ln_stats: Optional[torch.Tensor] = None,
         ln_stats_out: Optional[torch.Tensor] = None, ln_eps: float = 1e-5,
         a2: Optional[torch.Tensor] = None, gn_parts: int = 0):
    '''a [M][K] fp16 @ w[N][K]^T with fused epilogue -> [M][N] (GEGLU: [M][N/2]).  `a2` [M][K2]: a second operand
    accumulated by the same K loop against w[:, K:K+K2] (fd_gemm_desc.A2 / K2; `w` holds K + K2 columns).
    `gn_parts` = G > 0 (needs rows_per_sample): -> (out, GNParts or None): the GroupNorm partial sums of the output over G groups
    from the launch's own epilogue where the library can (fd_gemm_desc.gn_part_out), for the GroupNorm that reads `out` next.'''
    M, K = a.shape
    K2 = 0 if a2 is None else a2.shape[1]
    assert a.dtype == torch.float16 and a.stride(1) == 1 and K + K2 == w.K, (a.shape, K2, w.K)
    n_out = w.N // 2 if act == ACT_GEGLU else w.N
    if out is None:
        out = _empty((M, _round_up(n_out, 4)), torch.float32 if out_f32 else torch.float16, a)
    d = fd_gemm_desc()
    d.A, d.W, d.C = a.data_ptr(), w.w.data_ptr(), out.data_ptr()
    d.bias = _p(w.bias) if use_bias else None
    d.bias2 = _p(bias2)
    d.residual = _p(residual)
    d.M, d.N, d.K = M, w.N, K
    d.lda, d.ldw, d.ldc = a.stride(0), w.w.stride(0), out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    if residual is not None and residual.shape[0] != M:
        # the residual of replicated rows (the CFG fan-out of a shared prefix): row m adds residual row m % rows (fd_gemm_desc.residual_rows)
        assert M % residual.shape[0] == 0, (M, residual.shape)
        d.residual_rows = residual.shape[0]
    d.ld_bias2 = ld_bias2
    d.rows_per_sample = rows_per_sample
    d.act, d.out_f32, d.alpha = act, int(out_f32), alpha
    d.batch = 1
    if a2 is not None:
        assert a2.shape[0] == M and a2.stride(1) == 1 and a2.dtype == torch.float16
        d.A2, d.lda2, d.K2 = a2.data_ptr(), a2.stride(0), K2
    if ln_stats is not None:      # `a` holds the un-normalised rows, `w` comes from prep_linear_ln
        _set_ln_stats(d, ln_stats, w, M)
    if ln_stats_out is not None:  # the GEMM also writes the LayerNorm statistics of its output rows
        assert ln_stats_out.shape[-2:] == (M, 2) and ln_stats_out.dtype == torch.float32 and ln_stats_out.is_contiguous()
        d.ln_stats_out, d.ln_eps = ln_stats_out.data_ptr(), ln_eps
    _sched(d, a.device)
    if gn_parts:
        parts = _gn_parts_request(d, M // rows_per_sample, gn_parts, a) if rows_per_sample > 0 and M % rows_per_sample == 0 else None
        hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
        return out, parts
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return out


def gemm_vt(a: torch.Tensor, w: LinW, B: int, rows_per_sample: int, ld: int,
            out: Optional[torch.Tensor] = None, ln_stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    '''V projection with a transposed store: returns V^T [B][N][ld] fp16 (pad columns zero).'''
    M, K = a.shape
    assert M == B * rows_per_sample and K == w.K
    if out is not None:
        assert out.shape == (B, w.N, ld) and out.is_contiguous()
    elif ld == rows_per_sample:
        out = _empty((B, w.N, ld), torch.float16, a)
    else:
        out = torch.zeros((B, w.N, ld), dtype=torch.float16, device=a.device)
    d = fd_gemm_desc()
    d.A, d.W, d.C = a.data_ptr(), w.w.data_ptr(), out.data_ptr()
    d.bias = _p(w.bias)
    d.M, d.N, d.K = M, w.N, K
    d.lda, d.ldw, d.ldc = a.stride(0), w.w.stride(0), 0
    d.rows_per_sample = rows_per_sample
    d.trans_out, d.trans_ld, d.trans_sample_stride = 1, ld, w.N * ld
    d.alpha, d.batch = 1.0, 1
    if ln_stats is not None:
        _set_ln_stats(d, ln_stats, w, M)
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return out


QKV_MERGE = os.environ.get('FD_UNET_QKV', '1') != '0'     # self-attention q | k | v as ONE launch with a transposed tail (A/B: 0 = two launches)


def qkv_merge_supported(M: int, C: int, rows_per_sample: int) -> bool:
    '''Shapes gemm_qkv covers (fd_gemm_desc.trans_n0): 128-row tiles, 160-column tiles, whole 32-row blocks per sample, unpadded V^T rows.'''
    return QKV_MERGE and M % 128 == 0 and C % 160 == 0 and rows_per_sample % 32 == 0 and M % rows_per_sample == 0


def gemm_qkv(a: torch.Tensor, w: LinW, B: int, rows_per_sample: int, ln_stats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    '''LayerNorm-fold projection with the weights stacked [q | k | v] (3C rows, prep_linear_ln) in ONE launch: -> (q|k [M][2C] fp16
    row-major, V^T [B][C][rows_per_sample] fp16): the last C columns of the output are stored transposed (fd_gemm_desc.trans_n0 / C2).
    The hidden states are read once; same bits as gemm(..., wqk) + gemm_vt(..., wv).'''
    M, K = a.shape
    C = w.N // 3
    assert w.N == 3 * C and K == w.K and M == B * rows_per_sample and w.colsum is not None
    qk = _empty((M, 2 * C), torch.float16, a)
    vt = _empty((B, C, rows_per_sample), torch.float16, a)
    d = fd_gemm_desc()
    d.A, d.W, d.C, d.C2 = a.data_ptr(), w.w.data_ptr(), qk.data_ptr(), vt.data_ptr()
    d.bias = _p(w.bias)
    d.M, d.N, d.K = M, w.N, K
    d.lda, d.ldw, d.ldc = a.stride(0), w.w.stride(0), 2 * C
    d.rows_per_sample, d.alpha, d.batch = rows_per_sample, 1.0, 1
    d.trans_n0, d.trans_ld, d.trans_sample_stride = 2 * C, rows_per_sample, C * rows_per_sample
    _set_ln_stats(d, ln_stats, w, M)
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return qk, vt


def bgemm(a: torch.Tensor, w: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    '''Batched a[b] [M][K] @ w[b][N][K]^T -> [b][M][N] fp16 (VAE single-head attention).'''
    Bz, M, K = a.shape
    N = w.shape[1]
    out = _empty((Bz, M, N), torch.float16, a)
    d = fd_gemm_desc()
    d.A, d.W, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldw, d.ldc = a.stride(1), w.stride(1), N
    d.alpha, d.batch = alpha, Bz
    d.batch_stride_a, d.batch_stride_w, d.batch_stride_c = a.stride(0), w.stride(0), M * N
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return out


@dataclass
class GNSpec:
    '''A GroupNorm(+SiLU) that consumes a convolution's output (conv2d(..., gn=...)).'''
    gamma: torch.Tensor
    beta: torch.Tensor
    G: int
    eps: float
    silu: bool


@dataclass
class GNParts:
    '''GroupNorm partial sums [B][chunks][G][2] written by the producer of a tensor (fd_gemm_desc.gn_part_out).'''
    t: torch.Tensor
    chunks: int
    G: int


GN_PARTS = os.environ.get('FD_GN_PARTS', '1') != '0'     # GroupNorm statistics from the producing launch's epilogue (A/B: 0 = statistics pass)


def _gn_parts_request(d: 'fd_gemm_desc', B: int, G: int, like: torch.Tensor) -> Optional[GNParts]:
    '''Asks the library whether the launch described by `d` can write the GroupNorm partial sums of its output over G groups;
    if so allocates them, sets d.gn_part_out / d.gn_groups and returns the GNParts the consumer takes.'''
    if not GN_PARTS or d.act != ACT_NONE or d.out_f32:
        return None
    d.gn_groups = G
    chunks = hip.lib().fd_gemm_gn_parts_chunks(ctypes.byref(d))
    if chunks <= 0:
        return None
    t = _empty((B, chunks, G, 2), torch.float32, like)
    d.gn_part_out, d.gn_part_chunks = t.data_ptr(), chunks
    return GNParts(t, chunks, G)


def conv2d(x: Act, w: ConvW, *, stride: int = 1, pad: Tuple[int, int] = (1, 1), up: bool = False,
           out_hw: Optional[Tuple[int, int]] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, bias2: Optional[torch.Tensor] = None,
           ld_bias2: int = 0, out_f32: bool = False, out: Optional[torch.Tensor] = None,
           a2: Optional[torch.Tensor] = None, gn: Optional[GNSpec] = None, keep: bool = True, gn_parts: int = 0):
    '''NHWC conv (kh x kw) as implicit GEMM; `up` fuses a nearest-2x upsample of the input.  `a2` [M][Cx]
    (row stride free): the rows of the 1x1 shortcut appended to `w` by prep_conv_shortcut.
    `gn`: also return GroupNorm(+SiLU) of the output -> (Act or None, Act): where the library splits the convolution over K
    (the 16x16 / 8x8 UNet levels) the pass that sums the partial slabs normalises them too (fd_gemm_desc.gn_out: one launch
    less; with keep=False the un-normalised output is never written and the first element is None); everywhere else the
    convolution is followed by the groupnorm launch -- at the 64x64 level (one 320-wide tile spans the row) without its statistics
    pass: the convolution's epilogue writes the partial sums (fd_gemm_desc.gn_part_out).  The fused form gives the bits of the
    separate launches; the partial-sum form differs from them in the summation order of the statistics only.
    `gn_parts` = G > 0 (without `gn`): -> (Act, GNParts or None): just the partial sums over G groups next to the output, for a
    consumer that needs statistics only (gn_fold_linear) or normalises later.'''
    assert x.C == w.cin, (x.C, w.cin)
    Hv, Wv = (x.H * 2, x.W * 2) if up else (x.H, x.W)
    if out_hw is None:
        Ho = (Hv + 2 * pad[0] - w.kh) // stride + 1
        Wo = (Wv + 2 * pad[1] - w.kw) // stride + 1
    else:
        Ho, Wo = out_hw
    M = x.B * Ho * Wo
    if out is None:
        out = _empty((M, _round_up(w.cout, 4)), torch.float32 if out_f32 else torch.float16, x.t)
    assert out.shape == (M, _round_up(w.cout, 4)) and out.stride(1) == 1
    # the implicit-GEMM loaders take a pixel stride: x may be a column slice of a wider matrix (a skip tensor in its concat buffer)
    assert x.t.stride(1) == 1 and (x.t.is_contiguous() or (not w.im2col and x.t.stride(0) % 8 == 0 and x.t.data_ptr() % 16 == 0)), \
        'conv input must be an NHWC matrix with unit channel stride (contiguous for the explicit-im2col path)'
    d = fd_gemm_desc()
    d.W, d.C = w.w.data_ptr(), out.data_ptr()
    d.bias, d.bias2, d.residual = _p(w.bias), _p(bias2), _p(residual)
    d.M, d.N, d.K = M, w.cout, w.kpad
    d.ldw, d.ldc = w.w.stride(0), out.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.ld_bias2, d.rows_per_sample = ld_bias2, Ho * Wo
    d.act, d.out_f32, d.alpha, d.batch = act, int(out_f32), 1.0, 1
    if w.im2col:
        if up:
            raise ValueError(f'conv2d: the fused nearest-2x upsample needs Cin % 64 == 0 (got {w.cin}): the explicit-im2col '
                             'path of narrow inputs has no upsample form')
        cols = _empty((M, w.kpad), torch.float16, x.t)
        hip.call('fd_im2col_f16', x.t.data_ptr(), cols.data_ptr(), x.B, x.H, x.W, w.cin, Ho, Wo,
                 w.kh, w.kw, stride, pad[0], pad[1], w.kpad, hip.stream())
        d.A, d.lda = cols.data_ptr(), w.kpad
    else:
        d.A, d.lda = x.t.data_ptr(), x.t.stride(0)
        d.conv, d.in_h, d.in_w, d.in_c = 1, x.H, x.W, w.cin
        d.out_h, d.out_w, d.kh, d.kw = Ho, Wo, w.kh, w.kw
        d.stride, d.pad_t, d.pad_l, d.upsample2x = stride, pad[0], pad[1], int(up)
    k2 = w.k2
    if k2:
        assert a2 is not None and not w.im2col and a2.shape == (M, k2) and a2.stride(1) == 1 and a2.dtype == torch.float16
        d.A2, d.lda2, d.K2 = a2.data_ptr(), a2.stride(0), k2
    else:
        assert a2 is None
    _sched(d, x.t.device)
    if gn is not None:
        lib = hip.lib()
        tile, split = c_int32(0), c_int32(0)
        hip.check(lib.fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile), ctypes.byref(split)), 'fd_gemm_plan')
        inl = _inlaunch_request(d, x.t.device, tile.value, split.value)
        fuse = (not inl and GN_FINISH_FUSE and split.value > 1 and act == ACT_NONE and not out_f32 and w.cout % 8 == 0 and out.stride(0) % 8 == 0
                and (residual is None or (residual.stride(0) % 8 == 0 and residual.data_ptr() % 16 == 0))
                and (bias2 is None or (ld_bias2 % 4 == 0 and bias2.data_ptr() % 16 == 0))
                and lib.fd_gemm_can_fuse_groupnorm(M, w.cout, Ho * Wo, gn.G, split.value))
        global _last_conv_gn_fused, gn_fused_launches
        _last_conv_gn_fused = bool(fuse)      # (read by the tests: which form the last conv2d(..., gn=) took)
        gn_fused_launches += int(bool(fuse))
        if not fuse:
            parts = _gn_parts_request(d, x.B, gn.G, x.t)
            hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
            res = Act(out, x.B, Ho, Wo)
            return res, groupnorm(res, gn.gamma, gn.beta, gn.G, gn.eps, gn.silu, parts=parts)
        y = _empty((M, w.cout), torch.float16, x.t)
        d.gn_out, d.gn_gamma, d.gn_beta = y.data_ptr(), gn.gamma.data_ptr(), gn.beta.data_ptr()
        d.gn_groups, d.gn_silu, d.gn_eps, d.gn_skip_c = gn.G, int(gn.silu), gn.eps, int(not keep)
        hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
        return (Act(out, x.B, Ho, Wo) if keep else None), Act(y, x.B, Ho, Wo)
    if SPLITK_INLAUNCH and act == ACT_NONE and not out_f32:      # (experimental; conv2d(..., gn=) has asked above)
        tile, split = c_int32(0), c_int32(0)
        hip.check(hip.lib().fd_gemm_plan(ctypes.byref(d), ctypes.byref(tile), ctypes.byref(split)), 'fd_gemm_plan')
        _inlaunch_request(d, x.t.device, tile.value, split.value)
    if gn_parts:
        parts = _gn_parts_request(d, x.B, gn_parts, x.t)
        hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
        return Act(out, x.B, Ho, Wo), parts
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return Act(out, x.B, Ho, Wo)


# ------------------------------------------------------------------------------ attention
QK_LOG2E = 1.4426950408889634


def attention_accepts_prescaled(head_dim: int) -> bool:
    '''True when fd_attention_f16 takes q_prescaled for this head_dim (the VALU-lean 8-wave
    kernels; FD_ATTN_QT1=0/1 selects the older kernels for A/B runs).'''
    return os.environ.get('FD_ATTN_QT1', '2') == '2' and (head_dim <= 80 or head_dim > 128)


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, B: int, heads: int, n_q: int,
              n_k: int, head_dim: int, causal: bool = False, q_prescaled: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    '''q [B*n_q][C], k [B*n_k][C], vt [B][C][ldvt] -> [B*n_q][C] fp16 (into `out` if given).
    q_prescaled: q already carries head_dim^-0.5 * QK_LOG2E (the UNet folds it into the q
    projection weights).'''
    if out is None:
        out = _empty((B * n_q, heads * head_dim), torch.float16, q)
    assert out.shape == (B * n_q, heads * head_dim) and out.dtype == torch.float16
    d = fd_attention_desc()
    d.Q, d.K, d.Vt, d.O = q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr()
    d.ldq, d.ldk, d.ldvt, d.ldo = q.stride(0), k.stride(0), vt.stride(1), out.stride(0)
    d.q_sample_stride, d.k_sample_stride = n_q * q.stride(0), n_k * k.stride(0)
    d.vt_sample_stride, d.o_sample_stride = vt.stride(0), n_q * out.stride(0)
    d.batch, d.heads, d.n_q, d.n_k, d.head_dim = B, heads, n_q, n_k, head_dim
    d.causal, d.scale, d.q_prescaled = int(causal), 0.0, int(q_prescaled)
    hip.call('fd_attention_f16', ctypes.byref(d), hip.stream())
    return out


class fd_xattn_desc(ctypes.Structure):
    _fields_ = [('x', c_void_p), ('wq', c_void_p), ('bias', c_void_p), ('ln_colsum', c_void_p),
                ('ln_stats', c_void_p), ('k_image', c_void_p), ('v_image', c_void_p), ('out', c_void_p),
                ('M', c_int32), ('ldx', c_int32), ('ldw', c_int32), ('ldo', c_int32),
                ('rows_per_sample', c_int32), ('n_rep', c_int32), ('n_keys', c_int32), ('heads', c_int32),
                ('head_dim', c_int32), ('ln_stats_parts', c_int32), ('ln_fold_eps', c_float)]


def xattn_row_tile(head_dim: int) -> int:
    '''Rows per workgroup tile of fd_xattn_q_f16 (rows per sample must be a multiple): 256 at head dim 40, 128 at 80.'''
    return 256 if head_dim == 40 else 128


def xattn_supported(heads: int, head_dim: int, n_keys: int, rows_per_sample: int) -> bool:
    '''True when fd_xattn_q_f16 (fused LayerNorm-fold q projection + cross-attention) covers the shape.'''
    return (os.environ.get('FD_UNET_XATTN', '1') != '0' and 64 < n_keys <= 80
            and hip.lib().fd_xattn_image_bytes(heads, head_dim) > 0 and rows_per_sample % xattn_row_tile(head_dim) == 0)


def xattn_pack_kv(k: torch.Tensor, vt: torch.Tensor, samples: int, n_keys: int, heads: int, head_dim: int,
                  out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    '''Context K [samples*n_keys][C] / V^T [samples][C][ldv] -> per-sample (K image, V^T image) in MFMA
    fragment order (once per context; the images are what fd_xattn_q_f16 stages into LDS).'''
    nbytes = hip.lib().fd_xattn_image_bytes(heads, head_dim)
    assert nbytes > 0
    if out is None:
        out = (torch.empty((samples, nbytes), dtype=torch.uint8, device=k.device),
               torch.empty((samples, nbytes), dtype=torch.uint8, device=k.device))
    hip.call('fd_xattn_pack_kv_f16', k.data_ptr(), vt.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), samples,
             n_keys, heads, head_dim, k.stride(0), vt.stride(1), n_keys * k.stride(0), vt.stride(0), hip.stream())
    return out


def xattn_q(x: torch.Tensor, w: LinW, ln_stats: torch.Tensor, images: Tuple[torch.Tensor, torch.Tensor],
            rows_per_sample: int, n_keys: int, heads: int, head_dim: int, n_rep: int = 1,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    '''softmax(LN-fold(x) Wq^T  K^T) V over the packed context for `n_rep` context replicas sharing the
    queries: x [M][C] un-normalised fp16 -> [n_rep*M][C] fp16.  `w` from prep_linear_ln with the softmax
    scale * log2(e) folded in.'''
    M, C = x.shape
    assert w.colsum is not None and w.bias is not None and x.stride(1) == 1 and ln_stats.dtype == torch.float32 and ln_stats.is_contiguous()
    # [M][2]: finished pairs; [k][M][2]: the producer's partial slabs, finalised by the kernel's tiles (fd_xattn_desc.ln_stats_parts)
    assert ln_stats.shape == (M, 2) or (ln_stats.dim() == 3 and ln_stats.shape[0] in (2, 4, 8) and ln_stats.shape[1:] == (M, 2)), tuple(ln_stats.shape)
    assert images[0].shape[0] == n_rep * (M // rows_per_sample)
    if out is None:
        out = _empty((n_rep * M, C), torch.float16, x)
    d = fd_xattn_desc()
    d.x, d.wq, d.bias, d.ln_colsum = x.data_ptr(), w.w.data_ptr(), w.bias.data_ptr(), w.colsum.data_ptr()
    d.ln_stats, d.k_image, d.v_image, d.out = ln_stats.data_ptr(), images[0].data_ptr(), images[1].data_ptr(), out.data_ptr()
    d.M, d.ldx, d.ldw, d.ldo = M, x.stride(0), w.w.stride(0), out.stride(0)
    d.rows_per_sample, d.n_rep, d.n_keys, d.heads, d.head_dim = rows_per_sample, n_rep, n_keys, heads, head_dim
    if ln_stats.dim() == 3:
        d.ln_stats_parts, d.ln_fold_eps = ln_stats.shape[0], 1e-5
    hip.call('fd_xattn_q_f16', ctypes.byref(d), hip.stream())
    return out


# ---------------------------------------------------------------------------------- norms
_gn_ws = {}
_gn_ws_retired = []    # outgrown scratch buffers: never freed (see _gn_workspace)


def _gn_workspace(B: int, G: int, dev) -> torch.Tensor:
    '''Per-(device, slot) GroupNorm scratch, grow-only.  A recorded launch plan (hip.Plan) replays raw
    device addresses, this one included, and the plan cannot see a later reallocation -- so an outgrown
    buffer is RETIRED, not freed: its address stays valid (and private to GroupNorm launches) for every
    plan that recorded it.  A buffer is a few KB (B x G partial sums); growth happens a handful of times
    per process (first call per batch size).'''
    n = hip.lib().fd_groupnorm_workspace_floats(B, G)
    key = (dev.index if dev.index is not None else 0, WS_SLOT)
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < n:
        if ws is not None:
            _gn_ws_retired.append(ws)
        ws = torch.empty(n, dtype=torch.float32, device=dev)
        _gn_ws[key] = ws
    return ws


def groupnorm(x: Act, gamma: torch.Tensor, beta: torch.Tensor, G: int, eps: float,
              silu: bool, parts: Optional['GNParts'] = None) -> Act:
    '''`parts`: the partial sums the producer of x wrote (conv2d(..., gn_parts=) / gn=): only the apply pass runs.'''
    # x may be a column slice of a wider matrix (a skip tensor living in its concat buffer)
    assert x.t.stride(1) == 1
    out = _empty(tuple(x.t.shape), torch.float16, x.t)
    if parts is not None:
        assert parts.G == G and parts.t.shape == (x.B, parts.chunks, G, 2)
        hip.call('fd_groupnorm_apply_parts_f16', x.t.data_ptr(), x.t.stride(0), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                 parts.t.data_ptr(), parts.chunks, x.B, x.HW, x.C, G, eps, int(silu), hip.stream())
        return Act(out, x.B, x.H, x.W)
    ws = _gn_workspace(x.B, G, x.t.device)
    hip.call('fd_groupnorm_nhwc_ld_f16', x.t.data_ptr(), x.t.stride(0), out.data_ptr(), gamma.data_ptr(),
             beta.data_ptr(), ws.data_ptr(), x.B, x.HW, x.C, G, eps, int(silu), hip.stream())
    return Act(out, x.B, x.H, x.W)


@dataclass
class GNFold:
    '''Model constants of a GroupNorm folded into the linear layer behind it (prep_gn_fold).'''
    wg: torch.Tensor      # fp16 [N][C] = W diag(gamma)
    biasf: torch.Tensor   # fp32 [N] = bias + W beta
    G: int
    eps: float
    N: int
    C: int


def prep_gn_fold(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, G: int, eps: float,
                 dev) -> GNFold:
    '''proj_in(GroupNorm(x)) as a per-sample linear layer on x itself (fd_groupnorm_fold_linear_f16).'''
    w = w.reshape(w.shape[0], -1).float()
    N, C = w.shape
    wg = (w * gamma.float()[None, :]).half()
    biasf = (b.float() if b is not None else torch.zeros(N)) + w @ beta.float()
    return GNFold(wg.contiguous().to(dev), biasf.contiguous().to(dev), G, eps, N, C)


GN_FOLD_MAX_C, GN_FOLD_MAX_G = 48 * 1024 // (16 * 2), 64   # fd_groupnorm_fold_linear_f16: fp16 [16][C] weight tile <= 48 KiB of LDS; gsum_s[16][64]


def gn_fold_supported(B: int, HW: int, N: int, C: Optional[int] = None, G: int = 32) -> bool:
    '''The fold pays where the per-sample weights (B x N x C) are smaller than the activation (B x HW x C) they stand
    in for, and the consumer GEMM needs whole row tiles per sample.  `C` (input channels of the folded layer, default N:
    the transformer's proj_in is square) and `G` must fit the fold kernel's LDS tiles (C <= 1536, G <= 64): a wider
    layer falls back to the unfused GroupNorm here instead of failing with FD_ESHAPE inside the forward.'''
    mode = os.environ.get('FD_UNET_GN_FOLD', '1')    # 0: never; 1: where the map is >= 8x wider than the layer; 2: wherever N < HW
    C = N if C is None else C
    return (mode != '0' and HW % 256 == 0 and B > 1 and (N * 8 <= HW if mode == '1' else N < HW)
            and C <= GN_FOLD_MAX_C and G <= GN_FOLD_MAX_G)


def gn_fold_linear(x: Act, gf: GNFold, parts: Optional['GNParts'] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    '''One statistics pass over x (none when the producer of x wrote `parts`), then sample b's scaled weights [B][N][C] fp16 and
    bias [B][N] fp32.'''
    assert x.t.stride(1) == 1 and x.C == gf.C
    wb = _empty((x.B, gf.N, gf.C), torch.float16, x.t)
    bb = _empty((x.B, gf.N), torch.float32, x.t)
    if parts is not None:
        assert parts.G == gf.G and parts.t.shape == (x.B, parts.chunks, gf.G, 2)
        hip.call('fd_groupnorm_fold_linear_parts_f16', parts.t.data_ptr(), parts.chunks, x.B, x.HW, gf.C, gf.G, gf.eps,
                 gf.wg.data_ptr(), gf.biasf.data_ptr(), gf.N, wb.data_ptr(), bb.data_ptr(), hip.stream())
        return wb, bb
    ws = _gn_workspace(x.B, gf.G, x.t.device)
    hip.call('fd_groupnorm_fold_linear_f16', x.t.data_ptr(), x.t.stride(0), ws.data_ptr(), x.B, x.HW, gf.C, gf.G, gf.eps,
             gf.wg.data_ptr(), gf.biasf.data_ptr(), gf.N, wb.data_ptr(), bb.data_ptr(), hip.stream())
    return wb, bb


def gemm_per_sample(a: torch.Tensor, wb: torch.Tensor, bb: torch.Tensor, B: int, HW: int,
                    ln_stats_out: Optional[torch.Tensor] = None, ln_eps: float = 1e-5) -> torch.Tensor:
    '''a [B*HW][K] (row stride lda) @ wb[b][N][K]^T + bb[b] for the HW rows of sample b -> [B*HW][N] fp16: ONE launch
    with batch = B, per-batch weights and bias (fd_gemm_desc.batch_stride_w / batch_stride_bias).'''
    M, K = a.shape
    _, N, Kw = wb.shape
    assert M == B * HW and K == Kw and a.stride(1) == 1 and wb.is_contiguous() and bb.is_contiguous()
    out = _empty((M, N), torch.float16, a)
    d = fd_gemm_desc()
    d.A, d.W, d.C, d.bias = a.data_ptr(), wb.data_ptr(), out.data_ptr(), bb.data_ptr()
    d.M, d.N, d.K = HW, N, K
    d.lda, d.ldw, d.ldc = a.stride(0), K, N
    d.alpha, d.batch = 1.0, B
    d.batch_stride_a, d.batch_stride_w, d.batch_stride_c, d.batch_stride_bias = HW * a.stride(0), N * K, HW * N, N
    if ln_stats_out is not None:
        assert ln_stats_out.shape[-2:] == (M, 2) and ln_stats_out.dtype == torch.float32 and ln_stats_out.is_contiguous()
        d.ln_stats_out, d.ln_eps = ln_stats_out.data_ptr(), ln_eps
    _sched(d, a.device)
    hip.call('fd_gemm_f16', ctypes.byref(d), hip.stream())
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
              out_f32: bool = False) -> torch.Tensor:
    rows, C = x.shape
    out = _empty((rows, C), torch.float32 if out_f32 else torch.float16, x)
    hip.call('fd_layernorm_f16', x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
             rows, C, x.stride(0), out.stride(0), eps, int(out_f32), hip.stream())
    return out


def softmax_rows_(x: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    rows = x.numel() // x.shape[-1]
    hip.call('fd_softmax_rows_f16', x.data_ptr(), rows, x.shape[-1], x.stride(-2), scale,
             hip.stream())
    return x


# ----------------------------------------------------------------------------- elementwise
def nchw_to_nhwc(x: torch.Tensor, rep: int = 1, c_pad: int = 0, scale: float = 1.0) -> Act:
    '''fp32 (B,C,H,W) -> NHWC fp16 Act with the batch replicated `rep` times.'''
    B, C, H, W = x.shape
    x = x.to(torch.float32).contiguous()
    cp = max(C, c_pad)
    out = _empty((rep * B * H * W, cp), torch.float16, x)
    hip.call('fd_nchw_f32_to_nhwc_f16', x.data_ptr(), out.data_ptr(), B, C, H * W, rep, cp, scale,
             hip.stream())
    return Act(out, rep * B, H, W)


def nhwc_to_nchw(x: torch.Tensor, B: int, C: int, H: int, W: int, a: float = 1.0, b: float = 0.0,
                 clamp01: bool = False) -> torch.Tensor:
    '''fp32 [B*H*W][ld] -> fp32 (B,C,H,W), y = x*a + b.'''
    assert x.dtype == torch.float32
    out = _empty((B, C, H, W), torch.float32, x)
    hip.call('fd_nhwc_f32_to_nchw_f32', x.data_ptr(), out.data_ptr(), B, C, H * W, x.stride(0), a,
             b, int(clamp01), hip.stream())
    return out


def copy_rows(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    '''dst[r][:] = src[r][:] for fp16 matrices with arbitrary row strides (a library launch, so a
    launch plan records it -- torch's copy_ would not be replayed).'''
    assert dst.shape == src.shape and dst.dim() == 2 and dst.dtype == src.dtype == torch.float16
    assert dst.stride(1) == 1 and src.stride(1) == 1
    hip.call('fd_copy2d_f16', src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), src.shape[0],
             src.shape[1], hip.stream())
    return dst


def contiguous_rows(x: torch.Tensor) -> torch.Tensor:
    '''x itself when its rows are dense, else a dense copy made by a library launch.'''
    if x.is_contiguous():
        return x
    return copy_rows(_empty(tuple(x.shape), torch.float16, x), x)


# FD_UNET_RES_WRAP=0: the CFG fan-out materialises the residuals' replicas (fd_repeat_rows_f16) instead of wrapping the residual rows (A/B)
RES_WRAP = os.environ.get('FD_UNET_RES_WRAP', '1') != '0'


def residual_wrap_supported(rows: int, rep: int) -> bool:
    '''True when fd_gemm_f16 can add a residual of `rows` rows to `rep * rows` output rows modulo `rows` (fd_gemm_desc.residual_rows) whatever tile
    the rule picks: the wrap must fall on a tile boundary -- 256 rows covers every tile but the 288-row one, which needs full tiles of M.'''
    return RES_WRAP and rep > 1 and rows % 256 == 0 and (rows % 288 == 0 or (rep * rows) % 288 != 0)


def repeat_rows(x: torch.Tensor, rep: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    '''fp16 [rows][C] -> [rep*rows][C] (into `out`, row stride free, if given): the CFG fan-out of the shared UNet prefix, one launch.'''
    rows, C = x.shape
    if out is None:
        out = _empty((rep * rows, C), torch.float16, x)
    assert out.shape == (rep * rows, C) and out.dtype == x.dtype == torch.float16 and out.stride(1) == 1 and x.stride(1) == 1
    hip.call('fd_repeat_rows_f16', x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), rows, C, rep, hip.stream())
    return out


def concat_channels(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    M = a.shape[0]
    out = _empty((M, a.shape[1] + b.shape[1]), torch.float16, a)
    hip.call('fd_concat_channels_f16', a.data_ptr(), b.data_ptr(), out.data_ptr(), M, a.shape[1],
             b.shape[1], hip.stream())
    return out


def cfg_ddim_step(x: Optional[torch.Tensor], eps_nhwc: torch.Tensor, B: int, C: int, HW: int,
                  cfg: bool, guidance: float, coef=(0.0, 1.0, 1.0, 0.0), v_prediction: bool = False,
                  do_step: bool = True, eps_out: Optional[torch.Tensor] = None):
    hip.call('fd_cfg_ddim_step_f32', _p(x), eps_nhwc.data_ptr(), _p(eps_out), B, C, HW,
             eps_nhwc.stride(0), int(cfg), float(guidance), float(coef[0]), float(coef[1]),
             float(coef[2]), float(coef[3]), int(v_prediction), int(do_step), hip.stream())


def axpby(x: torch.Tensor, y: Optional[torch.Tensor], a: float, b: float,
          exp_half_x: bool = False) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty_like(x)
    hip.call('fd_axpby_f32', x.data_ptr(), _p(y.contiguous() if y is not None else None),
             out.data_ptr(), x.numel(), a, b, int(exp_half_x), hip.stream())
    return out


def cast_f16(x: torch.Tensor) -> torch.Tensor:
    x = x.to(torch.float32).contiguous()
    out = _empty(x.shape, torch.float16, x)
    hip.call('fd_cast_f32_to_f16', x.data_ptr(), out.data_ptr(), x.numel(), hip.stream())
    return out


def cast_f32(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    out = _empty(x.shape, torch.float32, x)
    hip.call('fd_cast_f16_to_f32', x.data_ptr(), out.data_ptr(), x.numel(), hip.stream())
    return out
