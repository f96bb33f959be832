// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (v_mfma_f32_16x16x32_f16).
//
//   C[M][N] = epilogue( sum_k A(m,k) * W[n][k] )          fp16 in, fp32 accumulate
//
// One kernel family serves every dense contraction of the UNet / VAE / CLIP towers:
//   * linear / 1x1 conv  : A(m,k) = A[m*lda + k]
//   * conv KHxKW (NHWC)  : A(m,k) gathered on the fly from the [B][Hi][Wi][Cin] input
//                          (k = (kh*KW+kw)*Cin + ci), with stride, asymmetric padding and
//                          an optional fused nearest-2x upsample of the input -- no im2col
//                          buffer ever touches HBM;
//   * batched (blockIdx.z) for per-sample attention matmuls of the VAE.
// Weights are [N][K] row-major (K contiguous), i.e. conv weights are stored
// [Cout][KH][KW][Cin].
//
// Tiling: BM x BN x 64 block tile, 4 wavefronts (2x2), 16x16x32 MFMA fragments, LDS rows
// of 64 halfs (128 B) XOR-swizzled by (row&7)<<4 so that ds_read_b128 fragment reads are
// at most 2-way conflicted; global->register->LDS staging with the next tile's loads
// issued before the current tile's MFMAs (one barrier per K tile, two LDS buffers).
// Workgroups are remapped so that each XCD (private 4 MiB L2) owns a contiguous range of
// tiles with the n-tile index fastest: the A panel of an m-tile is reused from L2.
//
// Epilogue (fused, no extra HBM round trip): alpha, bias[n], per-sample bias2[b][n]
// (ResBlock time-embedding add), SiLU / quick-GELU / GELU, GEGLU (x * gelu(gate) on
// interleaved weight rows), residual add, fp16 or fp32 store, or a transposed store
// ([b][n][m]) used to emit V^T for the attention kernel.
#include "gemm_epilogue.h"
#include "gn_slab.h"

typedef const __attribute__((address_space(1))) half_t* gptr_h;   // force global_load (not flat)
typedef const __attribute__((address_space(1))) u32x4* gptr_u4;

template <int BM, int BN, bool TRANS, bool CONV>
__global__ __launch_bounds__(256) void k_gemm_f16(GemmArgs g) {
    constexpr int WTM = BM / 2, WTN = BN / 2;   // wave tile
    constexpr int MI = WTM / 16, NI = WTN / 16; // 16x16 fragments per wave
    constexpr int AR = BM / 32, BR = BN / 32;   // 16-byte chunks per thread per tile
    constexpr int STAGE = (BM + BN) * 128;      // bytes per LDS stage: A tile then B tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping (blocks round-robin over the 8 XCDs) ----------------
    const int nb = g.tiles_m * g.tiles_n;
    int id = blockIdx.x;
    {
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_n = id % g.tiles_n, tile_m = id / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    gptr_h Ab = (gptr_h)g.A + (size_t)z * g.strideA;
    gptr_h Wb = (gptr_h)g.W + (size_t)z * g.strideW;

    // ---- per-thread staging rows (fixed for the whole K loop) -------------------------
    const int ck = tid & 7;    // 16-byte chunk (8 halfs) within the 64-wide K tile
    const int lr = tid >> 3;   // 0..31
    int a_off[AR];             // linear: m*lda ; conv: sample base offset + ck*8
    int a_y[AR], a_x[AR];      // conv: oy*stride - pad_t, ox*stride - pad_l
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + lr + 32 * i;
        a_ok[i] = m < g.M;
        const int mm = a_ok[i] ? m : 0;
        if (CONV) {
            const int hw = g.Ho * g.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_off[i] = b * g.Hi * g.Wi * g.Cpix + ck * 8;
            a_y[i] = oy * g.stride - g.pad_t;
            a_x[i] = ox * g.stride - g.pad_l;
        } else {
            a_off[i] = mm * g.lda;
            a_y[i] = a_x[i] = 0;
        }
    }
    int b_off[BR];
    bool b_ok[BR];
#pragma unroll
    for (int i = 0; i < BR; ++i) {
        const int n = n0 + lr + 32 * i;
        b_ok[i] = n < g.N;
        b_off[i] = (b_ok[i] ? n : 0) * g.ldw;
    }
    int lds_w[AR > BR ? AR : BR];  // swizzled LDS byte offset of this thread's chunk in row i
#pragma unroll
    for (int i = 0; i < (AR > BR ? AR : BR); ++i) {
        const int r = lr + 32 * i;
        lds_w[i] = r * 128 + ((ck ^ (r & 7)) << 4);
    }

    u32x4 ra[AR], rb[BR];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int Hv = g.up ? g.Hi * 2 : g.Hi, Wv = g.up ? g.Wi * 2 : g.Wi;
    const int nk_all = (g.K + BK - 1) / BK;
    const int kt_per = (nk_all + g.split_k - 1) / g.split_k;
    const int kt0 = blockIdx.y * kt_per;                       // this block's K-tile range
    const int nk = min(nk_all, kt0 + kt_per);
    const int ktail = g.K - ck * 8;  // this thread's chunk is inside K while kt*64 < ktail
    int kh = 0, kw = 0, ci0 = 0;     // conv: filter tap and channel base of the tile being loaded
    if (CONV && kt0 > 0) {
        const int tap = (kt0 * BK) / g.Cin;
        ci0 = kt0 * BK - tap * g.Cin;
        kh = tap / g.KW;
        kw = tap - kh * g.KW;
    }

    // Unconditional loads from clamped (always valid) addresses + select: no exec-mask
    // branches around the memory ops, and the waits stay counted (vmcnt) not drained.
#define GEMM_LOAD_TILE(KT)                                                                  \
    {                                                                                       \
        const bool kok = (KT) * BK < ktail;                                                 \
        const int kofs = kok ? (KT) * BK + ck * 8 : 0; /* never read past a row's K */      \
        if (CONV) {                                                                         \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                \
                int iy = a_y[i] + kh, ix = a_x[i] + kw;                                     \
                const bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv; \
                iy = ok ? iy : 0;                                                           \
                ix = ok ? ix : 0;                                                           \
                if (g.up) {                                                                 \
                    iy >>= 1;                                                               \
                    ix >>= 1;                                                               \
                }                                                                           \
                const u32x4 v = *(gptr_u4)(Ab + a_off[i] + (iy * g.Wi + ix) * g.Cpix + ci0); \
                ra[i] = ok ? v : zero4;                                                     \
            }                                                                               \
            ci0 += BK;                                                                      \
            if (ci0 >= g.Cin) {                                                             \
                ci0 = 0;                                                                    \
                if (++kw == g.KW) {                                                         \
                    kw = 0;                                                                 \
                    ++kh;                                                                   \
                }                                                                           \
            }                                                                               \
        } else {                                                                            \
            _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                \
                const u32x4 v = *(gptr_u4)(Ab + a_off[i] + kofs);                           \
                ra[i] = (a_ok[i] && kok) ? v : zero4;                                       \
            }                                                                               \
        }                                                                                   \
        _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                    \
            const u32x4 v = *(gptr_u4)(Wb + b_off[i] + kofs);                               \
            rb[i] = (b_ok[i] && kok) ? v : zero4;                                           \
        }                                                                                   \
    }
#define GEMM_STORE_TILE(BUF)                                                                \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < AR; ++i)                                      \
            *reinterpret_cast<u32x4*>(smem + (BUF) * STAGE + lds_w[i]) = ra[i];             \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                      \
            *reinterpret_cast<u32x4*>(smem + (BUF) * STAGE + BM * 128 + lds_w[i]) = rb[i];  \
    }

    floatx4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    GEMM_LOAD_TILE(kt0);
    GEMM_STORE_TILE(0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    // fragment read offsets: row r = base + fr, chunk (ks*4 + fq) ^ (r & 7); (r & 7) == (fr & 7)
    const int frag_a = (wm * WTM + fr) * 128, frag_b = BM * 128 + (wn * WTN + fr) * 128;
    const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;
    for (int kt = kt0; kt < nk; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < nk) GEMM_LOAD_TILE(kt + 1);
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            half8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (TRANS)  // lane: 4 consecutive m for one n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j],
                                                                           acc[i][j], 0, 0, 0);
                    else        // lane: 4 consecutive n for one m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i],
                                                                           acc[i][j], 0, 0, 0);
                }
        }
        if (kt + 1 < nk) GEMM_STORE_TILE(cur ^ 1);
        __syncthreads();
    }
#undef GEMM_LOAD_TILE
#undef GEMM_STORE_TILE

    gemm_epilogue<BM, BN, TRANS>(g, acc, m0, n0, wm, wn, fr, fq, z);
}

// ---------------------------------------------------------------------------------------
// LDS-DMA main loop: tiles go HBM/L2 -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR
// staging and no ds_write pass.  One wave instruction moves 64 lanes x 16 B = 8 rows of the
// 128-byte-row tile; the LDS image is lane-linear, so the XOR swizzle is applied to the
// per-lane SOURCE chunk (lane slot p of row r loads chunk p ^ (r & 7)) and the fragment reads
// use the same involution.  Out-of-tile rows, conv zero padding and the K tail are produced
// by the buffer descriptor's bounds check (an offset past num_records reads 0), so there is
// no select or branch on the load path.  The K-tile offset rides in the scalar soffset, so
// per-lane address math only runs when the filter tap changes.
template <int BM, int BN, bool CONV, int WM, bool TRANS = false, int NS = 2, int WN = 2, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_f16_dma(GemmArgs g, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)  // body uses device-only builtins (host pass sees a stub)
    constexpr int NW = WN * WM;                 // waves: WM along M x WN along N
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int AG = BM / 8, BG = BN / 8;     // 8-row DMA groups of the A / B tile
    constexpr int AR = (AG + NW - 1) / NW, BR = (BG + NW - 1) / NW;  // DMA instr. per wave
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nb = g.tiles_m * g.tiles_n;
    int id = blockIdx.x;
    int kslice = blockIdx.y;
    if (g.sk_flat) {
        // split-K on a flat grid: workgroups go round-robin over the 8 XCDs, so with slice = id % split_k every
        // workgroup of one K slice (all m- and n-tiles) lands on the same XCD(s): that slice of W and of the
        // gathered input is fetched into ONE L2 instead of up to eight (PMC, M 1024 N 1280 K 11520 split 8 with the
        // 2-D grid: 125 MB read per launch for 53 MB of operands)
        kslice = id % g.split_k;
        id /= g.split_k;
    } else {
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_n = id % g.tiles_n, tile_m = id / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    const bool tr_tile = EPI == 13 && __builtin_amdgcn_readfirstlane(n0 >= g.tr_n0 ? 1 : 0) != 0;   // fd_gemm_desc.trans_n0
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.A + (size_t)z * g.strideA), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.W + (size_t)z * g.strideW), 0, w_bytes, 0x00020000);
    // appended phase (a second operand accumulated by the same K loop: the ResBlock shortcut in conv2, the
    // transformer's proj_out folded through its feed-forward output GEMM): rows of A2
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.K2 ? g.A2 : g.A), 0, g.K2 ? g.a2_bytes : 0u, 0x00020000);
    // this tile's bias -> LDS (4 B per lane, 64 columns per wave instruction; zeros past N),
    // ahead of the first K-tile so that it lands with it
    float* bias_s = reinterpret_cast<float*>(smem + NS * STAGE);
    // EPI != 0 (lean epilogue): both bias tiles are always staged -- through a zero-length
    // descriptor (every read returns 0) when the GEMM has no bias / no per-sample bias -- so the
    // epilogue reads them without a branch.
    if ((g.bias && g.bias_lds) || (EPI != 0 && EPI != 7)) {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(g.bias ? (const void*)(g.bias + (size_t)z * g.strideBias) : (const void*)g.W), 0, g.bias ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
        if (wave * 64 + lane < BN)   // lanes past the tile would spill into the next LDS buffer
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(bias_s + wave * 64), 4,
                                                     (unsigned)(n0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }
    // per-sample bias (ResBlock time embedding): one row when the whole tile lies in one sample
    const int b_first = m0 / g.rows_per_batch;
    // (EPI 13: the LayerNorm fold's column sums are one row for every sample -- ldb2 == 0 --, staged whatever the sample boundary)
    const bool b2_staged = g.bias2 && g.bias_lds && ((EPI == 13 && g.ln_stats != nullptr) || (min(m0 + BM, g.M) - 1) / g.rows_per_batch == b_first);
    if (b2_staged || (EPI != 0 && EPI != 7)) {
        const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(b2_staged ? (const void*)(g.bias2 + (size_t)b_first * g.ldb2) : (const void*)g.W), 0,
            b2_staged ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
        if (wave * 64 + lane < BN)   // lanes past the tile would spill into the next LDS buffer
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(bias_s + BN + wave * 64), 4,
                                                     (unsigned)(n0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }

    const int rsub = lane >> 3;            // row inside the 8-row group
    const int ck = (lane & 7) ^ rsub;      // source chunk for this lane's LDS slot (swizzle)
    // byte offsets; >= *_bytes means "reads zero".  Register diet (the 64-row wave tiles run at
    // the 128-VGPR limit of 4 waves/SIMD): the conv row state is (sample base, packed y|x); a
    // row past M carries y = 0xffff so every tap fails the bounds test; the W rows of one lane
    // are a fixed stride apart, so ONE per-lane offset + a scalar row-group offset in soffset
    // addresses them all (rows past N land beyond w_bytes by themselves: ldw >= K).
    unsigned a_voff[AR];
    int a_off[AR];
    unsigned a_yx[AR];                     // (oy*stride - pad_t + 16) << 16 | (ox*stride - pad_l + 16)
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = m0 + (i * NW + wave) * 8 + rsub;
        const bool ok = m < g.M;
        const int mm = ok ? m : 0;
        if (CONV) {
            const int hw = g.Ho * g.Wo;
            const int b = mm / hw, rem = mm - b * hw;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_off[i] = b * g.Hi * g.Wi * g.Cpix + ck * 8;
            // phase mode: the 2x2 window of output parity (py, px) = (z >> 1, z & 1) starts one row / column earlier for parity 0
            const int pad_t = g.phase ? 1 - (z >> 1) : g.pad_t, pad_l = g.phase ? 1 - (z & 1) : g.pad_l;
            a_yx[i] = ok ? ((unsigned)(oy * g.stride - pad_t + 16) << 16) |
                               (unsigned)(ox * g.stride - pad_l + 16)
                         : 0xffff0000u;
            a_voff[i] = a_bytes;
        } else {
            a_off[i] = 0;
            a_yx[i] = 0;
            a_voff[i] = ok ? (unsigned)(mm * g.lda + ck * 8) * 2u : a_bytes;
        }
    }
    const unsigned b_voff0 = (unsigned)((n0 + wave * 8 + rsub) * g.ldw + ck * 8) * 2u;
    const int b_group = NW * 8 * g.ldw * 2;   // bytes between a lane's consecutive W rows
    const int Hv = g.up ? g.Hi * 2 : g.Hi, Wv = g.up ? g.Wi * 2 : g.Wi;
    const int nkc = (g.K + BK - 1) / BK;                                  // K-tiles of the convolution / GEMM proper
    const int nk_all = nkc + g.K2 / BK;                                   // + the appended phase over A2
    const int kt_per = (nk_all + g.split_k - 1) / g.split_k;
    const int kt0 = kslice * kt_per;
    const int nk = min(nk_all, kt0 + kt_per);
    const int ktail = g.K + g.K2 - ck * 8;   // (with an appended phase K and K2 are multiples of 64: no tail)
    int kh = 0, kw = 0, ci0 = 0;
    bool new_tap = true;
    // tap_fast: K-tiles visit all filter taps of one 64-channel slice before the next slice
    // (the nine taps re-read the same input rows, so the re-use distance in the XCD's L2 drops
    // from Cin/64 K-tiles to one); the W K-offset follows, the sum is only re-ordered
    const int ntaps = CONV ? g.K / g.Cin : 1;
    if (CONV && kt0 > 0 && kt0 < nkc) {
        if (g.tap_fast) {
            const int tap = kt0 % ntaps;
            ci0 = (kt0 / ntaps) * BK;
            kh = tap / g.KW;
            kw = tap - kh * g.KW;
        } else {
            const int tap = (kt0 * BK) / g.Cin;
            ci0 = kt0 * BK - tap * g.Cin;
            kh = tap / g.KW;
            kw = tap - kh * g.KW;
        }
    }

#define GEMM_DMA_TILE(KT, BUF)                                                              \
    {                                                                                       \
        char* stage = smem + (BUF) * STAGE;                                                 \
        int wko = CONV ? ((kh * g.KW + kw) * g.Cin + ci0) * 2 : (KT) * BK * 2;              \
        if (g.K2 && (KT) >= nkc) { /* appended phase: plain rows of A2, W columns continue after K */ \
            if (new_tap) {                                                                  \
                _Pragma("unroll") for (int i = 0; i < AR; ++i) {                            \
                    const int m = m0 + (i * NW + wave) * 8 + rsub;                          \
                    a_voff[i] = m < g.M ? (unsigned)(m * g.lda2 + ck * 8) * 2u : g.a2_bytes; \
                }                                                                           \
                new_tap = false;                                                            \
            }                                                                               \
            const int soff = ((KT) - nkc) * BK * 2;                                         \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                  \
                if (AG % NW == 0 || i * NW + wave < AG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsA2, (lds_ptr)(stage + (i * NW + wave) * 1024), 16, a_voff[i], soff, 0, 0); \
            wko = (g.K + ((KT) - nkc) * BK) * 2;                                            \
        } else if (CONV) {                                                                  \
            if (new_tap) {                                                                  \
                _Pragma("unroll") for (int i = 0; i < AR; ++i) {                            \
                    int iy = (int)(a_yx[i] >> 16) + kh - 16;                                \
                    int ix = (int)(a_yx[i] & 0xffffu) + kw - 16;                            \
                    const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv; \
                    if (g.up) {                                                             \
                        iy >>= 1;                                                           \
                        ix >>= 1;                                                           \
                    }                                                                       \
                    a_voff[i] = ok ? (unsigned)(a_off[i] + (iy * g.Wi + ix) * g.Cpix) * 2u   \
                                   : a_bytes;                                               \
                }                                                                           \
                new_tap = false;                                                            \
            }                                                                               \
            const int soff = ci0 * 2;                                                       \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                  \
                if (AG % NW == 0 || i * NW + wave < AG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsA, (lds_ptr)(stage + (i * NW + wave) * 1024), 16, a_voff[i], soff, 0, 0); \
            if (g.tap_fast) {                                                               \
                new_tap = true;                                                             \
                if (++kw == g.KW) {                                                         \
                    kw = 0;                                                                 \
                    if ((kh + 1) * g.KW == ntaps) {                                         \
                        kh = 0;                                                             \
                        ci0 += BK;                                                          \
                    } else {                                                                \
                        ++kh;                                                               \
                    }                                                                       \
                }                                                                           \
            } else {                                                                        \
                ci0 += BK;                                                                  \
                if (ci0 >= g.Cin) {                                                         \
                    ci0 = 0;                                                                \
                    new_tap = true;                                                         \
                    if (++kw == g.KW) {                                                     \
                        kw = 0;                                                             \
                        ++kh;                                                               \
                    }                                                                       \
                }                                                                           \
            }                                                                               \
        } else {                                                                            \
            const bool kok = (KT) * BK < ktail;                                             \
            const int soff = (KT) * BK * 2;                                                 \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                  \
                if (AG % NW == 0 || i * NW + wave < AG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsA, (lds_ptr)(stage + (i * NW + wave) * 1024), 16,                      \
                    kok ? a_voff[i] : a_bytes, soff, 0, 0);                                 \
        }                                                                                   \
        {                                                                                   \
            const bool kok = CONV || (KT) * BK < ktail; /* conv: K is a multiple of 64 */   \
            const unsigned bv = kok ? b_voff0 : w_bytes;                                    \
            const int soff = wko;                                                           \
            _Pragma("unroll") for (int i = 0; i < BR; ++i)                                  \
                if (BG % NW == 0 || i * NW + wave < BG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsW, (lds_ptr)(stage + BM * 128 + (i * NW + wave) * 1024), 16,           \
                    bv, soff + i * b_group, 0, 0);                                          \
        }                                                                                   \
    }

    floatx4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    GEMM_DMA_TILE(kt0, 0);
    if (NS == 3 && kt0 + 1 < nk) GEMM_DMA_TILE(kt0 + 1, 1);
    // LayerNorm fold fed with the producer's partial sums: this tile's rows are finalised into LDS here, behind the first DMAs (their wait is
    // the one the loop makes anyway); every barrier of the K loop lies between this write and the epilogue's reads
    float* const stats_s = bias_s + 4 * BN;
    const bool ln_lds = (EPI == 5 || EPI == 6 || EPI == 7 || EPI == 13) && __builtin_amdgcn_readfirstlane(g.ln_parts > 1 ? 1 : 0) != 0;
    if constexpr (EPI == 5 || EPI == 6 || EPI == 7 || EPI == 13) {
        if (ln_lds) ln_tile_stats_to_lds<BM>(g, m0, tid, stats_s);
    }
    const lds_cfloat stats_tile = ln_lds ? (lds_cfloat)stats_s : (lds_cfloat) nullptr;
    if (NS == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int fr = lane & 15, fq = lane >> 4;
    const int frag_a = (wm * WTM + fr) * 128, frag_b = BM * 128 + (wn * WTN + fr) * 128;
    const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;
    int cur = 0;
    for (int kt = kt0; kt < nk; ++kt) {
        if (NS == 2) {
            if (kt + 1 < nk) GEMM_DMA_TILE(kt + 1, cur ^ 1);
        } else {
            // 3 LDS stages: tile kt must have landed, tile kt+1 may stay in flight across the
            // barrier (counted vmcnt + raw s_barrier: __syncthreads() would drain the DMA queue)
            if (kt + 1 < nk) {
                if (BG % NW == 0 || wave < BG % NW)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + BR) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR + BR - 1) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) GEMM_DMA_TILE(kt + 2, cur == 0 ? 2 : cur - 1);
        }
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            if constexpr (MI * NI >= 20 && !TRANS && FD_T16_MODE > 0) {
                // 64x80 wave tiles: 80 accumulator registers leave no room for all nine
                // fragments at 4 waves/SIMD.  Keep the A fragments, stream the W fragments with
                // FD_T16_MODE in flight, and pin that order (the scheduler would hoist every read).
                half8 fa[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
                half8 fb[NI];
#pragma unroll
                for (int j = 0; j < FD_T16_MODE && j < NI; ++j)
                    fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    if (j + FD_T16_MODE < NI)
                        fb[j + FD_T16_MODE] = *reinterpret_cast<const half8*>(st + frag_b + (j + FD_T16_MODE) * 2048 + sw);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            half8 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
            if constexpr (EPI == 13) {
                // transposed tail: the tiles that store transposed take the operands in the TRANS order (a lane then owns 4 consecutive
                // rows of one column); one workgroup-uniform branch per K step, no select on the results
                if (tr_tile) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j],
                                                                               acc[i][j], 0, 0, 0)
                                      : __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i],
                                                                               acc[i][j], 0, 0, 0);
            }
            }
        }
        // keep the wait for the NEXT K-tile's DMA behind ALL of this tile's MFMAs: left alone, hipcc sinks the second
        // half of the MFMAs below the wait + barrier (they only touch registers), which halves the cover of the DMA
        // round trip (ISA of the 256x160 tile; the persistent kernel, whose wait sits behind a branch, was 2.5-5 % faster)
        // (the VAE's 256x128 / 256x256 tiles are 2 % FASTER with the compiler's own placement: 160-wide tiles only)
        if constexpr (BN == 160) __builtin_amdgcn_sched_barrier(0);
        if (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            cur ^= 1;
        } else {
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
#undef GEMM_DMA_TILE
    if constexpr (EPI == 0 || EPI == 7)   // 7: the generic epilogue with the LayerNorm fold compiled in
        gemm_epilogue<BM, BN, TRANS, WM, WN, EPI == 7>(g, acc, m0, n0, wm, wn, fr, fq, z, (g.bias && g.bias_lds) ? (lds_cfloat)bias_s : (lds_cfloat) nullptr,
                                                       b2_staged ? (lds_cfloat)(bias_s + BN) : (lds_cfloat) nullptr, kslice, stats_tile);
    else if constexpr (EPI == 8 || EPI == 9) {   // lean + LayerNorm statistics of the written rows; the
        // exchange buffer reuses stage 0 (the 2-stage K loop ends with a barrier: the stages are dead; the 3-stage loop
        // has its barrier at the TOP of an iteration, so the last K-tile may still be read: one more barrier)
        if constexpr (NS == 3) __syncthreads();
        gemm_epilogue_fast<MI, NI, FD_ACT_NONE, EPI == 9, true, false, true, WN>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            reinterpret_cast<float*>(smem), wm * WTM + fr, wn, m0);
    } else if constexpr (EPI == 13) {
        // LayerNorm-fold GEMM with a transposed tail (fd_gemm_desc.trans_n0): q|k tiles through the lean fold epilogue, V tiles through the
        // transposed-store epilogue into C2 with the tail's columns re-based to 0
        if (tr_tile) {
            GemmArgs gt = g;
            gt.C = g.C2;
            gt.N = g.N - g.tr_n0;
            gt.bias = g.bias ? g.bias + g.tr_n0 : nullptr;
            gt.bias2 = g.bias2 ? g.bias2 + g.tr_n0 : nullptr;
            gemm_epilogue<BM, BN, true, WM, WN, true>(gt, acc, m0, n0 - g.tr_n0, wm, wn, fr, fq, z, (g.bias && g.bias_lds) ? (lds_cfloat)bias_s : (lds_cfloat) nullptr,
                                                      b2_staged ? (lds_cfloat)(bias_s + BN) : (lds_cfloat) nullptr, -1, stats_tile);
        } else {
            gemm_epilogue_fast<MI, NI, FD_ACT_NONE, false, true, true>(
                g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
                nullptr, wm * WTM + fr, wn, m0, stats_tile);
        }
    } else if constexpr (EPI == 11 || EPI == 12) {   // lean (+ residual) + GroupNorm partial sums of the tile's output (gn_part_out)
        if constexpr (NS == 3) __syncthreads();
        gemm_epilogue_fast<MI, NI, FD_ACT_NONE, EPI == 12, true, false, false, WN, WM>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            reinterpret_cast<float*>(smem), wm * WTM + fr, wn, m0);
    } else
        gemm_epilogue_fast<MI, NI, (EPI == 3 || EPI == 6) ? FD_ACT_GEGLU : FD_ACT_NONE, EPI == 2, true, (EPI == 5 || EPI == 6)>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            nullptr, wm * WTM + fr, wn, m0, stats_tile);
#endif
}

// Persistent variant of the LDS-DMA loop (used for short K loops).
template <int BM, int BN, bool CONV, int WM, bool TRANS = false, int WN = 2, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_gemm_f16_dmap(GemmArgs g, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)  // body uses device-only builtins (host pass sees a stub)
    constexpr int NW = WN * WM;                 // waves: WM along M x WN along N
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    constexpr int AG = BM / 8, BG = BN / 8;     // 8-row DMA groups of the A / B tile
    constexpr int AR = (AG + NW - 1) / NW, BR = (BG + NW - 1) / NW;  // DMA instr. per wave
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nb = g.tiles_m * g.tiles_n;
    const int z = blockIdx.z;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.A + (size_t)z * g.strideA), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.W + (size_t)z * g.strideW), 0, w_bytes, 0x00020000);
    // per-tile bias staged in LDS (two buffers: the next tile's bias arrives with its first
    // K-tile while the current tile's epilogue still reads its own)
    float* bias_s = reinterpret_cast<float*>(smem + 2 * STAGE);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.bias ? g.bias : (const float*)g.W), 0, g.bias ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
    int bias_par = 0;
    // EPI 5 / 6 (LayerNorm fold): the column sums of the folded weights (bias2, one row) ride along
    const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((EPI == 5 || EPI == 6 || EPI == 7) && g.bias2) ? (const void*)g.bias2 : (const void*)g.W), 0,
        ((EPI == 5 || EPI == 6 || EPI == 7) && g.bias2) ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
#define GEMM_DMA_BIAS(PAR)                                                                  \
    if (wave * 64 + lane < BN) {                                                             \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(bias_s + (PAR) * BN + wave * 64), 4, \
                                                 (unsigned)(ld_n0 + wave * 64 + lane) * 4u, 0, 0, 0); \
        if constexpr (EPI == 5 || EPI == 6 || EPI == 7)                                      \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(bias_s + (2 + (PAR)) * BN + wave * 64), 4, \
                                                     (unsigned)(ld_n0 + wave * 64 + lane) * 4u, 0, 0, 0); \
    }

    const int rsub = lane >> 3;            // row inside the 8-row group
    const int ck = (lane & 7) ^ rsub;      // source chunk for this lane's LDS slot (swizzle)
    const int Hv = g.up ? g.Hi * 2 : g.Hi, Wv = g.up ? g.Wi * 2 : g.Wi;
    const int nk_all = (g.K + BK - 1) / BK;
    const int kt_per = (nk_all + g.split_k - 1) / g.split_k;
    const int kt0 = blockIdx.y * kt_per;
    const int nk = min(nk_all, kt0 + kt_per);
    const int ktail = g.K - ck * 8;

    // ---- loader state of the tile whose K-tiles are being fetched -----------------------
    unsigned a_voff[AR], b_voff[BR];       // byte offsets; >= *_bytes means "reads zero"
    int a_off[AR], a_y[AR], a_x[AR];
    bool a_ok[AR];
    int kh = 0, kw = 0, ci0 = 0;
    bool new_tap = true;
    int ld_m0 = 0, ld_n0 = 0;

    // XCD-aware tile order: every tile of this workgroup lives on its own XCD's chunk
#define GEMM_SETUP_TILE(T)                                                                  \
    {                                                                                       \
        int id_ = (T);                                                                      \
        {                                                                                   \
            const int q = nb >> 3, r = nb & 7, xcd = id_ & 7, slot = id_ >> 3;              \
            id_ = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;           \
        }                                                                                   \
        const int tile_n = id_ % g.tiles_n, tile_m = id_ / g.tiles_n;                       \
        ld_m0 = tile_m * BM;                                                                \
        ld_n0 = tile_n * BN;                                                                \
        _Pragma("unroll") for (int i = 0; i < AR; ++i) {                                    \
            const int m = ld_m0 + (i * NW + wave) * 8 + rsub;                                \
            a_ok[i] = m < g.M;                                                              \
            const int mm = a_ok[i] ? m : 0;                                                 \
            if (CONV) {                                                                     \
                const int hw = g.Ho * g.Wo;                                                 \
                const int b = mm / hw, rem = mm - b * hw;                                   \
                const int oy = rem / g.Wo, ox = rem - oy * g.Wo;                            \
                a_off[i] = b * g.Hi * g.Wi * g.Cpix + ck * 8;                                \
                a_y[i] = oy * g.stride - g.pad_t;                                           \
                a_x[i] = ox * g.stride - g.pad_l;                                           \
                a_voff[i] = a_bytes;                                                        \
            } else {                                                                        \
                a_off[i] = a_y[i] = a_x[i] = 0;                                             \
                a_voff[i] = a_ok[i] ? (unsigned)(mm * g.lda + ck * 8) * 2u : a_bytes;       \
            }                                                                               \
        }                                                                                   \
        _Pragma("unroll") for (int i = 0; i < BR; ++i) {                                    \
            const int n = ld_n0 + (i * NW + wave) * 8 + rsub;                                \
            b_voff[i] = n < g.N ? (unsigned)(n * g.ldw + ck * 8) * 2u : w_bytes;            \
        }                                                                                   \
        kh = kw = ci0 = 0;                                                                  \
        new_tap = true;                                                                     \
        if (CONV && kt0 > 0) {                                                              \
            const int tap = (kt0 * BK) / g.Cin;                                             \
            ci0 = kt0 * BK - tap * g.Cin;                                                   \
            kh = tap / g.KW;                                                                \
            kw = tap - kh * g.KW;                                                           \
        }                                                                                   \
    }

#define GEMM_DMA_TILE(KT, BUF)                                                              \
    {                                                                                       \
        char* stage_ = smem + (BUF) * STAGE;                                                \
        if (CONV) {                                                                         \
            if (new_tap) {                                                                  \
                _Pragma("unroll") for (int i = 0; i < AR; ++i) {                            \
                    int iy = a_y[i] + kh, ix = a_x[i] + kw;                                 \
                    const bool ok = a_ok[i] && (unsigned)iy < (unsigned)Hv &&               \
                                    (unsigned)ix < (unsigned)Wv;                            \
                    if (g.up) {                                                             \
                        iy >>= 1;                                                           \
                        ix >>= 1;                                                           \
                    }                                                                       \
                    a_voff[i] = ok ? (unsigned)(a_off[i] + (iy * g.Wi + ix) * g.Cpix) * 2u   \
                                   : a_bytes;                                               \
                }                                                                           \
                new_tap = false;                                                            \
            }                                                                               \
            const int soff = ci0 * 2;                                                       \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                  \
                if (AG % NW == 0 || i * NW + wave < AG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsA, (lds_ptr)(stage_ + (i * NW + wave) * 1024), 16, a_voff[i], soff, 0, 0); \
            ci0 += BK;                                                                      \
            if (ci0 >= g.Cin) {                                                             \
                ci0 = 0;                                                                    \
                new_tap = true;                                                             \
                if (++kw == g.KW) {                                                         \
                    kw = 0;                                                                 \
                    ++kh;                                                                   \
                }                                                                           \
            }                                                                               \
        } else {                                                                            \
            const bool kok = (KT) * BK < ktail;                                             \
            const int soff = (KT) * BK * 2;                                                 \
            _Pragma("unroll") for (int i = 0; i < AR; ++i)                                  \
                if (AG % NW == 0 || i * NW + wave < AG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsA, (lds_ptr)(stage_ + (i * NW + wave) * 1024), 16,                     \
                    kok ? a_voff[i] : a_bytes, soff, 0, 0);                                 \
        }                                                                                   \
        {                                                                                   \
            const bool kok = (KT) * BK < ktail;                                             \
            const int soff = (KT) * BK * 2;                                                 \
            _Pragma("unroll") for (int i = 0; i < BR; ++i)                                  \
                if (BG % NW == 0 || i * NW + wave < BG)                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(                                   \
                    rsW, (lds_ptr)(stage_ + BM * 128 + (i * NW + wave) * 1024), 16,          \
                    kok ? b_voff[i] : w_bytes, soff, 0, 0);                                 \
        }                                                                                   \
    }

    const int fr = lane & 15, fq = lane >> 4;
    const int frag_a = (wm * WTM + fr) * 128, frag_b = BM * 128 + (wn * WTN + fr) * 128;
    const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;

    // ---- persistent loop over this workgroup's tiles: the first K-tile of the NEXT output
    // tile is already in flight while the current tile's epilogue runs, so short-K GEMMs do
    // not expose the HBM/L2 latency of a fresh prologue for every tile ----------------------
    int t = blockIdx.x;
    int stage = 0;
    GEMM_SETUP_TILE(t);
    GEMM_DMA_BIAS(0);
    GEMM_DMA_TILE(kt0, 0);
    floatx4 acc[MI][NI];
#define GEMM_MFMA_TILE(ST)                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                      \
        const int sw = ks ? sw1 : sw0;                                                      \
        half8 fa[MI], fb[NI];                                                               \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                      \
            fa[i] = *reinterpret_cast<const half8*>((ST) + frag_a + i * 2048 + sw);         \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                      \
            fb[j] = *reinterpret_cast<const half8*>((ST) + frag_b + j * 2048 + sw);         \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                      \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                  \
                acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(                 \
                                        fa[i], fb[j], acc[i][j], 0, 0, 0)                   \
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(                 \
                                        fb[j], fa[i], acc[i][j], 0, 0, 0);                  \
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the first tile's first K-tile (+ bias)
#pragma clang loop unroll(disable)
    while (t < nb) {
        const int m0 = ld_m0, n0 = ld_n0;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        // bare barrier: every wave has already waited for its share of this tile's first K-tile (before the loop /
        // before the previous tile's epilogue).  __syncthreads() would also drain the previous epilogue's global
        // STORES (vmcnt counts them): a store round trip exposed once per tile
        __builtin_amdgcn_s_barrier();
        const int t_next = t + gridDim.x;
#pragma clang loop unroll(disable)
        for (int kt = kt0; kt < nk; ++kt) {
            const int cur = stage;
            const bool last = kt + 1 >= nk;
            if (!last) {
                GEMM_DMA_TILE(kt + 1, cur ^ 1);
            } else if (t_next < nb) {
                GEMM_SETUP_TILE(t_next);
                GEMM_DMA_BIAS(bias_par ^ 1);
                GEMM_DMA_TILE(kt0, cur ^ 1);
            }
            const char* st = smem + cur * STAGE;
            GEMM_MFMA_TILE(st);
            stage ^= 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // last: the next tile's first K-tile, ahead of the epilogue's stores
            if (!last) __syncthreads();
        }
        if constexpr (EPI == 0 || EPI == 7)
            gemm_epilogue<BM, BN, TRANS, WM, WN, EPI == 7>(g, acc, m0, n0, wm, wn, fr, fq, z,
                                                           (g.bias && g.bias_lds) ? (lds_cfloat)(bias_s + bias_par * BN) : (lds_cfloat) nullptr,
                                                           (EPI == 7 && g.ln_stats && g.bias2) ? (lds_cfloat)(bias_s + (2 + bias_par) * BN) : (lds_cfloat) nullptr);
        else
            gemm_epilogue_fast<MI, NI, (EPI == 3 || EPI == 6) ? FD_ACT_GEGLU : FD_ACT_NONE, (EPI == 2 || EPI == 9), false, (EPI == 5 || EPI == 6)>(
                g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z,
                (lds_cfloat)(bias_s + bias_par * BN), (lds_cfloat)(bias_s + (2 + bias_par) * BN));
        bias_par ^= 1;
        t = t_next;
    }
#undef GEMM_MFMA_TILE
#undef GEMM_DMA_TILE
#undef GEMM_SETUP_TILE
#undef GEMM_DMA_BIAS
#endif
}

// Sums the split-K partial slabs in a fixed order and applies the fused epilogue.  Templated on the slab count so
// that ALL of an element group's loads (slabs, biases, residual) are issued together: with a run-time slab loop hipcc
// emitted load / s_waitcnt vmcnt(0) / add per slab -- up to 8 + 3 dependent round trips per 16 bytes of output.
template <int S>
__device__ __forceinline__ void splitk_finish_body(const GemmArgs& g) {
    const int n4 = g.N >> 2;
    const size_t total = (size_t)g.M * n4;
    const size_t slab = (size_t)g.M * g.N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int m = (int)(e / n4), nb0 = (int)(e - (size_t)m * n4) * 4;
        const float* src = g.ws + (size_t)m * g.N + nb0;
        float4 p[S > 0 ? S : 1];
        if constexpr (S > 0) {
#pragma unroll
            for (int s = 0; s < S; ++s) p[s] = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
        } else {
            p[0] = *reinterpret_cast<const float4*>(src);
        }
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), b2 = bb;
        half4 rr = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if (g.bias) bb = *reinterpret_cast<const float4*>(g.bias + nb0);
        if (g.bias2) b2 = *reinterpret_cast<const float4*>(g.bias2 + (size_t)(m / g.rows_per_batch) * g.ldb2 + nb0);
        if (g.res) rr = *reinterpret_cast<const half4*>(g.res + (size_t)(g.res_rows ? m % g.res_rows : m) * g.ldr + nb0);
        float4 a = p[0];
        if constexpr (S > 0) {
#pragma unroll
            for (int s = 1; s < S; ++s) { a.x += p[s].x; a.y += p[s].y; a.z += p[s].z; a.w += p[s].w; }   // fixed order
        } else {
            for (int s = 1; s < g.split_k; ++s) {
                const float4 q = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
                a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
            }
        }
        // bias + per-sample bias first, then one fma (as the GEMM epilogues)
        const float bs[4] = {bb.x + b2.x, bb.y + b2.y, bb.z + b2.z, bb.w + b2.w};
        float v[4] = {fmaf(a.x, g.alpha, bs[0]), fmaf(a.y, g.alpha, bs[1]), fmaf(a.z, g.alpha, bs[2]), fmaf(a.w, g.alpha, bs[3])};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], g.act);
        if (g.res) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
        }
        if (g.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (size_t)m * g.ldc + nb0) =
                make_float4(v[0], v[1], v[2], v[3]);
        } else {
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
            *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(g.C) + (size_t)m * g.ldc + nb0) = o;
        }
    }
}

__global__ __launch_bounds__(256) void k_splitk_finish(GemmArgs g) {
    switch (g.split_k) {
        case 2: splitk_finish_body<2>(g); break;
        case 4: splitk_finish_body<4>(g); break;
        case 8: splitk_finish_body<8>(g); break;
        case 16: splitk_finish_body<16>(g); break;
        default: splitk_finish_body<0>(g); break;
    }
}


// Split-K finish FUSED with the GroupNorm(+SiLU) that consumes the output (fd_gemm_desc.gn_out): a workgroup owns one sample x
// gn_gb groups, sums the fp32 partial slabs of its [HW][gn_gb * cpg] slab in the finish kernel's fixed order, applies bias +
// per-sample bias (+ residual), rounds to fp16 -- the convolution's output, stored to C unless gn_skip_c -- and normalises those
// registers with the slab GroupNorm body (gn_slab.h): same bits as k_splitk_finish followed by k_gn_slab<256, 16>, one launch
// and one fp16 round trip through HBM less.  All of a vector's slab loads are issued together (S is a template parameter).
template <int S>
__global__ __launch_bounds__(256) void k_splitk_finish_gn(GemmArgs g) {
    extern __shared__ float gn_sm[];
    constexpr int NT = 256, NV = 16;
    const int HW = g.rows_per_batch, C = g.N, cpg = C / g.gn_groups, GB = g.gn_gb, CB = cpg * GB, c8 = CB >> 3;
    const int b = blockIdx.x, ch0 = blockIdx.y * CB, tid = threadIdx.x;
    const int cc = tid % c8;
    const int n = ch0 + cc * 8;
    float bs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] = 0.f;
    if (g.bias) {
        const float4 lo = *reinterpret_cast<const float4*>(g.bias + n), hi = *reinterpret_cast<const float4*>(g.bias + n + 4);
        bs[0] = lo.x; bs[1] = lo.y; bs[2] = lo.z; bs[3] = lo.w; bs[4] = hi.x; bs[5] = hi.y; bs[6] = hi.z; bs[7] = hi.w;
    }
    if (g.bias2) {   // bias + per-sample bias first, then one fma (as every epilogue of the family)
        const float* b2 = g.bias2 + (size_t)b * g.ldb2 + n;
        const float4 lo = *reinterpret_cast<const float4*>(b2), hi = *reinterpret_cast<const float4*>(b2 + 4);
        bs[0] += lo.x; bs[1] += lo.y; bs[2] += lo.z; bs[3] += lo.w; bs[4] += hi.x; bs[5] += hi.y; bs[6] += hi.z; bs[7] += hi.w;
    }
    const size_t slab = (size_t)g.M * g.N;
    const size_t row0 = (size_t)b * HW;
    const float* __restrict__ src0 = g.ws + row0 * g.N + n;
    const int PL = NT / c8, pl = tid / c8;
    const int nv = (HW + PL - 1) / PL;          // vectors a thread of this launch really holds (uniform)
    const bool active = pl < PL;
    // CH vectors at a time with ALL their 2 S CH slab loads issued before the first use (addresses clamped instead of
    // predicated: a branch per vector made every vector its own load -> wait -> add round trip, 6 of them at the 16x16 level)
    constexpr int CH = S >= 16 ? 1 : (S >= 8 ? 2 : (S >= 4 ? 4 : 8));
    static_assert(NV % CH == 0, "whole chunks");
    auto fill = [&](uint4(&v)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int i0 = 0; i0 < NV; i0 += CH) {
            if (i0 < nv) {
                float4 lo[CH][S], hi[CH][S];
                half8 rr[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int p = min(pl + PL * (i0 + j), HW - 1);
                    const float* src = src0 + (size_t)p * g.N;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        lo[j][s] = *reinterpret_cast<const float4*>(src + (size_t)s * slab);
                        hi[j][s] = *reinterpret_cast<const float4*>(src + (size_t)s * slab + 4);
                    }
                }
                if (g.res) {
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        rr[j] = *reinterpret_cast<const half8*>(g.res + (row0 + min(pl + PL * (i0 + j), HW - 1)) * g.ldr + n);
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    float a[8] = {lo[j][0].x, lo[j][0].y, lo[j][0].z, lo[j][0].w, hi[j][0].x, hi[j][0].y, hi[j][0].z, hi[j][0].w};
#pragma unroll
                    for (int s = 1; s < S; ++s) {   // fixed order
                        a[0] += lo[j][s].x; a[1] += lo[j][s].y; a[2] += lo[j][s].z; a[3] += lo[j][s].w;
                        a[4] += hi[j][s].x; a[5] += hi[j][s].y; a[6] += hi[j][s].z; a[7] += hi[j][s].w;
                    }
                    half8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float val = fmaf(a[k], g.alpha, bs[k]);
                        if (g.res) val += (float)rr[j][k];
                        o[k] = (half_t)val;
                    }
                    const int p = pl + PL * (i0 + j);
                    const bool valid = active && p < HW;
                    const uint4 packed = *reinterpret_cast<uint4*>(&o);
                    v[i0 + j] = valid ? packed : make_uint4(0u, 0u, 0u, 0u);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) v[i0 + j] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // the convolution's own output (unless gn_skip_c), in a loop of its own: stores inside the chunk loop above made hipcc keep
        // the whole v[] array in scratch memory
        if (!g.gn_skip_c && active) {
            half_t* __restrict__ cb = reinterpret_cast<half_t*>(g.C) + row0 * g.ldc + n;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int p = pl + PL * i;
                if (p < HW) *reinterpret_cast<uint4*>(cb + (size_t)p * g.ldc) = v[i];
            }
        }
    };
    gn_slab_body<NT, NV>(fill, g.gn_out + row0 * C + ch0, g.gn_gamma + ch0, g.gn_beta + ch0, HW, C, cpg, GB, g.gn_eps, g.gn_silu, gn_sm);
}

template <int S>
static int launch_finish_gn(const GemmArgs& g, hipStream_t st) {
    size_t lds = 0;
    const int GB = gn_slab_pick<256, 16>(g.rows_per_batch, g.N, g.gn_groups, &lds);
    static std::atomic<unsigned long long> configured{0};
    if (lds > 64 * 1024 && fd_first_on_device(&configured))
        FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_splitk_finish_gn<S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_splitk_finish_gn<S>, dim3(g.M / g.rows_per_batch, g.gn_groups / GB), dim3(256), lds, st, g);
    FD_CHECK_LAUNCH("k_splitk_finish_gn");
    return FD_OK;
}

extern "C" int fd_gemm_can_fuse_groupnorm(int M, int N, int rows_per_sample, int groups, int split_k) {
    if (M <= 0 || N <= 0 || rows_per_sample <= 0 || groups <= 0 || M % rows_per_sample != 0 || N % 8 != 0 || N % groups != 0) return 0;
    if (!(split_k == 2 || split_k == 4 || split_k == 8 || split_k == 16)) return 0;
    size_t lds = 0;
    return gn_slab_pick<256, 16>(rows_per_sample, N, groups, &lds) > 0 ? 1 : 0;
}

// --------------------------------------------------------------------------------------
static bool g_use_dma = getenv("FD_GEMM_NO_DMA") == nullptr;
static const int g_vae15 = 1;   // 256x256 tiles on the VAE widths (A/B closed in round 1: +19..33 %)
static int g_tap_fast = getenv("FD_CONV_TAPFAST") ? atoi(getenv("FD_CONV_TAPFAST")) : 1;   // 1: 256x320 tile, 2: every conv tile
static int g_pp = getenv("FD_GEMM_PP") ? atoi(getenv("FD_GEMM_PP")) : 1;   // 0: never pick the ping-pong tiles (A/B)
static int g_bias_lds = getenv("FD_GEMM_BIAS_LDS") ? atoi(getenv("FD_GEMM_BIAS_LDS")) : 1;
static int g_fast_epi = getenv("FD_GEMM_FAST_EPI") ? atoi(getenv("FD_GEMM_FAST_EPI")) : 1;   // 0: generic epilogue only (A/B)
// 0 = never, 1 = short-K GEMMs only (default), 2 = always
static int g_persist_mode = getenv("FD_GEMM_PERSIST") ? atoi(getenv("FD_GEMM_PERSIST")) : 1;

template <int BM, int BN, bool TRANS, bool CONV, int WM = 2, int NS = 2, int WN = 2, int EPI = 0>
static int launch_mode(GemmArgs& g, int batch, hipStream_t st) {
    g.tiles_m = fd_cdiv(g.M, BM);
    g.tiles_n = fd_cdiv(g.N, BN);
    // stages + 2 x (bias, bias2 / colsum) tiles + the tile's LayerNorm statistics when they come as partial sums (ln_tile_stats_to_lds)
    const size_t lds = NS * (size_t)(BM + BN) * 128 + 4 * BN * sizeof(float) + ((EPI == 5 || EPI == 6 || EPI == 7 || EPI == 13) ? BM * 2 * sizeof(float) : 0);
    dim3 grid(g.tiles_m * g.tiles_n, g.split_k, batch);
    // tensor extents for the buffer descriptors of the LDS-DMA loop (must fit 32 bits)
    const unsigned long long a_bytes =
        CONV ? 2ull * (((unsigned long long)(g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi - 1) * g.Cpix + g.Cin)
             : 2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K);
    const unsigned long long w_bytes = 2ull * ((unsigned long long)(g.N - 1) * g.ldw + g.K + g.K2);
    if (g_use_dma && a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull) {
        static std::atomic<unsigned long long> configured{0};
        if (lds > 64 * 1024 && fd_first_on_device(&configured)) {
            FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_dma<BM, BN, CONV, WM, TRANS, NS, WN, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if constexpr (BN != 320 && EPI != 13)   // the 256x320 tile has no persistent form (it would spill); nor has the transposed tail
                FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_dmap<BM, BN, CONV, WM, TRANS, WN, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        // Short K loops: persistent workgroups (<= 256 CUs x co-resident workgroups) walk the
        // tile list with the next tile's first K-tile prefetched under the epilogue.
        const int occ = (BM >= 256 || lds > 80 * 1024) ? 1 : (BM * BN >= 128 * 128) ? 2 : (BM == 128 ? 3 : 4);  // workgroups / CU
        const int slots = 256 * occ;
        const int nkt = (g.K + BK - 1) / BK / g.split_k;
        // (the persistent loop reads finished LayerNorm statistics only: partial sums run the one-tile kernel)
        const bool persistent = NS == 2 && BN != 320 && EPI != 13 && g.K2 == 0 && !g.phase && !g.ln_stats_out && g.ln_parts <= 1 && g.strideBias == 0 && (g_persist_mode == 2 ||
                                (g_persist_mode == 1 && nkt <= 20 && g.tiles_m * g.tiles_n > slots));
        if constexpr (EPI >= 1 && EPI <= 3) {
            // the persistent loop does not stage the per-sample bias: generic epilogue there
            if (persistent && g.bias2) return launch_mode<BM, BN, TRANS, CONV, WM, NS, WN, 0>(g, batch, st);
        }
        if (persistent) {
            if constexpr (BN != 320 && EPI != 13) {
                dim3 pgrid(g.tiles_m * g.tiles_n > slots ? slots : g.tiles_m * g.tiles_n, g.split_k, batch);
                hipLaunchKernelGGL((k_gemm_f16_dmap<BM, BN, CONV, WM, TRANS, WN, EPI>), pgrid, dim3(64 * WM * WN), lds,
                                   st, g, (unsigned)a_bytes, (unsigned)w_bytes);
            }
        } else {
            // (a flat split-K grid with one K slice per XCD -- GemmArgs.sk_flat -- measured neutral, profiles/r03_session_ab.txt
            // sec. 5: the A/B is closed and its switch removed; the kernel keeps the index path for the record)
            hipLaunchKernelGGL((k_gemm_f16_dma<BM, BN, CONV, WM, TRANS, NS, WN, EPI>), grid, dim3(64 * WM * WN), lds, st, g,
                               (unsigned)a_bytes, (unsigned)w_bytes);
        }
        FD_CHECK_LAUNCH("k_gemm_f16_dma");
        return FD_OK;
    }
    if (g.ln_stats) {
        fd_set_error("fd_gemm_f16: the LayerNorm fold needs the LDS-DMA path (tensor < 2 GiB, FD_GEMM_NO_DMA unset)");
        return FD_ESHAPE;
    }
    if (g.K2) {
        fd_set_error("fd_gemm_f16: the appended 1x1 phase (A2 / K2) needs the LDS-DMA path (tensor < 2 GiB, FD_GEMM_NO_DMA unset)");
        return FD_ESHAPE;
    }
    if constexpr (EPI != 0) {
        return launch_mode<BM, BN, TRANS, CONV, WM, NS, WN, 0>(g, batch, st);
    } else if constexpr (WM != 2 || WN != 2) {
        fd_set_error("fd_gemm_f16: 8-wave tiles need the LDS-DMA path (tensor < 2 GiB)");
        return FD_ESHAPE;
    } else {
        static std::atomic<unsigned long long> configured{0};
        if (lds > 64 * 1024 && fd_first_on_device(&configured))
            FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16<BM, BN, TRANS, CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_gemm_f16<BM, BN, TRANS, CONV>), grid, dim3(256), lds, st, g);
        FD_CHECK_LAUNCH("k_gemm_f16");
        return FD_OK;
    }
}

template <int BM, int BN, bool TRANS, int WM = 2, int NS = 2, int WN = 2, int EPI = 0>
static int launch(GemmArgs& g, int batch, hipStream_t st) {
    return g.mode == MODE_CONV ? launch_mode<BM, BN, TRANS, true, WM, NS, WN, EPI>(g, batch, st)
                               : launch_mode<BM, BN, TRANS, false, WM, NS, WN, EPI>(g, batch, st);
}

// Picks the lean epilogue (gemm_epilogue_fast) when every tile of the launch is full, the output
// rows are 16-byte aligned fp16, the biases are staged in LDS and the activation is one the lean
// form was instantiated for.  ALLOW: bit e set = EPI e exists for this tile (1 plain, 2 residual,
// 3 GEGLU, 5 LayerNorm fold, 6 LayerNorm fold + GEGLU -- the last two for linear GEMMs only);
// everything else runs the generic epilogue (which implements the same arithmetic at run time).
template <int BM, int BN, int WM, int NS, int WN, int ALLOW>
static int launch_epi(GemmArgs& g, int batch, hipStream_t st) {
    const bool full = g_fast_epi && g.split_k == 1 && !g.out_f32 && !g.trans_out && g.bias_lds &&
                      g.M % BM == 0 && g.N % BN == 0 && (g.ldc & 7) == 0 &&
                      (!g.bias2 || g.rows_per_batch % BM == 0);
    if (g.gn_part_out) {
        // GroupNorm partial sums of the output: the 256x320 tile spanning the row, inside one sample, lean epilogue
        if constexpr (BN == 320 && NS == 2) {
            if (full && g.act == FD_ACT_NONE && !g.ln_stats && !g.ln_stats_out && g.N == BN && g.rows_per_batch % BM == 0 && !g.phase && batch == 1 &&
                (!g.res || (g.ldr & 3) == 0)) {
                if (g.res) return launch<BM, BN, false, WM, NS, WN, 12>(g, batch, st);
                return launch<BM, BN, false, WM, NS, WN, 11>(g, batch, st);
            }
        }
        fd_set_error("fd_gemm_f16: gn_part_out needs full 320-wide tiles inside one sample and the lean plain / residual epilogue");
        return FD_ESHAPE;
    }
    if (g.ln_stats_out) {
        // row statistics of the output: only tiles that span the whole row (N == BN), lean epilogue
        if constexpr ((ALLOW & 256) != 0) {
            // (the 256x320 tile finalises its rows: N == BN; the 160-wide tiles write per-n-tile partial sums)
            if (full && (BN != 320 || g.N == BN) && g.act == FD_ACT_NONE && g.mode != MODE_CONV && !g.ln_stats && !g.bias2) {
                if (g.res && (g.ldr & 3) == 0) return launch_mode<BM, BN, false, false, WM, NS, WN, 9>(g, batch, st);
                if (!g.res) return launch_mode<BM, BN, false, false, WM, NS, WN, 8>(g, batch, st);
            }
        }
        fd_set_error("fd_gemm_f16: ln_stats_out needs full tiles of a tile shape with a statistics epilogue, a plain or residual linear GEMM");
        return FD_ESHAPE;
    }
    if (full && g.ln_stats) {
        if (g.mode != MODE_CONV && !g.res) {
            if constexpr ((ALLOW & 64) != 0 && (BN / WN / 16) % 2 == 0) {
                if (g.act == FD_ACT_GEGLU) return launch_mode<BM, BN, false, false, WM, NS, WN, 6>(g, batch, st);
            }
            if constexpr ((ALLOW & 32) != 0) {
                if (g.act == FD_ACT_NONE) return launch_mode<BM, BN, false, false, WM, NS, WN, 5>(g, batch, st);
            }
        }
        return launch_mode<128, 128, false, false, 2, 2, 2, 7>(g, batch, st);
    }
    if (g.ln_stats) return launch_mode<128, 128, false, false, 2, 2, 2, 7>(g, batch, st);
    if (full) {
        if constexpr ((ALLOW & 8) != 0 && (BN / WN / 16) % 2 == 0) {
            if (g.act == FD_ACT_GEGLU) return launch<BM, BN, false, WM, NS, WN, 3>(g, batch, st);
        }
        if constexpr ((ALLOW & 4) != 0) {
            if (g.act == FD_ACT_NONE && g.res && (g.ldr & 3) == 0) return launch<BM, BN, false, WM, NS, WN, 2>(g, batch, st);
        }
        if constexpr ((ALLOW & 2) != 0) {
            if (g.act == FD_ACT_NONE && !g.res) return launch<BM, BN, false, WM, NS, WN, 1>(g, batch, st);
        }
    }
    return launch<BM, BN, false, WM, NS, WN, 0>(g, batch, st);
}

// 1 when fd_gemm_f16 can honour fd_gemm_desc.ln_stats_out for an [M][N] fp16 output with row stride ldc and
// (ldr > 0) a residual of row stride ldr: the row-complete 256x320 tile on the LDS-DMA path with the lean
// epilogue and LDS-staged biases (FD_GEMM_FAST_EPI / FD_GEMM_BIAS_LDS / FD_GEMM_NO_DMA are A/B switches that
// take those away).  Callers that get 0 run fd_ln_row_stats_f16 on the output instead.
// fraction of the CU slots busy over the launch's rounds (slots = 256 CUs x workgroups per CU)
static const int g_t23 = 1;   // the 288x160 tile for row counts 9 * 2^k (A/B closed in round 3: c4 / c5 +16..18 %)
static double fd_round_eff(long long tiles, int slots) {
    return (double)tiles / (double)((long long)slots * ((tiles + slots - 1) / slots));
}

// How fd_gemm_f16 honours ln_stats_out for an [M][N] output: 0 not at all, 1 finished pairs (the row-complete 256x320
// tile), k > 1 slabs of per-n-tile partial sums from a 160-wide tile; *tile = the tile shape it will use.
static int fd_stats_plan(int M, int N, int* tile, int batch = 1) {
    if (N < 320 || N % 160 != 0) return 0;
    // rows = 9 x 2^k (768x768 images): the 288-row tile where it fills the rounds better than the 256-row shapes
    const bool ok23 = g_t23 && M % 288 == 0 && M >= 1152;
    if (N == 320 && M % 256 == 0) {
        if (ok23 && fd_round_eff((long long)(M / 288) * 2, 256) > fd_round_eff(M / 256, 256) + 0.08) {
            *tile = 23;
            return 2;
        }
        *tile = 16;
        return 1;
    }
    if (ok23 && (M % 128 != 0 ||
                 fd_round_eff((long long)(M / 288) * (N / 160), 256) >
                     (M % 256 == 0 && M > 4096 ? fd_round_eff((long long)(M / 256) * (N / 160), 256)
                                                : fd_round_eff((long long)(M / 128) * (N / 160), 512)) + 0.08)) {
        *tile = 23;
        return N / 160;
    }
    if (N == 320 || M % 128 != 0) return 0;
    *tile = (M % 256 == 0 && (long long)M * batch > 4096) ? 13 : 12;   // (the slab count does not depend on the tile)
    return N / 160;
}

extern "C" int fd_gemm_can_emit_row_stats(int M, int N, int K, int ldc, int ldr) {
    if (!(g_use_dma && g_fast_epi && g_bias_lds)) return 0;
    if (M <= 0 || K <= 0 || K % 8 != 0) return 0;
    if (ldc < N || (ldc & 7) != 0 || (ldr != 0 && (ldr < N || (ldr & 3) != 0))) return 0;
    if (2ull * ((unsigned long long)(M - 1) * (unsigned long long)ldc + N) >= 0x7fffffffull) return 0;
    int tile = 0;
    return fd_stats_plan(M, N, &tile);   // slabs of partial sums the caller must provide and finalise (1: finalised in place)
}

static int gemm_impl(const fd_gemm_desc* d, void* stream, int* choice);

// Rows of the tile (= rows per chunk of the partial sums) when the launch of `g` on (tile, split) can emit GroupNorm partial sums
// of its output (fd_gemm_desc.gn_part_out): a tile that spans the row (N == 320: the 256x320 tiles 16 / 30, the 128x320 tile 32) and lies
// inside one sample, lean plain / residual epilogue; 0 otherwise.  The launchers check the same conditions.
static int gn_parts_bm(const GemmArgs& g, int tile, int split, int batch) {
    const int bm = (tile == 16 || tile == 30) ? 256 : (tile == 32 ? 128 : 0);
    if (!bm || split != 1 || batch != 1 || g.N != 320 || !g_fast_epi || !g.bias_lds || !g_use_dma) return 0;
    if (g.out_f32 || g.trans_out || g.act != FD_ACT_NONE || g.ln_stats || g.ln_stats_out || g.phase || g.gn_out) return 0;
    if ((g.ldc & 7) || (g.res && (g.ldr & 3)) || g.M % bm || g.rows_per_batch % bm || g.M % g.rows_per_batch) return 0;
    if (g.bias2 && g.rows_per_batch % bm) return 0;
    if (g.gn_groups <= 0 || g.gn_groups > 128 || g.N % g.gn_groups || ((g.N / g.gn_groups) & 1)) return 0;
    return bm;
}

extern "C" int fd_gemm_gn_parts_chunks(const fd_gemm_desc* d) {
    int choice[3] = {-1, 0, 0};
    if (!d || gemm_impl(d, nullptr, choice) != FD_OK || choice[2] <= 0) return 0;
    const int rows = d->rows_per_sample > 0 ? d->rows_per_sample : d->M;
    return rows / choice[2];
}

extern "C" int fd_gemm_f16(const fd_gemm_desc* d, void* stream) {
    if (fd_plan_recording() && d) {
        const fd_gemm_desc dc_ = *d;
        fd_plan_push([dc_](void* fd_s_) -> int { return fd_gemm_f16(&dc_, fd_s_); });
    }
    return gemm_impl(d, stream, nullptr);
}

// The tile id and split-K factor fd_gemm_f16 would launch `d` with (after the same argument checks), without launching anything:
// host logic only, no HIP call -- usable without a device (tests/test_gemm_rule_table.py pins the rule's choices on the launches
// of the bench forward).  Pointers are only tested for NULL / alignment.
extern "C" int fd_gemm_plan(const fd_gemm_desc* d, int* tile, int* split_k) {
    FD_CHECK_ARG(tile && split_k, FD_EINVAL, "fd_gemm_plan: null output pointer");
    int choice[2] = {0, 0};
    const int rc = gemm_impl(d, nullptr, choice);
    *tile = choice[0];
    *split_k = choice[1];
    return rc;
}

static int gemm_impl(const fd_gemm_desc* d, void* stream, int* choice) {
    FD_CHECK_ARG(d && d->A && d->W && (d->C || (d->gn_out && d->gn_skip_c)), FD_EINVAL, "fd_gemm_f16: null pointer");
    FD_CHECK_ARG(d->M > 0 && d->N > 0 && d->K > 0, FD_EINVAL, "fd_gemm_f16: M/N/K must be > 0");
    FD_CHECK_ARG(d->K % 8 == 0 && d->ldw % 8 == 0, FD_ESHAPE,
                 "fd_gemm_f16: K=%d and ldw=%d must be multiples of 8", d->K, d->ldw);
    FD_CHECK_ARG(((uintptr_t)d->A | (uintptr_t)d->W | (uintptr_t)d->C) % 16 == 0, FD_ESHAPE,
                 "fd_gemm_f16: pointers must be 16-byte aligned");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const half_t*)d->A;
    g.W = (const half_t*)d->W;
    g.C = d->C;
    g.bias = d->bias;
    g.bias2 = d->bias2;
    g.res = (const half_t*)d->residual;
    g.M = d->M; g.N = d->N; g.K = d->K;
    g.lda = d->lda; g.ldw = d->ldw; g.ldc = d->C ? d->ldc : d->N; g.ldr = d->ldr;
    g.ldb2 = d->ld_bias2 > 0 ? d->ld_bias2 : d->N;
    g.strideA = d->batch_stride_a; g.strideW = d->batch_stride_w;
    g.strideC = d->batch_stride_c; g.strideRes = d->batch_stride_res;
    g.strideBias = d->batch_stride_bias;
    g.rows_per_batch = d->rows_per_sample > 0 ? d->rows_per_sample : d->M;
    g.act = d->act; g.out_f32 = d->out_f32; g.trans_out = d->trans_out;
    g.strideT = d->trans_sample_stride; g.ldt = d->trans_ld;
    g.alpha = d->alpha == 0.f ? 1.f : d->alpha;
    const int batch = d->batch > 0 ? d->batch : 1;
    if (d->conv) {
        g.mode = MODE_CONV;
        g.Hi = d->in_h; g.Wi = d->in_w; g.Cin = d->in_c; g.Ho = d->out_h; g.Wo = d->out_w;
        g.KW = d->kw; g.stride = d->stride; g.pad_t = d->pad_t; g.pad_l = d->pad_l;
        g.up = d->upsample2x == 1;
        g.phase = d->upsample2x == 2;
        FD_CHECK_ARG(d->in_c % BK == 0, FD_ESHAPE,
                     "fd_gemm_f16: conv needs Cin %% 64 == 0 (got %d); use fd_im2col_f16", d->in_c);
        // the input may be a column slice of a wider NHWC matrix (a skip tensor living in its concat buffer): lda = pixel stride
        g.Cpix = d->lda > 0 ? d->lda : d->in_c;
        FD_CHECK_ARG(g.Cpix >= d->in_c && g.Cpix % 8 == 0 && (uintptr_t)d->A % 16 == 0, FD_ESHAPE,
                     "fd_gemm_f16: conv input pixel stride lda=%d must be 0 (= Cin) or >= Cin=%d and a multiple of 8, A 16-byte aligned", d->lda, d->in_c);
        FD_CHECK_ARG(d->K == d->kh * d->kw * d->in_c, FD_EINVAL, "fd_gemm_f16: K != kh*kw*Cin");
        FD_CHECK_ARG(d->M % (d->out_h * d->out_w) == 0, FD_EINVAL,
                     "fd_gemm_f16: M is not a multiple of out_h*out_w");
        if (g.phase) {
            // nearest-2x upsample + 3x3 conv as four 2x2 convolutions of the LOW-resolution input, one per output-pixel
            // parity (batch = 4 = blockIdx.z, weights pre-summed per parity): 4/9 of the MACs.  Only through the lean
            // plain epilogue (it maps GEMM rows to the interleaved output pixels).
            FD_CHECK_ARG(batch == 4 && d->kh == 2 && d->kw == 2 && d->stride == 1 && d->out_h == d->in_h && d->out_w == d->in_w &&
                             d->batch_stride_a == 0 && d->batch_stride_c == 0 && !d->out_f32 && !d->residual && !d->bias2 &&
                             d->act == FD_ACT_NONE && !d->trans_out && d->ldc % 8 == 0 &&
                             g_fast_epi && g_bias_lds && g_use_dma,
                         FD_ESHAPE, "fd_gemm_f16: upsample2x == 2 needs batch 4, a 2x2 kernel, stride 1, out = in size, plain fp16 "
                                    "output and the lean epilogue on the LDS-DMA path");
        }
    } else {
        FD_CHECK_ARG(d->lda % 8 == 0, FD_ESHAPE, "fd_gemm_f16: lda=%d must be a multiple of 8",
                     d->lda);
    }
    if (d->K2 > 0) {
        // appended phase: C += A2 W[:, K:K+K2]^T inside the same K loop
        FD_CHECK_ARG(d->A2 && d->K2 % BK == 0 && d->K % BK == 0 && d->lda2 % 8 == 0 && d->lda2 >= d->K2 &&
                         (uintptr_t)d->A2 % 16 == 0 && d->ldw >= d->K + d->K2 && batch == 1 && !d->trans_out && !d->ln_stats,
                     FD_ESHAPE, "fd_gemm_f16: A2 / K2 need K %% 64 == 0, K2 %% 64 == 0, lda2 %% 8 == 0, ldw >= K + K2, "
                                "16-byte aligned A2, no batch / transposed store / LayerNorm fold");
        const unsigned long long a2b = 2ull * ((unsigned long long)(d->M - 1) * d->lda2 + d->K2);
        FD_CHECK_ARG(a2b < 0x7fffffffull, FD_ESHAPE, "fd_gemm_f16: A2 >= 2 GiB");
        g.A2 = (const half_t*)d->A2; g.lda2 = d->lda2; g.K2 = d->K2; g.a2_bytes = (unsigned)a2b;
    }
    if (g.act == FD_ACT_GEGLU)
        FD_CHECK_ARG(d->N % 32 == 0 && !d->trans_out && !d->out_f32 && !d->residual, FD_ESHAPE,
                     "fd_gemm_f16: GEGLU needs N %% 32 == 0, fp16 output, no residual");
    if (!d->trans_out && !(g.act == FD_ACT_GEGLU))
        FD_CHECK_ARG(d->ldc % 4 == 0 && (d->N % 4 == 0 || true), FD_ESHAPE,
                     "fd_gemm_f16: ldc must be a multiple of 4");
    if (g.bias) FD_CHECK_ARG((uintptr_t)g.bias % 16 == 0, FD_ESHAPE, "fd_gemm_f16: bias align");
    if (g.res) FD_CHECK_ARG(d->ldr % 4 == 0, FD_ESHAPE, "fd_gemm_f16: ldr must be a multiple of 4");
    if (d->residual_rows) {
        FD_CHECK_ARG(d->residual_rows > 0 && g.res && !d->conv && batch == 1 && !d->trans_out && g.M % d->residual_rows == 0 &&
                         d->residual_rows % 32 == 0, FD_ESHAPE,
                     "fd_gemm_f16: residual_rows=%d needs a residual, a linear GEMM, batch 1 and M=%d a multiple of it (itself a multiple of 32)",
                     d->residual_rows, g.M);
        g.res_rows = d->residual_rows;
    }

    if (d->ln_stats) {
        // LayerNorm fold: C = act(rstd_m (A W'^T)[m][n] - rstd_m mean_m colsum_n + bias_n), W' = W diag(gamma)
        FD_CHECK_ARG(d->ln_colsum && !d->conv && !d->bias2 && !d->residual && !d->out_f32 && batch == 1 &&
                         (d->act == FD_ACT_NONE || d->act == FD_ACT_GEGLU) && (d->alpha == 0.f || d->alpha == 1.f),
                     FD_EINVAL, "fd_gemm_f16: ln_stats needs ln_colsum, a linear GEMM, no bias2 / residual / fp32 "
                                "output, act NONE or GEGLU, alpha 1");
        FD_CHECK_ARG(((uintptr_t)d->ln_stats % 16 == 0) && ((uintptr_t)d->ln_colsum % 16 == 0), FD_ESHAPE,
                     "fd_gemm_f16: ln_stats / ln_colsum must be 16-byte aligned");
        // transposed store: a lane folds 4 consecutive rows with two 16-byte statistics loads (row block clamped
        // to M - 4): a ragged last block would shift the statistics onto the wrong rows
        FD_CHECK_ARG(!d->trans_out || d->M % 4 == 0, FD_ESHAPE,
                     "fd_gemm_f16: ln_stats with trans_out needs M %% 4 == 0 (got M=%d)", d->M);
        g.ln_stats = d->ln_stats;
        g.bias2 = d->ln_colsum;   // one row for every sample: row stride 0
        g.ldb2 = 0;
        if (d->ln_stats_parts > 0) {
            // the statistics come as the producer's partial slabs: each tile finalises its rows into LDS (ln_tile_stats_to_lds)
            FD_CHECK_ARG((d->ln_stats_parts == 2 || d->ln_stats_parts == 4 || d->ln_stats_parts == 8) && d->ln_stats_rows >= d->M && g_use_dma,
                         FD_ESHAPE, "fd_gemm_f16: ln_stats_parts must be 2, 4 or 8 with ln_stats_rows >= M, on the LDS-DMA path");
            g.ln_parts = d->ln_stats_parts;
            g.ln_rows = d->ln_stats_rows;
            g.ln_inv_n = 1.0f / (float)d->K;
            g.ln_eps_in = d->ln_fold_eps > 0.f ? d->ln_fold_eps : 1e-5f;
        }
    }
    if (d->trans_n0 > 0) {
        // transposed tail: one launch for the stacked [q | k | v] projection of a self-attention (see the header)
        FD_CHECK_ARG(d->C2 && d->ln_stats && !d->conv && !d->trans_out && d->act == FD_ACT_NONE && !d->residual && !d->out_f32 && batch == 1 &&
                         !d->K2 && !d->ln_stats_out && !d->gn_out && !d->gn_part_out,
                     FD_EINVAL, "fd_gemm_f16: trans_n0 needs C2 and a plain LayerNorm-fold linear GEMM (act NONE, no residual / batch / appended operand)");
        FD_CHECK_ARG(d->M % 128 == 0 && d->N % 160 == 0 && d->trans_n0 % 160 == 0 && d->trans_n0 < d->N && g.rows_per_batch % 32 == 0 &&
                         d->M % g.rows_per_batch == 0 && d->trans_ld % 8 == 0 && d->trans_ld >= g.rows_per_batch && d->trans_sample_stride % 8 == 0 &&
                         (uintptr_t)d->C2 % 16 == 0 && (d->ldc & 7) == 0 && g_use_dma && g_fast_epi && g_bias_lds,
                     FD_ESHAPE, "fd_gemm_f16: trans_n0 needs M %% 128 == 0, N and trans_n0 multiples of 160, rows_per_sample %% 32 == 0 and | M, "
                                "trans_ld %% 8 == 0 and >= rows_per_sample, ldc %% 8 == 0, 16-byte aligned C2, the LDS-DMA path with the lean epilogue");
        const unsigned long long ab = 2ull * ((unsigned long long)(d->M - 1) * d->lda + d->K), wb = 2ull * ((unsigned long long)(d->N - 1) * d->ldw + d->K);
        FD_CHECK_ARG(ab < 0x7fffffffull && wb < 0x7fffffffull, FD_ESHAPE, "fd_gemm_f16: trans_n0: operands >= 2 GiB");
        g.tr_n0 = d->trans_n0;
        g.C2 = (half_t*)d->C2;
        g.split_k = 1;
        g.bias_lds = g_bias_lds;
        if (choice) {
            choice[0] = 9;
            choice[1] = 1;
            return FD_OK;
        }
        hipStream_t st2 = (hipStream_t)stream;
        const double fl = 2.0 * (double)d->M * d->N * d->K;
        fd_prof_begin(FD_FAMILY_GEMM, st2, fl, fl, fd_tag(14u, g.M, g.N, g.K, g.tr_n0));
        const int rc2 = launch_mode<128, 160, false, false, 4, 2, 2, 13>(g, 1, st2);
        fd_prof_end(FD_FAMILY_GEMM, st2);
        return rc2;
    }
    if (d->gn_out) {
        // GroupNorm(+SiLU) of the output inside the split-K finish (k_splitk_finish_gn)
        FD_CHECK_ARG(d->gn_gamma && d->gn_beta && d->gn_groups > 0 && d->N % d->gn_groups == 0 && d->N % 8 == 0 && d->M % g.rows_per_batch == 0,
                     FD_EINVAL, "fd_gemm_f16: gn_out needs gn_gamma, gn_beta, gn_groups | N, N %% 8 == 0 and whole samples (rows_per_sample | M)");
        FD_CHECK_ARG(!d->out_f32 && !d->trans_out && d->act == FD_ACT_NONE && batch == 1 && !d->ln_stats && !d->ln_stats_out &&
                         (uintptr_t)d->gn_out % 16 == 0 && (g.ldc & 7) == 0 && (!g.res || (g.ldr & 7) == 0) && (g.ldb2 & 3) == 0 &&
                         (!g.bias2 || (uintptr_t)g.bias2 % 16 == 0) && (!g.res || (uintptr_t)g.res % 16 == 0),
                     FD_ESHAPE, "fd_gemm_f16: gn_out needs a plain fp16 output (act NONE, batch 1, no LayerNorm fold), ldc / ldr %% 8 == 0, 16-byte aligned gn_out / bias2 / residual");
        size_t gn_lds = 0;
        g.gn_gb = gn_slab_pick<256, 16>(g.rows_per_batch, d->N, d->gn_groups, &gn_lds);
        FD_CHECK_ARG(g.gn_gb > 0, FD_ESHAPE, "fd_gemm_f16: gn_out: a [%d][%d / %d groups] slab does not fit the finish kernel (fd_gemm_can_fuse_groupnorm)",
                     g.rows_per_batch, d->N, d->gn_groups);
        g.gn_out = (half_t*)d->gn_out; g.gn_gamma = d->gn_gamma; g.gn_beta = d->gn_beta;
        g.gn_groups = d->gn_groups; g.gn_silu = d->gn_silu; g.gn_skip_c = d->gn_skip_c;
        g.gn_eps = d->gn_eps > 0.f ? d->gn_eps : 1e-5f;
    }
    if (d->sk_sync) {
        FD_CHECK_ARG(!d->gn_out && (uintptr_t)d->sk_sync % 4 == 0, FD_EINVAL, "fd_gemm_f16: sk_sync excludes gn_out (the GroupNorm would read an unfinished tile)");
        g.sk_sync = (unsigned*)d->sk_sync;
    }
    if (d->gn_part_out || (choice && choice[0] == -1)) {
        g.gn_groups = d->gn_groups;
        g.gn_part_out = d->gn_part_out;
    }
    hipStream_t st = (hipStream_t)stream;
    // priced at the ALGORITHMIC work of the op it implements: the phase-decomposed upsample convolution (batch 4, K = 4 Cin)
    // stands for a 3x3 convolution over the 4 M upsampled pixels (K = 9 Cin), of which it executes 4/9 of the MACs
    const double flops_exec = 2.0 * (double)d->M * d->N * ((double)d->K + d->K2) * batch;
    const double flops = (d->conv && d->upsample2x == 2 ? 2.25 : 1.0) * flops_exec;
    g.split_k = 1;
    g.bias_lds = g_bias_lds;
    g.ws = (float*)d->workspace;
    int rc;
    if (d->trans_out) {
        // 128x160 with 8 waves where 160 | N and the rows fill the chip: -16..18 % on the 64x64 / 32x32
        // self-attention V projections (tools/ab_vt.py; 256x160 / 16 waves is no better, 16x16 maps tie).
        // FD_GEMM_VT_TILE=0: the 128x64 / 4-wave tile everywhere (A/B)
        static const int vt_tile = getenv("FD_GEMM_VT_TILE") ? atoi(getenv("FD_GEMM_VT_TILE")) : 9;
        const bool vt160 = vt_tile && g.N % 160 == 0 && batch == 1 && g.M >= 8192 &&
                           2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K) < 0x7fffffffull;
        if (choice) {
            choice[0] = vt160 ? 9 : 3;      // (the transposed-store forms of the 128x160 / 128x64 tiles)
            choice[1] = 1;
            return FD_OK;
        }
        fd_prof_begin(FD_FAMILY_GEMM, st, flops, flops_exec, fd_tag(10u, g.M, g.N, g.K, vt160 ? 9 : 3));
        if (vt160)
            rc = g.ln_stats ? launch_mode<128, 160, true, false, 4, 2, 2, 7>(g, batch, st) : launch_mode<128, 160, true, false, 4, 2, 2, 0>(g, batch, st);
        else
        rc = g.ln_stats ? launch_mode<128, 64, true, false, 2, 2, 2, 7>(g, batch, st) : launch<128, 64, true>(g, batch, st);
        fd_prof_end(FD_FAMILY_GEMM, st);
        return rc;
    }
    // ---- tile / split-K selection (deterministic; rules fitted to an exhaustive sweep of
    // every (tile, split_k) over all GEMM shapes of the SD1.5 UNet + VAE on MI355X,
    // tools/sweep_gemm.py, profiles/r01_gemm_sweep.txt):
    //  * 128x160 has the best fragment reuse (up to 860 TFLOP/s) whenever 160 | N (all UNet
    //    widths); 128x128 otherwise (VAE widths);
    //  * short-K GEMMs (the transformer projections) are latency- not MFMA-bound: 128x64 with
    //    3 workgroups per CU wins;
    //  * when the tile count cannot fill 2 workgroups on each of the 256 CUs, split K until it
    //    does (fp32 slabs + fixed-order finish kernel, so results stay deterministic).
    const bool geglu = g.act == FD_ACT_GEGLU;
    const int nk_all = (g.K + g.K2 + BK - 1) / BK;
    const bool n160 = (g.N % 160 == 0) && !geglu;
    const long long tiles_wide =
        (long long)fd_cdiv(g.M, 128) * (n160 ? fd_cdiv(g.N, 160) : fd_cdiv(g.N, 128)) * batch;
    int best_tile, best_split = 1;
    if (geglu) {
        // GEGLU (K = C, N = 8C): 16 waves on 256x128 beat 8 on 128x128 by 3-10 %
        best_tile = (g.M <= 1024) ? 11 : 14;
        // 256x256 with 64x64 wave tiles: fewer LDS reads per MAC; needs whole rounds of tiles
        const long long t15 = (long long)fd_cdiv(g.M, 256) * fd_cdiv(g.N, 256);
        if (g.N % 256 == 0 && (t15 % 256 == 0 || t15 >= 512 || (t15 >= 128 && t15 <= 256))) best_tile = 15;
        // K >= 1280 (the 16x16 and 8x8 levels, >= 20 K-tiles per tile): the ping-pong 256x256 tile, 3-8 % faster than the persistent
        // 16-wave tile; at K = 640 it is 3-5 % slower, at K = 320 (5 K-tiles: un-overlapped prologue + epilogue) 15 % slower
        if (g_pp && g_use_dma && g_fast_epi && g_bias_lds && g.K >= 1280 && g.M % 256 == 0 && g.N % 256 == 0 && batch == 1 && !d->tile) {
            g.split_k = 1;
            if (fd_gemm_pp_ok(g, batch, 31)) best_tile = 31;
        }
    } else if (g.N <= 64) {
        best_tile = (g.M <= 64) ? 4 : 3;
        // the UNet's / VAE's output convolutions (N = 4 / 3, K = 9 C): 16 waves on 128x64; with <= 2 workgroups per CU two K slices
        // halve the serial K loop (conv_out of the bench forward: 41 vs 54 us, profiles/r05_gemm_sweep_vae.txt)
        if (g.mode == MODE_CONV && g.M >= 32768 && nk_all >= 32 && batch == 1) {
            best_tile = 11;
            if (fd_cdiv(g.M, 128) <= 512 && g.N % 4 == 0 && g.ws && (size_t)2 * g.M * g.N * 4 <= (size_t)d->workspace_bytes) best_split = 2;
        }
    } else if (g.K <= 640 || (g.K <= 1280 && tiles_wide < 512)) {
        // short K loops (transformer projections, GEGLU, 1x1 shortcuts) are latency-bound:
        // many-wave tiles win by 12-23 % over 4-wave 128x64 (16 waves on 128x160 for few rows,
        // on 256x160 otherwise; 8 waves on 128x128 where 160 does not divide N or for GEGLU);
        // tiny row counts keep the 64x64 tile
        best_tile = (g.M <= 1024) ? 4 : (!n160 ? 10 : (g.M <= 4096 ? 12 : 13));
    } else if (n160) {
        // large K, UNet widths: 256x160 with 16 waves (32x80 wave tiles, one workgroup per CU,
        // up to 1.23 PFLOP/s); split K until ~every CU has a workgroup
        best_tile = 13;
        // 256x320 with 64x80 wave tiles (36 % fewer LDS fragment reads and 31 % less LDS-DMA per
        // MAC than 256x160) wins 7-14 % when its tiles fill the 256 CUs in whole rounds
        if (g.N % 320 == 0 && ((long long)fd_cdiv(g.M, 256) * (g.N / 320) * batch) % 256 == 0) best_tile = 16;
        const long long tiles = (long long)fd_cdiv(g.M, 256) * fd_cdiv(g.N, 160) * batch;
        if (batch == 1 && g.N % 4 == 0 && g.ws) {
            while (tiles * best_split < 224 && best_split < 16 && nk_all / (best_split * 2) >= 4 &&
                   (size_t)(best_split * 2) * g.M * g.N * 4 <= (size_t)d->workspace_bytes)
                best_split *= 2;
        }
    } else {
        // large K, other widths (VAE: 128/256/512): 256x128 with 16 waves when not split
        // (2-7 % over 128x128 with 8)
        best_tile = 1;
        long long tiles = tiles_wide;
        int target = 448;
        if (g.M <= 2048 && !geglu) {
            best_tile = 6;
            tiles = (long long)fd_cdiv(g.M, 256) * fd_cdiv(g.N, 128) * batch;
            target = 224;
        }
        if (!geglu && batch == 1 && g.N % 4 == 0 && g.ws) {
            while (tiles * best_split < target && best_split < 16 &&
                   nk_all / (best_split * 2) >= 4 &&
                   (size_t)(best_split * 2) * g.M * g.N * 4 <= (size_t)d->workspace_bytes)
                best_split *= 2;
        }
        if (best_split == 1 && best_tile == 1) best_tile = (g.M >= 4096) ? 14 : 10;
        // VAE widths 256 / 512 on large maps: 256x256 with 64x64 wave tiles
        // (36 % fewer LDS fragment reads per MAC than 256x128: +19..33 % on the 128^2..256^2 maps)
        const long long t15 = (long long)fd_cdiv(g.M, 256) * (g.N / 256) * batch;
        if (best_split == 1 && best_tile == 14 && g.N % 256 == 0 && g_vae15 && (t15 >= 1024 || t15 % 256 == 0))
            best_tile = 15;
    }
    // 16x16-resolution transformer GEMMs (M = 4096, N = 1280, K >= 1280): 256 tiles of 128x160, one per
    // CU, each a 20..80-deep K loop whose DMA round trip (not the MFMAs) sets the pace -- a third LDS
    // stage (two K-tiles in flight) gives -10 % at K = 1280, -22 % at K = 2560 (vs 256x160 + split-K 2),
    // -3 % at K = 5120 (tools/ab_ns3.py).  Convolutions and M >= 16 k lose with it (one workgroup per CU).
    // ... unless 256x160 tiles already give every CU one tile (N = 2560, the fused q|k projection: 33 vs 41.5 us)
    // (only where its own 128x160 tiles fill most of the chip: swept at N >= 1280; a narrow N -- e.g. the batch-1
    // 4096x320x1280 FF-out -- would leave 64 workgroups on 256 CUs with split-K switched off)
    if (g.mode != MODE_CONV && n160 && g.M > 2048 && g.M <= 4096 && g.K >= 1280 && batch == 1 &&
        (long long)fd_cdiv(g.M, 256) * (g.N / 160) < 256 && (long long)fd_cdiv(g.M, 128) * (g.N / 160) >= 200) {
        best_tile = 20;
        best_split = 1;
    }
    // the parity-decomposed upsample convolution of a small map (16 x 8 x 8 -> 16 x 16: four slices of 1024 rows, no split-K in this mode):
    // 256-row tiles give 128 workgroups, the 3-stage 128x160 tile 256 -- 68 vs 90 us (tools/ab_up8.py; the fused-upsample form it replaces: 133 us)
    if (g.phase && n160 && g.M % 128 == 0 && g.K >= 1280 && (long long)fd_cdiv(g.M, 256) * (g.N / 160) * batch < 200 &&
        (long long)(g.M / 128) * (g.N / 160) * batch >= 200) {
        best_tile = 20;
        best_split = 1;
    }
    // Row counts of the form 9 x 2^k (768x768 images: 96x96 / 48x48 / 24x24 / 12x12 latents) leave the 256-row tiles
    // with 288-, 72- or 18-tile columns -- 2.25 rounds on 256 CUs, the last one a quarter full.  288 x 160 (12 waves,
    // 48x80 wave tiles, otherwise the 256x160 kernel) divides those rows exactly: 73728 x 320 -> 512 tiles = 2 rounds.
    if (g_t23 && n160 && g.M % 288 == 0 && g.M >= 1152 && (best_tile == 13 || best_tile == 12 || best_tile == 20)) {
        const auto eff = [](long long t) { return (double)t / (double)(256 * ((t + 255) / 256)); };
        const int bm_cur = best_tile == 13 ? 256 : 128;
        const long long t_cur = (long long)fd_cdiv(g.M, bm_cur) * (g.N / 160) * batch * best_split;
        long long t23 = (long long)(g.M / 288) * (g.N / 160) * batch;
        int split23 = 1;
        if (best_split > 1 || (g.K + g.K2 > 1280 && batch == 1 && g.N % 4 == 0 && g.ws)) {
            while (t23 * split23 < 224 && split23 < 16 && nk_all / (split23 * 2) >= 4 &&
                   (size_t)(split23 * 2) * g.M * g.N * 4 <= (size_t)d->workspace_bytes)
                split23 *= 2;
        }
        // (128-row tiles run two workgroups per CU: their rounds are 512 slots)
        const double e_cur = bm_cur == 256 ? eff(t_cur) : (double)t_cur / (double)(512 * ((t_cur + 511) / 512));
        if (eff(t23 * split23) > e_cur + 0.08) {
            best_tile = 23;
            best_split = split23;
        }
    }
    // ---- the deep-pipelined ping-pong tiles (gemm_pp.hip) where they measured faster than every choice above on one box
    // (tools/ab_pp.py, profiles/r05_ab_pp.txt): stride-1 3x3 convolutions without an appended phase on 256x320 tiles -- split
    // over K until ~every CU has a workgroup -- or 128x320 tiles (+7..18 %), and the long-K linears of the feed-forward output
    // (+13..14 %).  Convolutions with the ResBlock shortcut appended (their A2 rows stream from HBM inside a few K-tiles),
    // GEGLU and the short-K projections stay on the 2-barrier kernels (equal or faster there); FD_GEMM_PP=0 switches the rule off.
    if (g_pp && g_use_dma && g_fast_epi && g_bias_lds && !geglu && !d->ln_stats_out && (!g.ln_stats || g.mode != MODE_CONV) && (batch == 1 || g.phase) && !g.out_f32 &&
        g.act == FD_ACT_NONE && g.N % 320 == 0 && g.M % 128 == 0 && (g.ldc & 7) == 0 && (!g.res || (g.ldr & 3) == 0)) {
        // (the lean epilogue takes a per-sample bias only from tiles that lie in one sample; split-K launches leave it to the finish kernel)
        const bool b2ok = !g.bias2 || g.rows_per_batch % 256 == 0;
        // (the parity-decomposed upsample convolution is four launch slices -- blockIdx.z -- of the same tile grid: no split-K there)
        const long long t30 = (g.M % 256 == 0 ? (long long)(g.M / 256) * (g.N / 320) : 0) * batch, t32 = (long long)(g.M / 128) * (g.N / 320) * batch;
        int tile = 0, split = 1;
        if (g.mode == MODE_CONV && g.phase) {
            if (t30 >= 200 && fd_round_eff(t30, 256) >= 0.85 && nk_all >= 32) tile = 30;     // tools/.. rule guard: 168 vs 182 us at 32x32 -> 64x64
        } else if (g.mode == MODE_CONV && !g.K2 && nk_all >= 32) {
            if (t30 >= 200 && fd_round_eff(t30, 256) >= 0.85) {
                tile = 30;
            } else if (t30 > 0) {
                while (t30 * split < 200 && split < 4 && nk_all / (split * 2) >= 24 && g.ws && g.N % 4 == 0 &&
                       (size_t)(split * 2) * g.M * g.N * 4 <= (size_t)d->workspace_bytes)
                    split *= 2;
                if (t30 * split >= 200 && fd_round_eff(t30 * split, 256) >= 0.85) tile = 30;
                // two K slices of a short K loop lose to whole 128x320 tiles (the slab round trip + finish launch are fixed costs)
                if (tile == 30 && split == 2 && nk_all < 150 && b2ok && t32 >= 200 && fd_round_eff(t32, 256) >= 0.85) {
                    tile = 32;
                    split = 1;
                }
            }
            if (!tile && b2ok && t32 >= 200 && fd_round_eff(t32, 256) >= 0.85) tile = 32, split = 1;
            // few rows, long K (the 8x8 level: 1024 x 1280 x 11520 / 23040): 128x320 tiles x up to 8 K slices -- 44.4 vs 46.0 / 66.8 vs 71.4 us
            // against 256x160 x 8 on the 2-barrier kernel (4 launches each, profiles/r05_gemm_sweep_vae.txt header run)
            if (!tile && t32 > 0) {
                int s8 = 1;
                while (t32 * s8 < 200 && s8 < 8 && nk_all / (s8 * 2) >= 8 && g.ws && g.N % 4 == 0 &&
                       (size_t)(s8 * 2) * g.M * g.N * 4 <= (size_t)d->workspace_bytes)
                    s8 *= 2;
                if (s8 > 1 && t32 * s8 >= 200 && fd_round_eff(t32 * s8, 256) >= 0.85) tile = 32, split = s8;
            }
        } else if (g.mode != MODE_CONV && (g.K + g.K2 >= 1280 || (g.K + g.K2 >= 640 && g.N >= 1280))) {
            // FF-out (+ folded proj_out, + residual) and the wide LayerNorm-folded projections (fused q|k at the 32x32 / 16x16 levels:
            // 30.6 vs 35.0 us and 29.1 vs 32.0 us, tools/ab_pp_recorded.py)
            if (t30 >= 200 && fd_round_eff(t30, 256) >= 0.85) tile = 30;
            else if (t32 >= 200 && fd_round_eff(t32, 256) >= 0.85) tile = 32;
        }
        if (split == 1 && !b2ok) tile = 0;
        if (tile) {
            g.split_k = split;
            if (fd_gemm_pp_ok(g, batch, tile)) {
                best_tile = tile;
                best_split = split;
            }
            g.split_k = 1;
        }
    }
    // ... and the VAE decoder's widths (N = 256 / 512, no 160-wide tiles): the ping-pong 256x256 tile is 3-8 % faster than the 2-barrier 256x256
    // tile on its convolutions from the 64x64 maps up (profiles/r05_gemm_sweep_vae.txt), 10-18 % on its K = 512 attention projections, 160 vs 213 us on the
    // mid-block attention's batched QK^T (8 x 4096 x 4096 x 512)
    if (g_pp && g_use_dma && g_fast_epi && g_bias_lds && !geglu && !n160 && !d->ln_stats_out && !g.ln_stats && !g.out_f32 && !d->trans_out && g.act == FD_ACT_NONE &&
        best_split == 1 && g.N % 256 == 0 && g.M % 256 == 0 && (g.ldc & 7) == 0 && (!g.res || (g.ldr & 3) == 0) &&
        (!g.bias2 || g.rows_per_batch % 256 == 0)) {
        const long long t31 = (long long)(g.M / 256) * (g.N / 256) * batch;
        const bool pick = g.mode == MODE_CONV ? (nk_all >= (g.phase ? 16 : 32) && t31 >= (g.phase ? 512 : 256)) : (g.K >= 512 && t31 >= 256);
        if (pick && fd_gemm_pp_ok(g, batch, 31)) best_tile = 31;
    }
    if (d->tile) best_tile = d->tile;
    if (d->split_k > 0) best_split = d->split_k;
    if (g.ln_stats) best_split = 1;   // the split-K finish kernel does not know the fold
    if (d->ln_stats_out) {
        // N == 320: the 256x320 tile spans the whole row and finalises (rstd, -mean rstd) itself.  Wider rows (and N == 320
        // at row counts the 288-row tile divides better): the 160-wide tiles write raw per-n-tile partial sums
        // [N / 160][M][2] for fd_ln_finalize_stats_f32.  fd_gemm_can_emit_row_stats tells the caller which it will be.
        int stile = 0;
        const int slabs = fd_stats_plan(g.M, g.N, &stile, batch);
        // (batch > 1: the batches' rows must follow each other in C so that row index = batch * M + m)
        FD_CHECK_ARG(slabs > 0 && !d->conv && !d->trans_out && !d->out_f32 &&
                         (batch == 1 || d->batch_stride_c == (int64_t)g.M * g.ldc), FD_ESHAPE,
                     "fd_gemm_f16: ln_stats_out needs N == 320 with M %% 256 == 0, or N %% 160 == 0 with M %% 128 == 0 or M %% 288 == 0 (got M=%d N=%d)", g.M, g.N);
        g.ln_stats_out = d->ln_stats_out;
        g.stats_rows = g.M * batch;
        g.ln_eps = d->ln_eps > 0.f ? d->ln_eps : 1e-5f;
        // (a caller-forced 160-wide tile of the same row count keeps the slab layout)
        if (!(slabs > 1 && stile != 23 && (best_tile == 12 || best_tile == 20 || (best_tile == 13 && g.M % 256 == 0)))) best_tile = stile;
        best_split = 1;
    }
    {
        // The many-wave tiles exist only on the LDS-DMA path, whose buffer descriptors address a
        // tensor through 32-bit byte offsets (< 2 GiB).  Larger operands go to the register-staged
        // 4-wave kernel (32-bit ELEMENT offsets: < 2^31 halfs = 4 GiB); beyond that, refuse.
        const unsigned long long a_elems =
            g.mode == MODE_CONV ? ((unsigned long long)(g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi - 1) * g.Cpix + g.Cin
                                : (unsigned long long)(g.M - 1) * g.lda + g.K;
        const unsigned long long w_elems = (unsigned long long)(g.N - 1) * g.ldw + g.K;
        const unsigned long long c_elems = (unsigned long long)(g.M - 1) * g.ldc + g.N;
        FD_CHECK_ARG(a_elems < 0x7fffffffull && w_elems < 0x7fffffffull && c_elems < 0x7fffffffull, FD_ESHAPE,
                     "fd_gemm_f16: operand of %llu elements exceeds the 2^31-element addressing limit; "
                     "split the batch", a_elems > w_elems ? a_elems : w_elems);
        // a per-batch bias is only staged by the LDS-DMA kernels (the global-memory fallbacks of the epilogues read bias[n])
        FD_CHECK_ARG(g.strideBias == 0 || (batch > 1 && g.bias && g_use_dma && g.bias_lds && 2 * a_elems < 0x7fffffffull &&
                                          2 * w_elems < 0x7fffffffull && d->split_k <= 1),
                     FD_ESHAPE, "fd_gemm_f16: batch_stride_bias needs batch > 1, a bias, the LDS-DMA path with LDS-staged biases and no split-K");
        if (g.strideBias) best_split = 1;
        if (2 * a_elems >= 0x7fffffffull || 2 * w_elems >= 0x7fffffffull) {
            // the register-staged kernel knows neither the LayerNorm fold nor the producer statistics: refuse
            // rather than hand the consumer GEMM uninitialised statistics
            FD_CHECK_ARG(!g.ln_stats && !g.ln_stats_out, FD_ESHAPE,
                         "fd_gemm_f16: ln_stats / ln_stats_out need the LDS-DMA path (operands < 2 GiB); split the batch");
            best_tile = geglu ? 1 : (g.N % 160 == 0 ? 2 : 1);   // 128x160 / 128x128, 4 waves
            if (g.N <= 64) best_tile = 3;
        }
    }
    if (g.phase) {
        // only the lean plain epilogue knows the parity row map: the chosen tile must have one and every tile must be full
        int bm = 0, bn = 0;
        switch (best_tile) {
            case 9: case 12: case 20: bm = 128; bn = 160; break;
            case 10: bm = 128; bn = 128; break;
            case 13: bm = 256; bn = 160; break;
            case 23: bm = 288; bn = 160; break;
            case 14: bm = 256; bn = 128; break;
            case 15: bm = 256; bn = 256; break;
            case 16: bm = 256; bn = 320; break;
            case 30: case 31: case 32: case 33: fd_gemm_pp_tile_shape(best_tile, &bm, &bn); break;
            default: break;
        }
        FD_CHECK_ARG(bm && g.M % bm == 0 && g.N % bn == 0 && best_split == 1, FD_ESHAPE,
                     "fd_gemm_f16: upsample2x == 2: tile %d / split %d cannot run M=%d N=%d through the lean epilogue",
                     best_tile, best_split, g.M, g.N);
    }
    if (g.res_rows) {
        // the lean epilogues wrap the residual row once per wave row block: a tile must not straddle the wrap (the finish pass of a launch split
        // over K adds the residual and wraps per row)
        const int bm = best_tile == 23 ? 288 : 256;
        FD_CHECK_ARG(best_split > 1 || g.res_rows % bm == 0, FD_ESHAPE,
                     "fd_gemm_f16: residual_rows=%d with tile %d (needs a multiple of %d rows)", g.res_rows, best_tile, bm);
    }
    if (best_split > 1)
        FD_CHECK_ARG(!geglu && batch == 1 && g.N % 4 == 0 && g.ws &&
                         (size_t)best_split * g.M * g.N * 4 <= (size_t)d->workspace_bytes,
                     FD_ESHAPE, "fd_gemm_f16: split_k=%d not possible for this problem", best_split);
    if (geglu && (best_tile == 2 || best_tile == 5 || best_tile == 7 || best_tile == 9 || best_tile == 12 || best_tile == 13 || best_tile == 16 || best_tile == 20 || best_tile == 23)) best_tile = 1;
    g.split_k = best_split;
    g.tap_fast = g.mode == MODE_CONV && (g_tap_fast == 2 || (g_tap_fast == 1 && best_tile == 16) || best_tile >= 30);   // (the ping-pong tiles are tap-fastest only)
    if (best_tile >= 30)
        FD_CHECK_ARG(fd_gemm_pp_ok(g, batch, best_tile), FD_ESHAPE,
                     "fd_gemm_f16: ping-pong tile %d cannot run M=%d N=%d K=%d (full tiles, K %% 64 == 0, conv: Wo %% 8 == 0, no fused upsample)",
                     best_tile, g.M, g.N, g.K);
    if (g.ln_stats && !(best_tile == 9 || best_tile == 10 || (best_tile >= 12 && best_tile <= 16) || best_tile == 20 || best_tile == 23 || best_tile >= 30))
        best_tile = best_tile == 4 ? 4 : -7;     // (reported by fd_gemm_plan as -7: the 128x128 generic kernel with the fold compiled in)
    if (choice) {
        if (choice[0] == -1) choice[2] = gn_parts_bm(g, best_tile, best_split, batch);   // fd_gemm_gn_parts_chunks
        choice[0] = best_tile;
        choice[1] = best_split;
        return FD_OK;
    }
    if (g.gn_part_out)
        FD_CHECK_ARG(gn_parts_bm(g, best_tile, best_split, batch) > 0 && (uintptr_t)g.gn_part_out % 8 == 0 &&
                         g.rows_per_batch / gn_parts_bm(g, best_tile, best_split, batch) == d->gn_part_chunks, FD_ESHAPE,
                     "fd_gemm_f16: gn_part_out cannot be honoured by tile %d x split %d of M=%d N=%d (rows per sample %d, %d groups); ask fd_gemm_gn_parts_chunks first",
                     best_tile, best_split, g.M, g.N, g.rows_per_batch, g.gn_groups);
    if (g.gn_out)
        FD_CHECK_ARG(best_split == 2 || best_split == 4 || best_split == 8 || best_split == 16, FD_ESHAPE,
                     "fd_gemm_f16: gn_out is honoured by split-K launches only (this one: tile %d, split_k %d); ask fd_gemm_plan first", best_tile, best_split);
    fd_prof_begin(FD_FAMILY_GEMM, st, flops, flops_exec, fd_tag(11u, g.M * batch, g.N, g.K + g.K2, best_tile * 64 + best_split, (g.mode << 8) | (g.act << 4) | (g.res ? 2 : 0) | (g.ln_stats ? 1 : 0)));
    if (g.ln_stats && !(best_tile == 9 || best_tile == 10 || (best_tile >= 12 && best_tile <= 16) || best_tile == 20 || best_tile == 23 || best_tile >= 30)) {
        // small problems: the generic epilogue with the fold compiled in (64x64 for few rows)
        rc = best_tile == 4 ? launch_mode<64, 64, false, false, 2, 2, 2, 7>(g, batch, st)
                            : launch_mode<128, 128, false, false, 2, 2, 2, 7>(g, batch, st);
        fd_prof_end(FD_FAMILY_GEMM, st);
        return rc;
    }
    switch (best_tile) {
        case 2: rc = launch<128, 160, false>(g, batch, st); break;
        case 3: rc = launch<128, 64, false>(g, batch, st); break;
        case 4: rc = launch<64, 64, false>(g, batch, st); break;
        case 5: rc = launch<256, 160, false, 4>(g, batch, st); break;
        case 6: rc = launch<256, 128, false, 4>(g, batch, st); break;
        case 7: rc = launch<256, 160, false, 4, 3>(g, batch, st); break;
        case 9: rc = launch_epi<128, 160, 4, 2, 2, 38>(g, batch, st); break;    // 8 waves, 32x80 wave tiles
        case 10: rc = launch_epi<128, 128, 4, 2, 2, 110>(g, batch, st); break;
        case 11: rc = launch<128, 64, false, 4>(g, batch, st); break;
        case 12: rc = launch_epi<128, 160, 8, 2, 2, 38 + 256>(g, batch, st); break;   // 16 waves, 16x80 wave tiles
        case 13: rc = launch_epi<256, 160, 8, 2, 2, 38 + 256>(g, batch, st); break;   // 16 waves, 32x80 wave tiles
        case 23: rc = launch_epi<288, 160, 6, 2, 2, 38 + 256>(g, batch, st); break;   // 12 waves, 48x80 wave tiles (rows = 9 x 2^k: 768^2 images)
        case 14: rc = launch_epi<256, 128, 8, 2, 2, 110>(g, batch, st); break;   // 16 waves, 32x64 wave tiles
        case 8: rc = launch<256, 128, false, 4, 3>(g, batch, st); break;
        case 15: rc = launch_epi<256, 256, 4, 2, 4, 110>(g, batch, st); break;  // 16 waves, 64x64 wave tiles
        case 16: rc = launch_epi<256, 320, 4, 2, 4, 294>(g, batch, st); break;  // 16 waves, 64x80 wave tiles
        case 20: rc = launch_epi<128, 160, 4, 3, 2, 38 + 256>(g, batch, st); break;   // tile 9 with 3 LDS stages
        case 30: case 31: case 32: case 33: rc = fd_gemm_pp_launch(g, batch, st, best_tile); break;   // gemm_pp.hip
        // (a 3-stage form of tile 13 -- launch_epi<256, 160, 8, 3, 2, 6>: 3 x 53,248 B + bias tiles = 162,304 B, it does fit the 160 KiB --
        //  measured identical to the 2-stage tile on every deep-level convolution, profiles/r04_session_ab.txt sec. 5: not instantiated)
        default: rc = launch<128, 128, false>(g, batch, st); break;
    }
#ifdef FD_SPLITK_NO_FINISH   // timing-only variant (tools/seam_probe.py): the chain without the finish launch
    if (false) {
#else
    if (rc == FD_OK && g.split_k > 1 && g.sk_sync && best_tile >= 30) {
        // (experimental in-launch reduction: the partial launch finished its own tiles)
    } else if (rc == FD_OK && g.split_k > 1 && g.gn_out) {
        switch (g.split_k) {
            case 2: rc = launch_finish_gn<2>(g, st); break;
            case 4: rc = launch_finish_gn<4>(g, st); break;
            case 8: rc = launch_finish_gn<8>(g, st); break;
            default: rc = launch_finish_gn<16>(g, st); break;
        }
    } else if (rc == FD_OK && g.split_k > 1) {
#endif
        const size_t total = (size_t)g.M * (g.N / 4);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_splitk_finish, dim3(blocks), dim3(256), 0, st, g);
        FD_CHECK_LAUNCH("k_splitk_finish");
    }
    fd_prof_end(FD_FAMILY_GEMM, st);
    return rc;
}
