// GemmArgs and the fused epilogues shared by the GEMM kernel families (gemm.hip: the 2-barrier LDS-DMA loops,
// gemm_pp.hip: the deep-pipelined ping-pong loop).  Device code only; included once per translation unit.
#pragma once
#include <stdlib.h>

#include "common.h"

#define BK 64
// W fragments in flight in the pinned inner loop of the 64x80 wave tiles (0 = compiler-scheduled).
// Measured on the level-0 conv (16x64x64x320->320): 1 -> 117 us, 0 -> 134 us, 2 -> 140 us; pinning
// the 32x80 wave tiles the same way costs them 8 %.
#ifndef FD_T16_MODE
#define FD_T16_MODE 1
#endif

enum { MODE_LINEAR = 0, MODE_CONV = 1 };

struct GemmArgs {
    const half_t* A;
    const half_t* W;
    void* C;
    const float* bias;
    const float* bias2;
    const half_t* res;
    long long strideA, strideW, strideC, strideRes;
    int M, N, K, lda, ldw, ldc, ldr, ldb2;
    int mode;
    int Hi, Wi, Cin, Ho, Wo, KW, stride, pad_t, pad_l, up;
    int Cpix;   // conv: pixel stride of the NHWC input in halfs (>= Cin: the input may be a column slice of a wider matrix)
    int rows_per_batch;
    int act, out_f32, trans_out;
    long long strideT;  // per-sample stride of the transposed output
    int ldt;
    float alpha;
    int tiles_m, tiles_n;
    int tap_fast;  // conv: tap-fastest K order (see k_gemm_f16_dma)
    int bias_lds;  // stage the tile's bias through LDS (FD_GEMM_BIAS_LDS=0 reads it from global memory)
    float* ln_stats_out;    // row statistics (rstd, -mean*rstd) of the OUTPUT rows, written by full-row tiles (N == BN)
    float ln_eps;
    const float* ln_stats;  // LayerNorm fold: per row of A (rstd, -mean*rstd); bias2 = column sums of W, bias = folded bias
    int split_k;   // > 1: a workgroup owns a K slice (blockIdx.y, or see sk_flat) and stores fp32 partials to `ws`
    int phase;          // phase-decomposed nearest-2x upsample + 3x3 conv: blockIdx.z = output-pixel parity (py, px), see fd_gemm_desc.upsample2x == 2
    const half_t* A2;   // conv + appended 1x1 phase: after the conv's K-tiles the loop runs K2 more columns over the rows of
    unsigned a2_bytes;
    int lda2, K2;       // A2 [M][lda2] (the ResBlock's shortcut conv folded into conv2's accumulation); W is [N][K + K2]
    long long strideBias;   // batch > 1: floats between the biases of consecutive batches (fd_gemm_desc.batch_stride_bias)
    int stats_rows;         // ln_stats_out: rows of the whole launch (batch x M): slab stride of the partial sums
    int sk_flat;   // split-K on a flat 1-D grid: slice = blockIdx.x % split_k, tile = blockIdx.x / split_k (see k_gemm_f16_dma)
    float* ws;     // [split_k][M][N] fp32
    // GroupNorm(+SiLU) of the output fused into the split-K finish (k_splitk_finish_gn, fd_gemm_desc.gn_out)
    half_t* gn_out;
    const float* gn_gamma;
    const float* gn_beta;
    int gn_groups, gn_silu, gn_gb, gn_skip_c;   // gn_gb: groups per workgroup (gn_slab_pick)
    float gn_eps;
    // GroupNorm partial sums of the OUTPUT from the lean epilogue of a row-spanning tile (fd_gemm_desc.gn_part_out):
    // [sample][rows_per_batch / BM][gn_groups][2] = (sum, sum of squares) of the fp16-rounded values, the layout k_gn_stats writes
    float* gn_part_out;
    // transposed tail (fd_gemm_desc.trans_n0): tiles with n0 >= tr_n0 run the transposed-store epilogue into C2
    int tr_n0;
    half_t* C2;
    // experimental in-launch split-K reduction (fd_gemm_desc.sk_sync): [tiles] arrival counters, then [tiles] departure counters
    unsigned* sk_sync;
    // LayerNorm fold fed with the producer's PARTIAL sums (fd_gemm_desc.ln_stats_parts): ln_stats = [ln_parts][ln_rows][2] raw (sum, sum of squares);
    // each tile finalises its rows into LDS (ln_tile_stats_to_lds) and the epilogues read (rstd, -mean rstd) from there
    int ln_parts, ln_rows;
    float ln_inv_n, ln_eps_in;
    int res_rows;   // > 0: the residual row of output row m is m % res_rows (fd_gemm_desc.residual_rows; a multiple of the tile's rows)
};

// Exact-form GELU  x * Phi(x),  Phi(x) = 0.5 * (1 + erf(x / sqrt 2)),  with erf from Abramowitz &
// Stegun 7.1.26 (|abs error| <= 1.5e-7, far below the fp16 output's 2^-11 relative rounding).
// libm's erff (~40 VALU ops) made the fused GEGLU epilogue VALU-bound on the short-K feed-forward
// GEMMs; here every constant factor (1/sqrt 2, the 0.5 of Phi, log2 e) is folded into the
// coefficients and the sign is handled without a select:
//   q = 0.5 * (1 - erf(|x| / sqrt 2)) = poly(t) * 2^(-x^2 * log2(e) / 2),  t = 1 / (1 + p |x| / sqrt 2)
//   gelu(x) = max(x, 0) - |x| * q            (x >= 0: x - x q = x (1 - q);  x < 0: x q)
// = 1 v_rcp + 1 v_exp + 11 plain VALU ops.
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.0f));      // 0.3275911 / sqrt 2
    float p = fmaf(0.5307027145f, t, -0.7265760135f);                         // a5/2, a4/2
    p = fmaf(p, t, 0.7107068705f);                                            // a3/2
    p = fmaf(p, t, -0.142248368f);                                            // a2/2
    p = fmaf(p, t, 0.127414796f);                                             // a1/2
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * (x * x));  // exp(-x^2 / 2)
    const float q = p * t * e;
    return fmaf(-ax, q, fmaxf(x, 0.0f));
}

// Two GELUs at a time on the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of fp32 per issue slot).
// The fused GEGLU epilogue of the short-K feed-forward GEMMs spends MORE issue time in the VALU than in the matrix
// pipe (PMC, level-0 65536 x 2560 x 320: 32.5 M VALU instructions against 104.9 M MFMA-busy cycles), and a
// transcendental costs ~2.7 plain slots, so this form uses ONE of them per element instead of two: erf from Abramowitz &
// Stegun 7.1.28, erf(z) = 1 - (1 + a1 z + ... + a6 z^6)^-16 (|error| <= 3e-7), z = |x| / sqrt 2 folded into the
// coefficients:  q = 0.5 (1 - erf) = 0.5 / d^16,  gelu(x) = max(x, 0) - |x| q   (as gelu_fast).
// 7.1e-7 max abs error in fp32 against 3.3e-7 for gelu_fast (both far below the fp16 output's rounding); per pair
// 8 packed FMAs + 5 packed multiplies + 2 v_rcp + 2 v_and + 2 v_max = 10.7 issue slots per element against 16.4.
__device__ __forceinline__ floatx2 gelu_fast2(floatx2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const floatx2 ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
    floatx2 d = __builtin_elementwise_fma(floatx2{5.3829750e-06f, 5.3829750e-06f}, ax, floatx2{4.8890635643e-05f, 4.8890635643e-05f});
    d = __builtin_elementwise_fma(d, ax, floatx2{3.8003575000e-05f, 3.8003575000e-05f});
    d = __builtin_elementwise_fma(d, ax, floatx2{3.2776263241e-03f, 3.2776263241e-03f});
    d = __builtin_elementwise_fma(d, ax, floatx2{2.1141006150e-02f, 2.1141006150e-02f});
    d = __builtin_elementwise_fma(d, ax, floatx2{4.9867346967e-02f, 4.9867346967e-02f});
    d = __builtin_elementwise_fma(d, ax, floatx2{1.0f, 1.0f});
    d = d * d;
    d = d * d;
    d = d * d;
    d = d * d;   // d^16 (overflows to +inf beyond |x| ~ 40: the reciprocal is then 0, gelu = max(x, 0))
    const floatx2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const floatx2 m = {__builtin_fmaxf(x[0], 0.0f), __builtin_fmaxf(x[1], 0.0f)};
    return __builtin_elementwise_fma(ax * floatx2{-0.5f, -0.5f}, r, m);
#else
    return x;
#endif
}

__device__ __forceinline__ float act_apply(float x, int act) {
    switch (act) {
        case FD_ACT_SILU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));
        case FD_ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
        case FD_ACT_GELU: return gelu_fast(x);
        default: return x;
    }
}

typedef const __attribute__((address_space(3))) float* lds_cfloat;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // native vector: stays in VGPRs

// v_permlane16_swap: exchanges a's odd 16-lane rows with b's even rows (lane l <-> l ^ 16)
__device__ __forceinline__ void swap16(unsigned& a, unsigned& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
#endif
}

// A/B switch (compile time): FD_GEMM_NT_STORE=1 stores the lean epilogue's output rows with the non-temporal hint
// (a streamed output should not evict the A / W panels its neighbours still re-read from L2), 2: GEGLU rows only.
#ifndef FD_GEMM_NT_STORE
#define FD_GEMM_NT_STORE 0
#endif
template <bool NT>
__device__ __forceinline__ void epi_store16(half_t* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}

typedef const __attribute__((address_space(3))) floatx2* lds_cf2;

// LayerNorm statistics of a tile's rows from the producer's partial slabs (fd_gemm_desc.ln_stats_parts), finalised ONCE per tile into LDS:
// thread t < BM sums the k partial (sum, sum of squares) of row m0 + t in slab order and writes (rstd, -mean rstd) -- the arithmetic of
// k_ln_finalize (norm.hip), bit for bit, without the launch.  Called after the tile's first DMAs are issued and before the first barrier of the
// main loop: the loads ride on the wait the loop makes anyway.  K = compile-time slab count (all loads in flight together).
template <int K>
__device__ __forceinline__ void ln_tile_stats_body(const GemmArgs& g, const float* __restrict__ p, float* __restrict__ dst) {
    floatx2 v[K];
#pragma unroll
    for (int t = 0; t < K; ++t) v[t] = *reinterpret_cast<const floatx2*>(p + 2 * (size_t)t * g.ln_rows);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) {   // fixed order
        s1 += v[t][0];
        s2 += v[t][1];
    }
    const float mean = s1 * g.ln_inv_n;
    const float var = fmaxf(fmaf(-mean, mean, s2 * g.ln_inv_n), 0.f);
    const float rstd = rsqrtf(var + g.ln_eps_in);
    *reinterpret_cast<floatx2*>(dst) = floatx2{rstd, -mean * rstd};
}

template <int BM>
__device__ __forceinline__ void ln_tile_stats_to_lds(const GemmArgs& g, int m0, int tid, float* stats_s) {
    if (tid < BM) {
        const int m = min(m0 + tid, g.M - 1);
        const float* p = g.ln_stats + 2 * (size_t)m;
        float* dst = stats_s + 2 * tid;
        switch (g.ln_parts) {
            case 2: ln_tile_stats_body<2>(g, p, dst); break;
            case 4: ln_tile_stats_body<4>(g, p, dst); break;
            default: ln_tile_stats_body<8>(g, p, dst); break;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS writes are done before whatever barrier comes next
}

// Lean epilogue for the common case: a tile that lies completely inside [M][N], fp16 output with
// 16-byte-aligned rows, bias / per-sample bias already staged in LDS by the main loop, activation
// and residual fixed at COMPILE time.  The generic epilogue below handles every flag at run time
// inside the per-fragment loops; hipcc turns that into ~1200 executed instructions per wave and
// tile (exec-mask branches and bounds tests per fragment, integer divisions for the sample index,
// SGPR spills through v_readlane, a `switch (act)` per element), i.e. 6-8 us per tile on EVERY
// launch -- more than the whole K loop of the K <= 640 transformer projections.  Here a fragment
// costs one v_fma per element (alpha and the summed biases), the activation, the optional
// residual add, a packed convert and half a 16-byte store.  It is a COMPILE-time choice of the
// kernel (template parameter EPI of k_gemm_f16_dma / _dmap: 0 generic, 1 plain, 2 + residual,
// 3 GEGLU): with both epilogues inlined in one kernel the 16-wave persistent kernels (128-VGPR
// cap) spill ~500 bytes per lane to scratch and run 2x slower.  The host picks EPI != 0 only when
// every tile of the launch is full and the biases are LDS-staged (launch_epi).
template <int MI, int NI, int ACT, bool RES, bool B2, bool LNF = false, bool STATS = false, int WN_ = 1, int GNP_WM = 0>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmArgs& g, floatx4 (&acc)[MI][NI], int row0,
                                                   int col0, int coll, int fq, int z,
                                                   lds_cfloat bias_tile, lds_cfloat bias2_tile,
                                                   float* xch = nullptr, int trow0 = 0, int wn = 0, int m0 = 0,
                                                   lds_cfloat stats_tile = nullptr) {
    // LNF: the rows' (rstd, -mean rstd) from global memory, or -- stats_tile != nullptr (a workgroup-uniform choice) -- from the tile's LDS copy
    // that ln_tile_stats_to_lds finalised from the producer's partial sums; trow0 = this lane's row inside the tile
#define LNF_STAT(DST, I)                                                                                   \
    {                                                                                                      \
        if (stats_tile) DST = *reinterpret_cast<lds_cf2>(stats_tile + 2 * (trow0 + (I) * 16));             \
        else DST = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)(row0 + (I) * 16));          \
    }
    typedef const __attribute__((address_space(3))) floatx4* lds_cf4;
    const int pcol = (fq & 1) * 16 + (fq >> 1) * 8;   // column of this lane's paired 16-byte store
    if constexpr (ACT == FD_ACT_GEGLU) {
        // interleaved weight rows: even fragment = value, odd fragment = gate; output width N/2
        constexpr int NP = NI / 2;
        floatx4 bv[NP], bg[NP];
        floatx4 cv[LNF ? NP : 1], cg[LNF ? NP : 1];   // LayerNorm fold: column sums of the folded weights
        if constexpr (LNF) {
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                cv[jp] = *reinterpret_cast<lds_cf4>(bias2_tile + coll + jp * 32 + fq * 4);
                cg[jp] = *reinterpret_cast<lds_cf4>(bias2_tile + coll + jp * 32 + 16 + fq * 4);
            }
        }
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) {
            // the bias tile is ALWAYS valid here (zeros when the GEMM has no bias: the kernels stage
            // it through a zero-length buffer descriptor): a `if (bias)` around these reads makes
            // hipcc carry the whole array through phi copies and spill it
            bv[jp] = *reinterpret_cast<lds_cf4>(bias_tile + coll + jp * 32 + fq * 4);
            bg[jp] = *reinterpret_cast<lds_cf4>(bias_tile + coll + jp * 32 + 16 + fq * 4);
        }
        half_t* Cb = reinterpret_cast<half_t*>(g.C) + (size_t)z * g.strideC + (size_t)row0 * g.ldc + (col0 >> 1);
        // LNF: (rstd, -mean * rstd) of this lane's row, loaded ONE ROW BLOCK AHEAD: a load issued inside block i is
        // waited for with vmcnt(0), i.e. together with block i-1's stores -- a load and a store round trip per block
        floatx2 st_next = {g.alpha, 0.f};
        if constexpr (LNF) LNF_STAT(st_next, 0)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            half_t* Crow = Cb + (size_t)i * 16 * g.ldc;
            half4 og[NP];
            const floatx2 st = st_next;
            if constexpr (LNF) {
                if (i + 1 < MI) LNF_STAT(st_next, i + 1)
            }
            // two columns per issue slot (packed fp32): st = (rstd, -mean rstd) or (alpha, 0)
            const floatx2 s0 = {st[0], st[0]}, s1 = {st[1], st[1]};
#pragma unroll
            for (int jp = 0; jp < NP; ++jp)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const floatx2 av = {acc[i][2 * jp][r], acc[i][2 * jp][r + 1]};
                    const floatx2 ag = {acc[i][2 * jp + 1][r], acc[i][2 * jp + 1][r + 1]};
                    floatx2 tv = {bv[jp][r], bv[jp][r + 1]}, tg = {bg[jp][r], bg[jp][r + 1]};
                    if constexpr (LNF) {   // LN(x) W^T = rstd (x W'^T) - rstd mean colsum(W') + (b + beta W^T)
                        tv = __builtin_elementwise_fma(s1, floatx2{cv[jp][r], cv[jp][r + 1]}, tv);
                        tg = __builtin_elementwise_fma(s1, floatx2{cg[jp][r], cg[jp][r + 1]}, tg);
                    }
                    const floatx2 v = __builtin_elementwise_fma(av, s0, tv);
                    const floatx2 o = v * gelu_fast2(__builtin_elementwise_fma(ag, s0, tg));
                    og[jp][r] = (half_t)o[0];
                    og[jp][r + 1] = (half_t)o[1];
                }
#pragma unroll
            for (int jp = 0; jp < NP; jp += 2) {
                if (jp + 1 < NP) {
                    const u32x2 x = __builtin_bit_cast(u32x2, og[jp]);
                    const u32x2 y = __builtin_bit_cast(u32x2, og[jp + 1 < NP ? jp + 1 : jp]);
                    unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
                    swap16(x0, y0);
                    swap16(x1, y1);
                    epi_store16<FD_GEMM_NT_STORE != 0>(Crow + jp * 16 + pcol, u32x4{x0, x1, y0, y1});
                } else {
                    *reinterpret_cast<half4*>(Crow + jp * 16 + fq * 4) = og[jp];
                }
            }
            // one row block at a time: left alone the scheduler interleaves all MI blocks for ILP,
            // runs out of the 128 VGPRs of a 16-wave workgroup and spills hundreds of dwords
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    } else {
        floatx4 bb[NI];
        floatx4 cs[LNF ? NI : 1];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            bb[j] = *reinterpret_cast<lds_cf4>(bias_tile + coll + j * 16 + fq * 4);
            if constexpr (LNF) cs[j] = *reinterpret_cast<lds_cf4>(bias2_tile + coll + j * 16 + fq * 4);
            else if constexpr (B2) bb[j] += *reinterpret_cast<lds_cf4>(bias2_tile + coll + j * 16 + fq * 4);
        }
        half_t* Cb = reinterpret_cast<half_t*>(g.C) + (size_t)z * g.strideC + (size_t)row0 * g.ldc + col0;
        if (g.phase) Cb = reinterpret_cast<half_t*>(g.C) + col0;   // rows are mapped per 16-row block below
        // (res_rows: a multiple of the tile's rows, so one wrap of the wave's first row serves its whole row block)
        const int rrow0 = (RES && g.res_rows) ? row0 % g.res_rows : row0;
        const half_t* Rb = RES ? g.res + (size_t)z * g.strideRes + (size_t)rrow0 * g.ldr + col0 + fq * 4 : nullptr;
        floatx2 st_next = {g.alpha, 0.f};   // LNF: one row block ahead (see the GEGLU branch)
        if constexpr (LNF) LNF_STAT(st_next, 0)
        // GNP_WM > 0 (the tile spans the row, N == BN, and lies inside one sample): GroupNorm partial sums of the tile's output per
        // COLUMN PAIR, from the fp16-rounded values, with v_dot2_f32_f16 (a pair never straddles two groups: N / groups is even)
        float gsum[GNP_WM ? NI : 1][2], gsq[GNP_WM ? NI : 1][2];
#pragma unroll
        for (int j = 0; j < (GNP_WM ? NI : 1); ++j) gsum[j][0] = gsum[j][1] = gsq[j][0] = gsq[j][1] = 0.f;
        // (the residual rows one block ahead as well: measured neutral and 2-3 spilled VGPRs on the 256x320 tile -- not kept)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            half_t* Crow = Cb + (size_t)i * 16 * g.ldc;
            if (g.phase) {
                // GEMM row m = (b, y, x) of the LOW-resolution grid; this launch slice (z = py*2 + px) owns the
                // output pixels (2y + py, 2x + px) of the 2x-upsampled map
                const int m = row0 + i * 16;
                const int hw = g.Ho * g.Wo, b = m / hw, rem = m - b * hw, y = rem / g.Wo, x = rem - y * g.Wo;
                Crow = Cb + ((size_t)(b * 2 * g.Ho + 2 * y + (z >> 1)) * (2 * g.Wo) + 2 * x + (z & 1)) * g.ldc;
            }
            half4 rr[RES ? NI : 1];
            if constexpr (RES) {
                const half_t* Rrow = Rb + (size_t)i * 16 * g.ldr;
#pragma unroll
                for (int j = 0; j < NI; ++j) rr[j] = *reinterpret_cast<const half4*>(Rrow + j * 16);
            }
            half4 oh[NI];
            const floatx2 st = st_next;
            if constexpr (LNF) {
                if (i + 1 < MI) LNF_STAT(st_next, i + 1)
            }
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v;
                    if constexpr (LNF) v = fmaf(acc[i][j][r], st[0], fmaf(st[1], cs[j][r], bb[j][r]));
                    else v = fmaf(acc[i][j][r], g.alpha, bb[j][r]);
                    if constexpr (ACT == FD_ACT_SILU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
                    if constexpr (ACT == FD_ACT_QUICK_GELU) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
                    if constexpr (ACT == FD_ACT_GELU) v = gelu_fast(v);
                    if constexpr (RES) v += (float)rr[j][r];
                    oh[j][r] = (half_t)v;
                }
            if constexpr (STATS) {
                // LayerNorm statistics of the rows this kernel WRITES (the tile spans the whole row:
                // N == BN), from the fp16-rounded values the consumer will read: lane partial over its
                // 4 * NI columns -> the 4 lanes of a row (xor 16, 32) -> one slot per wave in LDS
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float f = (float)oh[j][r];
                        s1 += f;
                        s2 = fmaf(f, f, s2);
                    }
                s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
                s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                if (fq == 0) *reinterpret_cast<floatx2*>(xch + ((trow0 + i * 16) * WN_ + wn) * 2) = floatx2{s1, s2};
            }
            if constexpr (GNP_WM > 0) {
                const half2v one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    // (the two halves spelled out element by element: through bit_cast(u32x2, oh[j])[p] hipcc treated both halves as
                    // one value and emitted half of the dot products)
                    const half2v h01 = {oh[j][0], oh[j][1]}, h23 = {oh[j][2], oh[j][3]};
                    gsum[j][0] = __builtin_amdgcn_fdot2(h01, one2, gsum[j][0], false);
                    gsq[j][0] = __builtin_amdgcn_fdot2(h01, h01, gsq[j][0], false);
                    gsum[j][1] = __builtin_amdgcn_fdot2(h23, one2, gsum[j][1], false);
                    gsq[j][1] = __builtin_amdgcn_fdot2(h23, h23, gsq[j][1], false);
                }
            }
#pragma unroll
            for (int j = 0; j < NI; j += 2) {
                if (j + 1 < NI) {
                    const u32x2 x = __builtin_bit_cast(u32x2, oh[j]);
                    const u32x2 y = __builtin_bit_cast(u32x2, oh[j + 1 < NI ? j + 1 : j]);
                    unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
                    swap16(x0, y0);
                    swap16(x1, y1);
                    epi_store16<FD_GEMM_NT_STORE == 1>(Crow + j * 16 + pcol, u32x4{x0, x1, y0, y1});
                } else {
                    *reinterpret_cast<half4*>(Crow + j * 16 + fq * 4) = oh[j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // see the GEGLU branch
        }
        if constexpr (GNP_WM > 0) {
            // (1) the 16 rows of a fragment column live in the 16 lanes of a lane group: butterfly over lane bits 0..3;
            // (2) one slot per (wave row, column pair) in LDS (the K loop's stages are dead: the kernels synchronise before an
            //     epilogue that writes `xch`); (3) thread (group, stat) sums its wave rows and its N / groups / 2 pairs in a fixed
            //     order and writes the tile's partial: [sample][chunk = tile row inside the sample][group][2]
            constexpr int PAIRS = WN_ * NI * 8;   // column pairs of the tile (BN / 2)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        gsum[j][p] += __shfl_xor(gsum[j][p], o, 64);
                        gsq[j][p] += __shfl_xor(gsq[j][p], o, 64);
                    }
            const int wm_ = trow0 / (MI * 16);
            if ((threadIdx.x & 15) == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        *reinterpret_cast<floatx2*>(xch + (wm_ * PAIRS + ((coll + j * 16 + fq * 4) >> 1) + p) * 2) = floatx2{gsum[j][p], gsq[j][p]};
            }
            __syncthreads();
            const int G = g.gn_groups, ppg = (g.N / G) >> 1, t = threadIdx.x;
            if (t < 2 * G) {
                const int gi = t >> 1, stt = t & 1;
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < GNP_WM; ++w)
                    for (int p = 0; p < ppg; ++p) a += xch[(w * PAIRS + gi * ppg + p) * 2 + stt];
                constexpr int BM_ = GNP_WM * MI * 16;
                const int b = m0 / g.rows_per_batch, chunk = (m0 - b * g.rows_per_batch) / BM_, nchunk = g.rows_per_batch / BM_;
                g.gn_part_out[(((size_t)b * nchunk + chunk) * G + gi) * 2 + stt] = a;
            }
        }
        if constexpr (STATS) {
            // combine the WN_ wave slots of each row and finalise: one lane per row (waves with wn == 0)
            __syncthreads();
            if (wn == 0) {
                const float inv_n = 1.0f / (float)g.N;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    if (fq != (i & 3)) continue;   // spread the MI row blocks over the 4 lane groups
                    const int tr = trow0 + i * 16;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int w = 0; w < WN_; ++w) {
                        const floatx2 p = *reinterpret_cast<const floatx2*>(xch + (tr * WN_ + w) * 2);
                        s1 += p[0];
                        s2 += p[1];
                    }
                    if constexpr (WN_ == 2) {   // the 160-wide tiles (the 256x320 tile, WN_ == 4, spans its row)
                        // the row spans several n-tiles: this tile's raw (sum, sum of squares) goes to slab `tile_n` of
                        // ln_stats_out [N / BN][M][2]; fd_ln_finalize_stats_f32 combines the slabs in a fixed order
                        *reinterpret_cast<floatx2*>(g.ln_stats_out + 2 * ((size_t)(col0 / (NI * 16 * WN_)) * g.stats_rows + (size_t)z * g.M + m0 + tr)) = floatx2{s1, s2};
                    } else {
                        const float mean = s1 * inv_n;
                        const float var = fmaxf(fmaf(-mean, mean, s2 * inv_n), 0.f);
                        const float rstd = rsqrtf(var + g.ln_eps);
                        *reinterpret_cast<floatx2*>(g.ln_stats_out + 2 * ((size_t)z * g.M + m0 + tr)) = floatx2{rstd, -mean * rstd};
                    }
                }
            }
        }
    }
}

#undef LNF_STAT

// Fused epilogue shared by the register-staged and the LDS-DMA main loops.
template <int BM, int BN, bool TRANS, int WM = 2, int WN = 2, bool LN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g,
                                              floatx4 (&acc)[BM / WM / 16][BN / WN / 16], int m0, int n0,
                                              int wm, int wn, int fr, int fq, int z,
                                              lds_cfloat bias_tile = nullptr, lds_cfloat bias2_tile = nullptr, int kslice = -1,
                                              lds_cfloat stats_tile = nullptr) {
    // bias_tile: this tile's bias[n0 .. n0+BN) staged in LDS by the main loop's first DMA group
    // (zeros past N).  A bias read from global memory here is a dependent L2 round trip that
    // every wave of the workgroup sits out between the last MFMA and the first store
    // (65536x320x320: 31.9 us with it, 26.4 us without).  The pointers are LDS-typed on purpose:
    // a `cond ? lds : global` pointer turns the read into a FLAT load that the compiler brackets
    // with s_waitcnt vmcnt(0) -- a full drain of the in-flight LDS-DMA and stores per fragment.
    // bias2_tile: the same for the per-sample bias when the whole tile lies in one sample.
    // Loads that must come from global memory (residual, per-sample bias of multi-sample tiles)
    // are issued as one batch per 16-row block: hipcc drains the VM counter before the first use
    // of any VGPR-destination load while an LDS-DMA is in flight, so one wait serves them all.
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 16, NI = WTN / 16;
    // ---- split-K: raw fp32 partial tile, reduced + finished by k_splitk_finish ------------
    if (g.split_k > 1) {
        float* __restrict__ P = g.ws + (size_t)(kslice >= 0 ? kslice : (int)blockIdx.y) * g.M * g.N;
#ifdef FD_SPLITK_NO_STORE   // timing-only variant (tools/seam_probe.py): the partial pass without its slab stores
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(acc[i][j]));
        if (g.M < 0) P[0] = 0.f;
        return;
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
            if (m >= g.M) continue;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nb0 = n0 + wn * WTN + j * 16 + fq * 4;
                if (nb0 >= g.N) continue;
                if (nb0 + 3 < g.N) {
                    *reinterpret_cast<float4*>(P + (size_t)m * g.N + nb0) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
                    for (int r = 0; r < 4 && nb0 + r < g.N; ++r) P[(size_t)m * g.N + nb0 + r] = acc[i][j][r];
                }
            }
        }
        return;
    }
    // ---- epilogue -----------------------------------------------------------------------
    if (TRANS) {
        // out[b][n][m_local]: lane holds rows m = base + fq*4 + r for column n = base + fr, i.e.
        // 4 consecutive elements of output row n.  Row fragments i, i+1 are re-paired across
        // lanes l <-> l^16 (v_permlane16_swap) into 8 consecutive elements = one 16-byte store.
        half_t* __restrict__ T = reinterpret_cast<half_t*>(g.C);
        const bool vec_ok = (g.rows_per_batch & 7) == 0 && (g.ldt & 7) == 0 && (g.strideT & 7) == 0 &&
                            (g.strideC & 7) == 0;
        const int prow = (fq & 1) * 16 + (fq >> 1) * 8;
        // LayerNorm fold: the lane's 4 values of a fragment are 4 consecutive ROWS m, each with its own
        // (rstd, -mean * rstd); loaded once per row block, not per column fragment
        float ln_rs[LN ? MI : 1][4], ln_mr[LN ? MI : 1][4];
        if constexpr (LN) {
            if (g.ln_stats) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int mb = min(m0 + wm * WTM + i * 16 + fq * 4, g.M - 4 > 0 ? g.M - 4 : 0);
                    typedef const __attribute__((address_space(3))) floatx4* lds_cf4s;
                    // (stats_tile: the tile's rows finalised into LDS from the producer's partial sums, ln_tile_stats_to_lds)
                    floatx4 s01, s23;
                    if (stats_tile) {
                        s01 = *reinterpret_cast<lds_cf4s>(stats_tile + 2 * (mb - m0));
                        s23 = *reinterpret_cast<lds_cf4s>(stats_tile + 2 * (mb - m0) + 4);
                    } else {
                        s01 = *reinterpret_cast<const floatx4*>(g.ln_stats + 2 * (size_t)mb);
                        s23 = *reinterpret_cast<const floatx4*>(g.ln_stats + 2 * (size_t)mb + 4);
                    }
                    ln_rs[i][0] = s01[0]; ln_mr[i][0] = s01[1]; ln_rs[i][1] = s01[2]; ln_mr[i][1] = s01[3];
                    ln_rs[i][2] = s23[0]; ln_mr[i][2] = s23[1]; ln_rs[i][3] = s23[2]; ln_mr[i][3] = s23[3];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + wn * WTN + j * 16 + fr;
            const bool n_ok = n < g.N;
            float bn = 0.f;
            if (n_ok && g.bias) {
                if (bias_tile) bn = bias_tile[n - n0];
                else bn = g.bias[n];
            }
            half4 oh[MI];
            if (LN && g.ln_stats) {
                // LayerNorm fold on the transposed layout (row statistics hoisted out of the n loop)
                // column sum of the folded weights: from the LDS tile when the main loop staged it -- a global load
                // here is waited for with vmcnt(0), i.e. together with the next tile's DMA, once per column fragment
                const float csn = (n_ok && g.bias2) ? (bias2_tile ? bias2_tile[n - n0] : g.bias2[n]) : 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        oh[i][r] = (half_t)act_apply(fmaf(acc[i][j][r], ln_rs[i][r], fmaf(ln_mr[i][r], csn, bn)), g.act);
            } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) oh[i][r] = (half_t)act_apply(acc[i][j][r] * g.alpha + bn, g.act);
            }
#pragma unroll
            for (int i = 0; i < MI; i += 2) {
                const int mblk = m0 + wm * WTM + i * 16;           // 32-row block (wave-uniform)
                const int bb = mblk / g.rows_per_batch;
                const bool full = i + 1 < MI && vec_ok && mblk + 32 <= g.M &&
                                  (mblk + 31) / g.rows_per_batch == bb;
                if (full) {
                    const u32x2 x = __builtin_bit_cast(u32x2, oh[i]);
                    const u32x2 y = __builtin_bit_cast(u32x2, oh[i + 1 < MI ? i + 1 : i]);
                    unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
                    swap16(x0, y0);
                    swap16(x1, y1);
                    const int ml = mblk - bb * g.rows_per_batch + prow;
                    if (n_ok)
                        *reinterpret_cast<u32x4*>(T + (size_t)z * g.strideC + (size_t)bb * g.strideT +
                                                  (size_t)n * g.ldt + ml) = u32x4{x0, x1, y0, y1};
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (i + u >= MI) continue;
                        const int mb = mblk + u * 16 + fq * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int m = mb + r;
                            if (m >= g.M || !n_ok) continue;
                            const int b = m / g.rows_per_batch, ml = m - b * g.rows_per_batch;
                            T[(size_t)z * g.strideC + (size_t)b * g.strideT + (size_t)n * g.ldt + ml] =
                                oh[i + u < MI ? i + u : i][r];
                        }
                    }
                }
            }
        }
        return;
    }
    const bool geglu = g.act == FD_ACT_GEGLU;
    // fp16 stores: a lane owns 4 consecutive columns (8 B) of each 16-column fragment.  Two
    // neighbouring fragments are re-paired with v_permlane16_swap so that every lane stores 8
    // consecutive columns (16 B): half the store instructions, 64 B instead of 32 B runs per row.
    //   even lane rows (fq 0,2) keep fragment X and receive their right neighbour's X part,
    //   odd lane rows (fq 1,3) keep fragment Y and receive their left neighbour's Y part.
    const int pcol = (fq & 1) * 16 + (fq >> 1) * 8;   // column of the paired 16-byte store
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m >= g.M) continue;   // lanes l and l^16 share fr, so swap partners stay together
        const int b = m / g.rows_per_batch;
        if (geglu) {
            // interleaved weight rows: even fragment = value, odd fragment = gate
            constexpr int NP = NI / 2;
            half4 og[NP > 0 ? NP : 1];
#pragma unroll
            for (int jp = 0; jp < NP; ++jp) {
                const int j = 2 * jp;
                const int nb0 = n0 + wn * WTN + j * 16 + fq * 4;
                float v[4], gt[4];
                floatx4 bb = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
                if (g.bias && nb0 + 16 < g.N) {
                    if (bias_tile) {
                        bb = *reinterpret_cast<const __attribute__((address_space(3))) floatx4*>(bias_tile + (nb0 - n0));
                        bg = *reinterpret_cast<const __attribute__((address_space(3))) floatx4*>(bias_tile + (nb0 - n0) + 16);
                    } else {
                        bb = *reinterpret_cast<const floatx4*>(g.bias + nb0);
                        bg = *reinterpret_cast<const floatx4*>(g.bias + nb0 + 16);
                    }
                }
                if (LN && g.ln_stats) {
                    floatx2 st;
                    if (stats_tile) st = *reinterpret_cast<lds_cf2>(stats_tile + 2 * (m - m0));
                    else st = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)m);
                    floatx4 cv = {0.f, 0.f, 0.f, 0.f}, cg = {0.f, 0.f, 0.f, 0.f};
                    if (g.bias2 && nb0 + 16 < g.N) {
                        cv = *reinterpret_cast<const floatx4*>(g.bias2 + nb0);
                        cg = *reinterpret_cast<const floatx4*>(g.bias2 + nb0 + 16);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaf(acc[i][j][r], st[0], fmaf(st[1], cv[r], bb[r]));
                        gt[r] = fmaf(acc[i][j + 1][r], st[0], fmaf(st[1], cg[r], bg[r]));
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // one fma, as in gemm_epilogue_fast
                        v[r] = fmaf(acc[i][j][r], g.alpha, bb[r]);
                        gt[r] = fmaf(acc[i][j + 1][r], g.alpha, bg[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) og[jp][r] = (half_t)(v[r] * gelu_fast(gt[r]));
            }
            half_t* Crow = reinterpret_cast<half_t*>(g.C) + (size_t)z * g.strideC + (size_t)m * g.ldc;
#pragma unroll
            for (int jp = 0; jp < NP; jp += 2) {
                const int cb = (n0 + wn * WTN + jp * 32) >> 1;   // first output column of block jp
                if (jp + 1 < NP && n0 + wn * WTN + (jp + 2) * 32 <= g.N && (g.ldc & 7) == 0) {
                    const u32x2 x = __builtin_bit_cast(u32x2, og[jp]);
                    const u32x2 y = __builtin_bit_cast(u32x2, og[jp + 1 < NP ? jp + 1 : jp]);
                    unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
                    swap16(x0, y0);
                    swap16(x1, y1);
                    *reinterpret_cast<u32x4*>(Crow + cb + pcol) = u32x4{x0, x1, y0, y1};
                } else {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (jp + u < NP && n0 + wn * WTN + (jp + u) * 32 + fq * 4 + 16 < g.N)
                            *reinterpret_cast<half4*>(Crow + cb + u * 16 + fq * 4) = og[jp + u < NP ? jp + u : jp];
                }
            }
            continue;
        }
        half4 oh[NI];
        // global-memory operands of this row block, issued together (see the header comment)
        constexpr bool BATCH = MI * NI < 16;   // the 64-row wave tiles have no registers to spare
        half4 rres[BATCH ? NI : 1];
        floatx4 rb2[BATCH ? NI : 1];
        const bool b2_global = g.bias2 && !bias2_tile;
        if constexpr (BATCH) {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int nb0 = n0 + wn * WTN + j * 16 + fq * 4;
                rres[j] = half4{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                rb2[j] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (nb0 >= g.N) continue;
                if (g.res)
                    rres[j] = *reinterpret_cast<const half4*>(g.res + (size_t)z * g.strideRes + (size_t)(g.res_rows ? m % g.res_rows : m) * g.ldr + nb0);
                if (b2_global) rb2[j] = *reinterpret_cast<const floatx4*>(g.bias2 + (size_t)b * g.ldb2 + nb0);
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nb0 = n0 + wn * WTN + j * 16 + fq * 4;
            if (nb0 >= g.N) continue;
            // same arithmetic as gemm_epilogue_fast (bias + per-sample bias first, then ONE fma with
            // alpha), so a tensor computed partly by full and partly by edge tiles -- or by both
            // kernels at different batch sizes -- rounds identically
            float v[4];
            floatx4 bsum = {0.f, 0.f, 0.f, 0.f};
            if (g.bias) {
                if (bias_tile) bsum = *reinterpret_cast<const __attribute__((address_space(3))) floatx4*>(bias_tile + (nb0 - n0));
                else bsum = *reinterpret_cast<const floatx4*>(g.bias + nb0);
            }
            floatx4 b2v = {0.f, 0.f, 0.f, 0.f};
            if (g.bias2) {
                if (bias2_tile) b2v = *reinterpret_cast<const __attribute__((address_space(3))) floatx4*>(bias2_tile + (nb0 - n0));
                else if constexpr (BATCH) b2v = rb2[j];
                else b2v = *reinterpret_cast<const floatx4*>(g.bias2 + (size_t)b * g.ldb2 + nb0);
            }
            if (LN && g.ln_stats) {   // LayerNorm fold (see gemm_epilogue_fast): bias2 carries colsum(W')
                floatx2 st;
                if (stats_tile) st = *reinterpret_cast<lds_cf2>(stats_tile + 2 * (m - m0));
                else st = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)m);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[i][j][r], st[0], fmaf(st[1], b2v[r], bsum[r]));
            } else {
                bsum += b2v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(acc[i][j][r], g.alpha, bsum[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_apply(v[r], g.act);
            if (g.res) {
                half4 rr;
                if constexpr (BATCH) rr = rres[j];
                else rr = *reinterpret_cast<const half4*>(g.res + (size_t)z * g.strideRes + (size_t)(g.res_rows ? m % g.res_rows : m) * g.ldr + nb0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            }
            if (g.out_f32) {
                float* C = reinterpret_cast<float*>(g.C) + (size_t)z * g.strideC +
                           (size_t)m * g.ldc + nb0;
                if (nb0 + 3 < g.N) {
                    *reinterpret_cast<float4*>(C) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    for (int r = 0; r < 4 && nb0 + r < g.N; ++r) C[r] = v[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) oh[j][r] = (half_t)v[r];
            }
        }
        if (g.out_f32) continue;
        half_t* Crow = reinterpret_cast<half_t*>(g.C) + (size_t)z * g.strideC + (size_t)m * g.ldc;
#pragma unroll
        for (int j = 0; j < NI; j += 2) {
            const int cb = n0 + wn * WTN + j * 16;
            if (j + 1 < NI && cb + 32 <= g.N && (g.ldc & 7) == 0) {   // wave-uniform
                const u32x2 x = __builtin_bit_cast(u32x2, oh[j]);
                const u32x2 y = __builtin_bit_cast(u32x2, oh[j + 1 < NI ? j + 1 : j]);
                unsigned x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
                swap16(x0, y0);
                swap16(x1, y1);
                *reinterpret_cast<u32x4*>(Crow + cb + pcol) = u32x4{x0, x1, y0, y1};
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (j + u >= NI) continue;
                    const int nb0 = cb + u * 16 + fq * 4;
                    half_t* C = Crow + nb0;
                    const half4 o = oh[j + u < NI ? j + u : j];
                    if (nb0 + 3 < g.N) {
                        *reinterpret_cast<half4*>(C) = o;
                    } else {
                        for (int r = 0; r < 4 && nb0 + r < g.N; ++r) C[r] = o[r];
                    }
                }
            }
        }
    }
}

// ---- gemm_pp.hip: the deep-pipelined ping-pong kernels (tile ids 30..) --------------------------------------------
// 1 when tile id `tile` can run the problem in `g` (full tiles, K a multiple of 64, shapes the loop's scalar DMA
// addressing covers); fd_gemm_pp_launch runs it (g.split_k, g.tap_fast etc. already set by the caller).
bool fd_gemm_pp_ok(const GemmArgs& g, int batch, int tile);
int fd_gemm_pp_launch(GemmArgs& g, int batch, hipStream_t st, int tile);
void fd_gemm_pp_tile_shape(int tile, int* bm, int* bn);
