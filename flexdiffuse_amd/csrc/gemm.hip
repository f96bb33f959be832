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
template <int MI, int NI, int ACT, bool RES, bool B2, bool LNF = false, bool STATS = false, int WN_ = 1>
__device__ __forceinline__ void gemm_epilogue_fast(const GemmArgs& g, floatx4 (&acc)[MI][NI], int row0,
                                                   int col0, int coll, int fq, int z,
                                                   lds_cfloat bias_tile, lds_cfloat bias2_tile,
                                                   float* xch = nullptr, int trow0 = 0, int wn = 0, int m0 = 0) {
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
        if constexpr (LNF) st_next = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)row0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            half_t* Crow = Cb + (size_t)i * 16 * g.ldc;
            half4 og[NP];
            const floatx2 st = st_next;
            if constexpr (LNF) {
                if (i + 1 < MI) st_next = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)(row0 + (i + 1) * 16));
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
        const half_t* Rb = RES ? g.res + (size_t)z * g.strideRes + (size_t)row0 * g.ldr + col0 + fq * 4 : nullptr;
        floatx2 st_next = {g.alpha, 0.f};   // LNF: one row block ahead (see the GEGLU branch)
        if constexpr (LNF) st_next = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)row0);
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
                if (i + 1 < MI) st_next = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)(row0 + (i + 1) * 16));
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

// Fused epilogue shared by the register-staged and the LDS-DMA main loops.
template <int BM, int BN, bool TRANS, int WM = 2, int WN = 2, bool LN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g,
                                              floatx4 (&acc)[BM / WM / 16][BN / WN / 16], int m0, int n0,
                                              int wm, int wn, int fr, int fq, int z,
                                              lds_cfloat bias_tile = nullptr, lds_cfloat bias2_tile = nullptr, int kslice = -1) {
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
                    const floatx4 s01 = *reinterpret_cast<const floatx4*>(g.ln_stats + 2 * (size_t)mb);
                    const floatx4 s23 = *reinterpret_cast<const floatx4*>(g.ln_stats + 2 * (size_t)mb + 4);
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
                    const floatx2 st = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)m);
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
                    rres[j] = *reinterpret_cast<const half4*>(g.res + (size_t)z * g.strideRes + (size_t)m * g.ldr + nb0);
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
                const floatx2 st = *reinterpret_cast<const floatx2*>(g.ln_stats + 2 * (size_t)m);
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
                else rr = *reinterpret_cast<const half4*>(g.res + (size_t)z * g.strideRes + (size_t)m * g.ldr + nb0);
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
            a_off[i] = b * g.Hi * g.Wi * g.Cin + ck * 8;
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
                const u32x4 v = *(gptr_u4)(Ab + a_off[i] + (iy * g.Wi + ix) * g.Cin + ci0); \
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
    const bool b2_staged = g.bias2 && g.bias_lds && (min(m0 + BM, g.M) - 1) / g.rows_per_batch == b_first;
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
            a_off[i] = b * g.Hi * g.Wi * g.Cin + ck * 8;
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
                    a_voff[i] = ok ? (unsigned)(a_off[i] + (iy * g.Wi + ix) * g.Cin) * 2u   \
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
                                                       b2_staged ? (lds_cfloat)(bias_s + BN) : (lds_cfloat) nullptr, kslice);
    else if constexpr (EPI == 8 || EPI == 9) {   // lean + LayerNorm statistics of the written rows; the
        // exchange buffer reuses stage 0 (the 2-stage K loop ends with a barrier: the stages are dead; the 3-stage loop
        // has its barrier at the TOP of an iteration, so the last K-tile may still be read: one more barrier)
        if constexpr (NS == 3) __syncthreads();
        gemm_epilogue_fast<MI, NI, FD_ACT_NONE, EPI == 9, true, false, true, WN>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            reinterpret_cast<float*>(smem), wm * WTM + fr, wn, m0);
    } else
        gemm_epilogue_fast<MI, NI, (EPI == 3 || EPI == 6) ? FD_ACT_GEGLU : FD_ACT_NONE, EPI == 2, true, (EPI == 5 || EPI == 6)>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN));
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
                a_off[i] = b * g.Hi * g.Wi * g.Cin + ck * 8;                                \
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
                    a_voff[i] = ok ? (unsigned)(a_off[i] + (iy * g.Wi + ix) * g.Cin) * 2u   \
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
        if (g.res) rr = *reinterpret_cast<const half4*>(g.res + (size_t)m * g.ldr + nb0);
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

// --------------------------------------------------------------------------------------
static bool g_use_dma = getenv("FD_GEMM_NO_DMA") == nullptr;
static const int g_vae15 = 1;   // 256x256 tiles on the VAE widths (A/B closed in round 1: +19..33 %)
static int g_tap_fast = getenv("FD_CONV_TAPFAST") ? atoi(getenv("FD_CONV_TAPFAST")) : 1;   // 1: 256x320 tile, 2: every conv tile
static int g_bias_lds = getenv("FD_GEMM_BIAS_LDS") ? atoi(getenv("FD_GEMM_BIAS_LDS")) : 1;
static int g_fast_epi = getenv("FD_GEMM_FAST_EPI") ? atoi(getenv("FD_GEMM_FAST_EPI")) : 1;   // 0: generic epilogue only (A/B)
// 0 = never, 1 = short-K GEMMs only (default), 2 = always
static int g_persist_mode = getenv("FD_GEMM_PERSIST") ? atoi(getenv("FD_GEMM_PERSIST")) : 1;

template <int BM, int BN, bool TRANS, bool CONV, int WM = 2, int NS = 2, int WN = 2, int EPI = 0>
static int launch_mode(GemmArgs& g, int batch, hipStream_t st) {
    g.tiles_m = fd_cdiv(g.M, BM);
    g.tiles_n = fd_cdiv(g.N, BN);
    const size_t lds = NS * (size_t)(BM + BN) * 128 + 4 * BN * sizeof(float);   // stages + 2 x (bias, bias2 / colsum) tiles
    dim3 grid(g.tiles_m * g.tiles_n, g.split_k, batch);
    // tensor extents for the buffer descriptors of the LDS-DMA loop (must fit 32 bits)
    const unsigned long long a_bytes =
        CONV ? 2ull * (g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi * g.Cin
             : 2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K);
    const unsigned long long w_bytes = 2ull * ((unsigned long long)(g.N - 1) * g.ldw + g.K + g.K2);
    if (g_use_dma && a_bytes < 0x7fffffffull && w_bytes < 0x7fffffffull) {
        static bool configured = false;
        if (!configured && lds > 64 * 1024) {
            FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_dma<BM, BN, CONV, WM, TRANS, NS, WN, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if constexpr (BN != 320)   // the 256x320 tile has no persistent form (it would spill)
                FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_dmap<BM, BN, CONV, WM, TRANS, WN, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = true;
        }
        // Short K loops: persistent workgroups (<= 256 CUs x co-resident workgroups) walk the
        // tile list with the next tile's first K-tile prefetched under the epilogue.
        const int occ = (BM >= 256 || lds > 80 * 1024) ? 1 : (BM * BN >= 128 * 128) ? 2 : (BM == 128 ? 3 : 4);  // workgroups / CU
        const int slots = 256 * occ;
        const int nkt = (g.K + BK - 1) / BK / g.split_k;
        const bool persistent = NS == 2 && BN != 320 && g.K2 == 0 && !g.phase && !g.ln_stats_out && g.strideBias == 0 && (g_persist_mode == 2 ||
                                (g_persist_mode == 1 && nkt <= 20 && g.tiles_m * g.tiles_n > slots));
        if constexpr (EPI >= 1 && EPI <= 3) {
            // the persistent loop does not stage the per-sample bias: generic epilogue there
            if (persistent && g.bias2) return launch_mode<BM, BN, TRANS, CONV, WM, NS, WN, 0>(g, batch, st);
        }
        if (persistent) {
            if constexpr (BN != 320) {
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
        static bool configured = false;
        if (!configured && lds > 64 * 1024) {
            FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16<BM, BN, TRANS, CONV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            configured = true;
        }
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

extern "C" int fd_gemm_f16(const fd_gemm_desc* d, void* stream) {
    if (fd_plan_recording() && d) {
        const fd_gemm_desc dc_ = *d;
        fd_plan_push([dc_](void* fd_s_) -> int { return fd_gemm_f16(&dc_, fd_s_); });
    }
    FD_CHECK_ARG(d && d->A && d->W && d->C, FD_EINVAL, "fd_gemm_f16: null pointer");
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
    g.lda = d->lda; g.ldw = d->ldw; g.ldc = d->ldc; g.ldr = d->ldr;
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
        fd_prof_begin(FD_FAMILY_GEMM, st, flops, flops_exec);
        // 128x160 with 8 waves where 160 | N and the rows fill the chip: -16..18 % on the 64x64 / 32x32
        // self-attention V projections (tools/ab_vt.py; 256x160 / 16 waves is no better, 16x16 maps tie).
        // FD_GEMM_VT_TILE=0: the 128x64 / 4-wave tile everywhere (A/B)
        static const int vt_tile = getenv("FD_GEMM_VT_TILE") ? atoi(getenv("FD_GEMM_VT_TILE")) : 9;
        if (vt_tile && g.N % 160 == 0 && batch == 1 && g.M >= 8192 &&
            2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K) < 0x7fffffffull)
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
    } else if (g.N <= 64) {
        best_tile = (g.M <= 64) ? 4 : 3;
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
            g.mode == MODE_CONV ? (unsigned long long)(g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi * g.Cin
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
            default: break;
        }
        FD_CHECK_ARG(bm && g.M % bm == 0 && g.N % bn == 0 && best_split == 1, FD_ESHAPE,
                     "fd_gemm_f16: upsample2x == 2: tile %d / split %d cannot run M=%d N=%d through the lean epilogue",
                     best_tile, best_split, g.M, g.N);
    }
    if (best_split > 1)
        FD_CHECK_ARG(!geglu && batch == 1 && g.N % 4 == 0 && g.ws &&
                         (size_t)best_split * g.M * g.N * 4 <= (size_t)d->workspace_bytes,
                     FD_ESHAPE, "fd_gemm_f16: split_k=%d not possible for this problem", best_split);
    if (geglu && (best_tile == 2 || best_tile == 5 || best_tile == 7 || best_tile == 9 || best_tile == 12 || best_tile == 13 || best_tile == 16 || best_tile == 20 || best_tile == 23)) best_tile = 1;
    g.split_k = best_split;
    g.tap_fast = g.mode == MODE_CONV && (g_tap_fast == 2 || (g_tap_fast == 1 && best_tile == 16));
    fd_prof_begin(FD_FAMILY_GEMM, st, flops, flops_exec);
    if (g.ln_stats && !(best_tile == 9 || best_tile == 10 || (best_tile >= 12 && best_tile <= 16) || best_tile == 20 || best_tile == 23)) {
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
        // (a 3-stage form of tile 13 -- launch_epi<256, 160, 8, 3, 2, 6>: 3 x 53,248 B + bias tiles = 162,304 B, it does fit the 160 KiB --
        //  measured identical to the 2-stage tile on every deep-level convolution, profiles/r04_session_ab.txt sec. 5: not instantiated)
        default: rc = launch<128, 128, false>(g, batch, st); break;
    }
    if (rc == FD_OK && g.split_k > 1) {
        const size_t total = (size_t)g.M * (g.N / 4);
        const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_splitk_finish, dim3(blocks), dim3(256), 0, st, g);
        FD_CHECK_LAUNCH("k_splitk_finish");
    }
    fd_prof_end(FD_FAMILY_GEMM, st);
    return rc;
}
