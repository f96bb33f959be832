// Fused cross-attention front half for gfx950: LayerNorm-fold q projection + softmax(Q K^T) V over
// the SHORT, step-invariant text context, in ONE launch.
//
// Replaces, inside `unet(...)` (reference pipeline/guide.py:56-58; diffusers' BasicTransformerBlock
// attn2), the q-projection GEMM launch and the cross-attention launch at the 64x64 level
// (C = 320 = 8 heads x 40): the query matrix (42 MB at CFG batch 16) is no longer written to and
// re-read from HBM.  The hidden states are read once, the attention output is written once.
//
// Structure: the 256x320 / 16-wave tile of gemm.hip (64x80 wave tiles, LDS-DMA K loop, LayerNorm
// fold) computes Q for 256 rows x all 8 heads; wave column wn owns the 80 columns of head pair
// (2 wn, 2 wn + 1).  The accumulators -- rounded to fp16 exactly like the q tensor the unfused path
// stores -- ARE the B operands of S^T = K Q^T (a lane owns one query column): no exchange.  The
// context's K and V^T (77 keys) are step-invariant, so they are packed ONCE per context into the
// per-lane MFMA fragment order ("images", fd_xattn_pack_kv_f16); the kernel DMAs the 60 KB K image
// into the K-loop stage that is already dead during the last K-tile and the 60 KB V^T image into
// the other stage right after the loop, and reads fragments lane-linearly (conflict-free).
// Softmax over all 77 keys happens in one piece (exact row max, no online rescale); the
// denominator rides through the PV MFMA on a ones-row of the V^T image (row 40 of 48).
//
// Head dim 40 on 16-wide fragments: the wave's fragment 2 (columns 32..47) holds head A's channels
// 32..39 and head B's channels 0..7.  Instead of padding, the K image of each head carries zeros
// at the other head's k-slots of that fragment, so the contraction picks exactly its own channels;
// the 8-channel remainder uses v_mfma_f32_16x16x16_f16 (half the fragment bytes of the K = 32 form).
#include <stdlib.h>

#include "common.h"

#define XBK 64
typedef unsigned int xu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int xu32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float* xlds_cfloat;
typedef const __attribute__((address_space(3))) floatx4* xlds_cf4;

struct XattnArgs {
    const half_t* A;        // hidden states [M][lda], un-normalised
    const half_t* W;        // [C][ldw] gain-folded, pre-scaled q weights
    const float* bias;      // [C] folded bias
    const float* colsum;    // [C] column sums of the folded weights
    const float* ln_stats;  // [M][2] (rstd, -mean rstd)
    const char* kimg;       // [samples][n-tiles][image] K images
    const char* vimg;       // [samples][n-tiles][image] V^T images
    half_t* O;              // [nrep * M][ldo]
    int M, lda, ldw, ldo;
    int rows_per_sample, samples_per_rep, n_keys;
    int ln_parts;           // > 0: ln_stats holds the producer's k partial slabs [k][M][2] (sum, sum of squares): every tile finalises its own rows
    float ln_inv_n, ln_eps;
};

constexpr int XA_KB = 5;   // 16-key blocks (80 >= 77 keys)

// Shape classes.  D = 40 (C = 320: the 64x64 level of SD1.x): 256-row tiles, a wave owns a head PAIR (80 columns);
// D = 80 (C = 640: the 32x32 level): 128-row tiles, two 320-column n-tiles of four heads, a wave owns ONE head.
template <int D> struct XaCfg;
template <> struct XaCfg<40> {
    static constexpr int BM = 256, HPW = 2, HEADS_TILE = 8, KSF = 1, DT = 3, C = 320;
    static constexpr bool ONES = true;
};
template <> struct XaCfg<80> {
    static constexpr int BM = 128, HPW = 1, HEADS_TILE = 4, KSF = 2, DT = 5, C = 640;
    static constexpr bool ONES = false;
};
template <int D> constexpr int xa_khead() { return XA_KB * (XaCfg<D>::KSF * 1024 + 512); }   // K image bytes per head
template <int D> constexpr int xa_vhead() { return XaCfg<D>::DT * 2560; }                    // V^T image bytes per head
template <int D> constexpr int xa_kimg() { return XaCfg<D>::HEADS_TILE * xa_khead<D>(); }    // per (sample, n-tile)
template <int D> constexpr int xa_vimg() { return XaCfg<D>::HEADS_TILE * xa_vhead<D>(); }

// ---- image packing (once per context) ---------------------------------------------------------
// K image, head h, key block kb (key = kb*16 + fr, lane = (fr, fq)): KSF fragments of 8 halfs per lane, then a 4-half tail.
//   D = 40, even head (A of its pair): k32 = {ch fq*4 + 0..3, ch 16 + fq*4 + 0..3}; tail = ch 32 + fq*4 + 0..3 for fq < 2, else 0
//   D = 40, odd head  (B of its pair): k32 = {ch 8 + fq*4 + 0..3, ch 24 + fq*4 + 0..3}; tail = ch (fq-2)*4 + 0..3 for fq >= 2, else 0
//     (the wave's fragment 2 holds A's channels 32..39 and B's channels 0..7: each head's image is zero at the other's k-slots)
//   D = 80: k32[ks] = {ch ks*32 + fq*4 + 0..3, ch ks*32 + 16 + fq*4 + 0..3}, tail = ch 64 + fq*4 + 0..3
// V^T image, head h, d-tile dt (row d = dt*16 + fr), key group kg: kg 0 / 1 (8 halfs): {key (2kg)*16 + fq*4 + 0..3,
//   key (2kg+1)*16 + fq*4 + 0..3}; kg 2 (4 halfs): key 64 + fq*4 + 0..3.  D = 40: row 40 = ones (the softmax denominator
//   rides through the PV MFMA), rows 41..47 zero.  Keys >= n_keys are zero everywhere (ones-row included).
template <int D>
__global__ void k_xattn_pack(const half_t* __restrict__ K, const half_t* __restrict__ Vt, char* __restrict__ kimg,
                             char* __restrict__ vimg, int L, int ldk, int ldvt, long long sK, long long sVt, int heads) {
    typedef XaCfg<D> cfg;
    const int b = blockIdx.y, h = blockIdx.x;
    const half_t* Kb = K + (size_t)b * sK + h * D;
    const half_t* Vb = Vt + (size_t)b * sVt + (size_t)h * D * ldvt;
    // images of a sample: [n-tile][head within tile]; heads are consecutive, so head h sits at h * per-head bytes
    half_t* ki = reinterpret_cast<half_t*>(kimg + ((size_t)b * heads + h) * xa_khead<D>());
    half_t* vi = reinterpret_cast<half_t*>(vimg + ((size_t)b * heads + h) * xa_vhead<D>());
    constexpr int KBH = (cfg::KSF * 1024 + 512) / 2;    // halfs per key block
    const bool odd = h & 1;
    for (int e = threadIdx.x; e < XA_KB * 64; e += blockDim.x) {
        const int kb = e >> 6, lane = e & 63, fr = lane & 15, fq = lane >> 4;
        const int key = kb * 16 + fr;
        for (int ks = 0; ks < cfg::KSF; ++ks) {
            half_t* o32 = ki + kb * KBH + ks * 512 + lane * 8;
            for (int i = 0; i < 8; ++i) {
                int ch;
                if (D == 40) ch = (odd ? 8 : 0) + (i < 4 ? fq * 4 + i : 16 + fq * 4 + (i - 4));
                else ch = ks * 32 + (i < 4 ? fq * 4 + i : 16 + fq * 4 + (i - 4));
                o32[i] = key < L ? Kb[(size_t)key * ldk + ch] : (half_t)0.f;
            }
        }
        half_t* o16 = ki + kb * KBH + cfg::KSF * 512 + lane * 4;
        for (int i = 0; i < 4; ++i) {
            half_t v = (half_t)0.f;
            if (key < L) {
                if (D == 40) {
                    if (!odd && fq < 2) v = Kb[(size_t)key * ldk + 32 + fq * 4 + i];
                    if (odd && fq >= 2) v = Kb[(size_t)key * ldk + (fq - 2) * 4 + i];
                } else {
                    v = Kb[(size_t)key * ldk + 64 + fq * 4 + i];
                }
            }
            o16[i] = v;
        }
    }
    for (int e = threadIdx.x; e < cfg::DT * 3 * 64; e += blockDim.x) {
        const int lane = e & 63, kg = (e >> 6) % 3, dt = e / 192, fr = lane & 15, fq = lane >> 4;
        const int d = dt * 16 + fr;
        half_t* o = vi + dt * 1280 + (kg == 0 ? 0 : kg == 1 ? 512 : 1024) + lane * (kg == 2 ? 4 : 8);
        const int n = kg == 2 ? 4 : 8;
        for (int i = 0; i < n; ++i) {
            const int key = kg == 2 ? 64 + fq * 4 + i
                                    : (i < 4 ? (2 * kg) * 16 + fq * 4 + i : (2 * kg + 1) * 16 + fq * 4 + (i - 4));
            half_t v = (half_t)0.f;
            if (key < L) {
                if (d < D) v = Vb[(size_t)d * ldvt + key];
                else if (cfg::ONES && d == D) v = (half_t)1.f;
            }
            o[i] = v;
        }
    }
}

static bool xa_supported(int heads, int head_dim) { return heads == 8 && (head_dim == 40 || head_dim == 80); }

extern "C" int64_t fd_xattn_image_bytes(int heads, int head_dim) {
    if (!xa_supported(heads, head_dim)) return 0;
    return head_dim == 40 ? (int64_t)heads * xa_khead<40>() : (int64_t)heads * xa_khead<80>();   // K and V^T images: same size
}

extern "C" int fd_xattn_pack_kv_f16(const void* K, const void* Vt, void* kimg, void* vimg, int samples, int n_keys,
                                    int heads, int head_dim, int ldk, int ldvt, int64_t k_sample_stride,
                                    int64_t vt_sample_stride, void* stream) {
    FD_PLAN(fd_xattn_pack_kv_f16(K, Vt, kimg, vimg, samples, n_keys, heads, head_dim, ldk, ldvt, k_sample_stride,
                                 vt_sample_stride, fd_s_));
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(4u, __LINE__));
    FD_CHECK_ARG(K && Vt && kimg && vimg && samples > 0, FD_EINVAL, "fd_xattn_pack_kv_f16: args");
    FD_CHECK_ARG(xa_supported(heads, head_dim) && n_keys >= 1 && n_keys <= 80, FD_ESHAPE,
                 "fd_xattn_pack_kv_f16: 8 heads x 40 or 8 x 80, at most 80 keys (got %d x %d, %d keys)", heads, head_dim, n_keys);
    FD_CHECK_ARG(ldvt >= n_keys && ldk >= heads * head_dim, FD_ESHAPE, "fd_xattn_pack_kv_f16: leading dimensions");
    static_assert(xa_khead<40>() == xa_vhead<40>() && xa_khead<80>() == xa_vhead<80>(), "K and V^T images share a size");
    if (head_dim == 40)
        hipLaunchKernelGGL(k_xattn_pack<40>, dim3(heads, samples), dim3(256), 0, (hipStream_t)stream, (const half_t*)K,
                           (const half_t*)Vt, (char*)kimg, (char*)vimg, n_keys, ldk, ldvt, (long long)k_sample_stride,
                           (long long)vt_sample_stride, heads);
    else
        hipLaunchKernelGGL(k_xattn_pack<80>, dim3(heads, samples), dim3(256), 0, (hipStream_t)stream, (const half_t*)K,
                           (const half_t*)Vt, (char*)kimg, (char*)vimg, n_keys, ldk, ldvt, (long long)k_sample_stride,
                           (long long)vt_sample_stride, heads);
    FD_CHECK_LAUNCH("k_xattn_pack");
    return FD_OK;
}

// ---- the fused kernel -------------------------------------------------------------------------
// grid: (row tiles, context replicas, 320-column n-tiles).  Replicas sharing the queries (CFG fan-out of a shared prefix)
// are a grid dimension: the q tile is recomputed per replica -- at those shapes half the CUs would otherwise idle.
template <int D>
__global__ __launch_bounds__(1024) void k_xattn(XattnArgs g, unsigned a_bytes, unsigned w_bytes, unsigned o_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef XaCfg<D> cfg;
    constexpr int BM = cfg::BM, BN = 320, WN = 4, NW = 16;
    constexpr int WTM = BM / 4, WTN = 80, MI = WTM / 16, NI = 5;
    constexpr int AG = BM / 8, BG = BN / 8;            // 8-row DMA groups
    constexpr int AR = AG / NW, BR = (BG + NW - 1) / NW;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int KIMG = xa_kimg<D>(), VIMG = xa_vimg<D>(), KHEAD = xa_khead<D>(), VHEAD = xa_vhead<D>();
    constexpr int KBB = cfg::KSF * 1024 + 512;          // K image bytes per key block
    static_assert(STAGE >= KIMG && STAGE >= VIMG, "an image must fit one dead K-loop stage");
    static_assert(AG % NW == 0 && MI % 2 == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order (workgroups round-robin over the 8 XCDs; grid.x is what varies fastest): each XCD takes a
    // contiguous range of row tiles, i.e. whole samples, so a sample's K / V^T images are fetched into ONE XCD's L2
    // instead of all eight (PMC, round-robin order: 60.1 MB read per launch for 42 MB of hidden states + 2 MB of images)
    int tile = blockIdx.x;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = tile & 7, slot = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = tile * BM;
    const int rep = blockIdx.y, nt = blockIdx.z, ntiles = gridDim.z;
    const int n0 = nt * BN;
    const int fr = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, w_bytes, 0x00020000);
    float* bias_s = reinterpret_cast<float*>(smem + 2 * STAGE);   // [320] bias | [320] colsum | [BM][2] row statistics
    float* stats_s = bias_s + 2 * BN;
    {
        // the tile's LayerNorm row statistics ride along too: a global load in the epilogue would be waited
        // for with vmcnt(0), i.e. together with the V^T image DMA issued just before it
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(g.ln_stats + 2 * (size_t)m0), 0, BM * 8u, 0x00020000);
        if (g.ln_parts) {
            // the producer's partial sums (fd_xattn_desc.ln_stats_parts): fd_ln_finalize_stats_f32's arithmetic, bit for bit (fixed slab order), without
            // the launch; the LDS writes are ordered before their readers by the barriers of the projection's K loop
            const int t = wave * 64 + lane;
            if (t < BM) {
                const float* p = g.ln_stats + 2 * (size_t)(m0 + t);
                float s1 = 0.f, s2 = 0.f;
                for (int k = 0; k < g.ln_parts; ++k) {
                    const floatx2 v = *reinterpret_cast<const floatx2*>(p + 2 * (size_t)k * g.M);
                    s1 += v[0];
                    s2 += v[1];
                }
                const float mean = s1 * g.ln_inv_n;
                const float var = fmaxf(fmaf(-mean, mean, s2 * g.ln_inv_n), 0.f);
                const float rstd = rsqrtf(var + g.ln_eps);
                *reinterpret_cast<floatx2*>(stats_s + 2 * t) = floatx2{rstd, -mean * rstd};
            }
        } else if (wave < BM * 2 / 64) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (lds_ptr)(stats_s + wave * 64), 4, (unsigned)(wave * 64 + lane) * 4u, 0, 0, 0);
        }
    }
    {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bias + n0), 0, BN * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(g.colsum + n0), 0, BN * 4u, 0x00020000);
        if (wave * 64 + lane < BN) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(bias_s + wave * 64), 4, (unsigned)(wave * 64 + lane) * 4u, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (lds_ptr)(bias_s + BN + wave * 64), 4, (unsigned)(wave * 64 + lane) * 4u, 0, 0, 0);
        }
    }
    const int rsub = lane >> 3;
    const int ck = (lane & 7) ^ rsub;
    unsigned a_voff[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) a_voff[i] = (unsigned)((m0 + (i * NW + wave) * 8 + rsub) * g.lda + ck * 8) * 2u;
    const unsigned b_voff0 = (unsigned)((n0 + wave * 8 + rsub) * g.ldw + ck * 8) * 2u;
    const int b_group = NW * 8 * g.ldw * 2;
    constexpr int nk = cfg::C / XBK;   // K = C: 5 or 10 K-tiles

#define XA_DMA_TILE(KT, BUF)                                                                     \
    {                                                                                            \
        char* stage = smem + (BUF) * STAGE;                                                      \
        const int soff = (KT) * XBK * 2;                                                         \
        _Pragma("unroll") for (int i = 0; i < AR; ++i)                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(stage + (i * NW + wave) * 1024), 16, a_voff[i], soff, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < BR; ++i)                                           \
            if (i * NW + wave < BG)                                                              \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(stage + BM * 128 + (i * NW + wave) * 1024), 16, b_voff0, soff + i * b_group, 0, 0); \
    }
    // an image (lane-linear, a multiple of 1 KB) into a dead stage: chunk c by wave c % 16
#define XA_DMA_IMAGE(RS, BYTES, BUF)                                                             \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < ((BYTES) / 1024 + NW - 1) / NW; ++i)               \
            if (i * NW + wave < (BYTES) / 1024)                                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_ptr)(smem + (BUF) * STAGE + (i * NW + wave) * 1024), 16, \
                                                         (unsigned)lane * 16u, (i * NW + wave) * 1024, 0, 0); \
    }

    floatx4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // the tile lies in one sample (rows_per_sample % BM == 0); replica r of sample b is image r * samples + b
    const size_t img = ((size_t)rep * g.samples_per_rep + m0 / g.rows_per_sample) * ntiles + nt;
    const __amdgpu_buffer_rsrc_t rsKi = __builtin_amdgcn_make_buffer_rsrc((void*)(g.kimg + img * KIMG), 0, KIMG, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsVi = __builtin_amdgcn_make_buffer_rsrc((void*)(g.vimg + img * VIMG), 0, VIMG, 0x00020000);

    XA_DMA_TILE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frag_a = (wm * WTM + fr) * 128, frag_b = BM * 128 + (wn * WTN + fr) * 128;
    const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;
    int cur = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) {
            XA_DMA_TILE(kt + 1, cur ^ 1);
        } else {
            XA_DMA_IMAGE(rsKi, KIMG, cur ^ 1);   // the other stage is dead during the last K-tile
        }
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            half8 fa[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
            half8 fb[NI];
            fb[0] = *reinterpret_cast<const half8*>(st + frag_b + sw);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (j + 1 < NI) fb[j + 1] = *reinterpret_cast<const half8*>(st + frag_b + (j + 1) * 2048 + sw);
#pragma unroll
                for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    // here: every wave is past the last K-tile; stage `cur` holds the K image, stage `cur ^ 1` is dead
    const char* sK = smem + cur * STAGE;
    const char* sV = smem + (cur ^ 1) * STAGE;
    XA_DMA_IMAGE(rsVi, VIMG, cur ^ 1);

    // ---- Q = LN-fold(acc) rounded to fp16 (exactly what the unfused q projection stores), held directly as MFMA B
    // operands.  D = 40: q0 = fragments {0,1} (head A, channels 0..31), q1 = fragments {3,4} (head B, channels 8..39),
    // qT = fragment 2 zero-extended (A's channels 32..39 | B's 0..7).  D = 80 (one head per wave): q0 = {0,1},
    // q1 = {2,3}, qT = fragment 4 zero-extended -------------------------------------------------------------------
    half8 q0[MI], q1[MI], qT[MI];
    {
        floatx2 st[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
            st[i] = *reinterpret_cast<const __attribute__((address_space(3))) floatx2*>(
                (xlds_cfloat)stats_s + 2 * (wm * WTM + i * 16 + fr));
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const floatx4 bb = *reinterpret_cast<xlds_cf4>((xlds_cfloat)bias_s + wn * WTN + j * 16 + fq * 4);
            const floatx4 cs = *reinterpret_cast<xlds_cf4>((xlds_cfloat)bias_s + BN + wn * WTN + j * 16 + fq * 4);
            constexpr int JT = D == 40 ? 2 : 4;               // the fragment that becomes the zero-extended tail
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const half_t q = (half_t)fmaf(acc[i][j][r], st[i][0], fmaf(st[i][1], cs[r], bb[r]));
                    if (j == JT) { qT[i][r] = q; qT[i][4 + r] = (half_t)0.f; }
                    else {
                        const int jj = j < JT ? j : j - 1;    // the four full fragments in order
                        if (jj == 0) q0[i][r] = q;
                        if (jj == 1) q0[i][4 + r] = q;
                        if (jj == 2) q1[i][r] = q;
                        if (jj == 3) q1[i][4 + r] = q;
                    }
                }
            // one fragment column at a time: its accumulators die as its q halves are born (left alone the
            // scheduler converts everything at once: all accumulators + all operand registers live -> spills)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int kb_last_valid = g.n_keys - 64 - fq * 4;   // key 64 + fq*4 + r is real iff r < kb_last_valid
    // output through a buffer descriptor: 32-bit per-lane offsets instead of 64-bit pointers in VGPRs
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)g.O, 0, o_bytes, 0x00020000);
    const unsigned o_lane = (unsigned)(((m0 + wm * WTM + fr) * g.ldo + n0 + wn * WTN + fq * 4) * 2);
    const unsigned o_rep = (unsigned)rep * (unsigned)g.M * (unsigned)g.ldo * 2u;   // scalar byte offset of the replica
    bool v_ready = false;
#pragma unroll
    for (int hh = 0; hh < cfg::HPW; ++hh) {
        const char* kh = sK + (wn * cfg::HPW + hh) * KHEAD;
        const char* vh = sV + (wn * cfg::HPW + hh) * VHEAD;
#pragma unroll
        for (int ip = 0; ip < MI / 2; ++ip) {
            // ---- S^T = K Q^T for two 16-query blocks (they share every K fragment read) ----
            floatx4 s[2][XA_KB];
#pragma unroll
            for (int kb = 0; kb < XA_KB; ++kb) {
                const half8 ka = *reinterpret_cast<const half8*>(kh + kb * KBB + lane * 16);
                half8 kbf = ka;
                if constexpr (D == 80) kbf = *reinterpret_cast<const half8*>(kh + kb * KBB + 1024 + lane * 16);
                const half4 k16 = *reinterpret_cast<const half4*>(kh + kb * KBB + cfg::KSF * 1024 + lane * 8);
                const half_t z = (half_t)0.f;
                const half8 k16x = {k16[0], k16[1], k16[2], k16[3], z, z, z, z};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = ip * 2 + t;
                    floatx4 v;
                    if constexpr (D == 40) {
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, hh ? q1[i] : q0[i], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    } else {
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, q0[i], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        v = __builtin_amdgcn_mfma_f32_16x16x32_f16(kbf, q1[i], v, 0, 0, 0);
                    }
                    s[t][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k16x, qT[i], v, 0, 0, 0);
                }
            }
            // ---- softmax over the n_keys keys (base-2 logits: the scale is folded into Wq); P held as the
            // B operands of the PV product: pk[t][kg] = key blocks {2 kg, 2 kg + 1}, block 4 zero-extended ----
            half8 pk[2][3];
            float lsum[2] = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r >= kb_last_valid) s[t][4][r] = -INFINITY;
                float mx = fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3]));
#pragma unroll
                for (int kb = 1; kb < XA_KB; ++kb)
                    mx = fmaxf(mx, fmaxf(fmaxf(s[t][kb][0], s[t][kb][1]), fmaxf(s[t][kb][2], s[t][kb][3])));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
                for (int kb = 0; kb < XA_KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const half_t ph = (half_t)__builtin_amdgcn_exp2f(s[t][kb][r] - mx);
                        pk[t][kb >> 1][(kb & 1) * 4 + r] = ph;
                        if constexpr (!cfg::ONES) lsum[t] += (float)ph;   // the denominator of the ROUNDED numerators
                    }
#pragma unroll
                for (int r = 0; r < 4; ++r) pk[t][2][4 + r] = (half_t)0.f;
                if constexpr (!cfg::ONES) {
                    lsum[t] += __shfl_xor(lsum[t], 16, 64);
                    lsum[t] += __shfl_xor(lsum[t], 32, 64);
                }
            }
            if (!v_ready) {   // first use of the V^T image in this launch
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                v_ready = true;
            }
            // ---- O^T = V^T P^T (D = 40: row 40 of the image is all ones: the softmax denominator) ----
            floatx4 o[2][cfg::DT];
#pragma unroll
            for (int dt = 0; dt < cfg::DT; ++dt) {
                const half8 v0 = *reinterpret_cast<const half8*>(vh + dt * 2560 + lane * 16);
                const half8 v1 = *reinterpret_cast<const half8*>(vh + dt * 2560 + 1024 + lane * 16);
                const half4 v2 = *reinterpret_cast<const half4*>(vh + dt * 2560 + 2048 + lane * 8);
                const half_t z = (half_t)0.f;
                const half8 v2x = {v2[0], v2[1], v2[2], v2[3], z, z, z, z};
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    floatx4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v0, pk[t][0], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(v1, pk[t][1], a, 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v2x, pk[t][2], a, 0, 0, 0);
                }
            }
            // ---- normalise and store: lane (query fr, fq) holds d = dt*16 + fq*4 + 0..3 -------
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float l = lsum[t];
                if constexpr (cfg::ONES) l = __shfl(o[t][2][0], 32 + fr, 64);   // O^T row 40 = dt 2, fq 2, element 0
                const float inv = __builtin_amdgcn_rcpf(l);
                const unsigned off = o_lane + (unsigned)(((ip * 2 + t) * 16 * g.ldo + hh * D) * 2);
#pragma unroll
                for (int dt = 0; dt < cfg::DT; ++dt) {
                    half4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[t][dt][r] * inv);
                    // D = 40, dt 2: only d 32..39 (fq < 2) exist; the other lanes aim past the descriptor (dropped)
                    const unsigned vo = (D == 40 && dt == 2 && fq >= 2) ? 0xffffff00u : off + dt * 32;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(xu32x2, v), rsO, vo, o_rep, 0);
                }
            }
        }
    }
#undef XA_DMA_TILE
#undef XA_DMA_IMAGE
#endif
}

template <int D>
static int xa_launch(const XattnArgs& g, int n_rep, unsigned a_bytes, unsigned w_bytes, unsigned o_bytes, hipStream_t st) {
    typedef XaCfg<D> cfg;
    constexpr size_t lds = 2 * (size_t)(cfg::BM + 320) * 128 + (2 * 320 + 2 * cfg::BM) * sizeof(float);
    static std::atomic<unsigned long long> configured{0};
    if (fd_first_on_device(&configured))
        FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_xattn<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_xattn<D>, dim3(g.M / cfg::BM, n_rep, cfg::C / 320), dim3(1024), lds, st, g, a_bytes, w_bytes, o_bytes);
    return FD_OK;
}

extern "C" int fd_xattn_q_f16(const fd_xattn_desc* d, void* stream) {
    if (fd_plan_recording() && d) {
        const fd_xattn_desc dc_ = *d;
        fd_plan_push([dc_](void* fd_s_) -> int { return fd_xattn_q_f16(&dc_, fd_s_); });
    }
    FD_CHECK_ARG(d && d->x && d->wq && d->bias && d->ln_colsum && d->ln_stats && d->k_image && d->v_image && d->out,
                 FD_EINVAL, "fd_xattn_q_f16: null pointer");
    FD_CHECK_ARG(xa_supported(d->heads, d->head_dim), FD_ESHAPE,
                 "fd_xattn_q_f16: 8 heads x 40 or 8 x 80 only (got %d x %d)", d->heads, d->head_dim);
    const int C = d->heads * d->head_dim, bm = d->head_dim == 40 ? 256 : 128;
    FD_CHECK_ARG(d->M > 0 && d->M % bm == 0 && d->rows_per_sample > 0 && d->rows_per_sample % bm == 0 &&
                     d->M % d->rows_per_sample == 0,
                 FD_ESHAPE, "fd_xattn_q_f16: M=%d and rows_per_sample=%d must be multiples of %d", d->M, d->rows_per_sample, bm);
    FD_CHECK_ARG(d->n_keys > 64 && d->n_keys <= 80 && d->n_rep >= 1, FD_ESHAPE, "fd_xattn_q_f16: 65..80 keys, n_rep >= 1 (got %d keys)", d->n_keys);
    FD_CHECK_ARG(d->ldx % 8 == 0 && d->ldw % 8 == 0 && d->ldo % 4 == 0 && d->ldx >= C && d->ldw >= C && d->ldo >= C,
                 FD_ESHAPE, "fd_xattn_q_f16: leading dimensions");
    FD_CHECK_ARG((((uintptr_t)d->x | (uintptr_t)d->wq | (uintptr_t)d->k_image | (uintptr_t)d->v_image | (uintptr_t)d->bias |
                   (uintptr_t)d->ln_colsum | (uintptr_t)d->ln_stats) % 16 == 0) && (uintptr_t)d->out % 8 == 0,
                 FD_ESHAPE, "fd_xattn_q_f16: pointers must be 16-byte aligned");
    const unsigned long long a_bytes = 2ull * ((unsigned long long)(d->M - 1) * d->ldx + C);
    const unsigned long long w_bytes = 2ull * ((unsigned long long)(C - 1) * d->ldw + C);
    const unsigned long long o_bytes = 2ull * ((unsigned long long)((long long)d->n_rep * d->M - 1) * d->ldo + C);
    FD_CHECK_ARG(a_bytes < 0x7fffffffull && o_bytes < 0x7fffffffull, FD_ESHAPE, "fd_xattn_q_f16: tensor >= 2 GiB");
    XattnArgs g;
    g.A = (const half_t*)d->x; g.W = (const half_t*)d->wq; g.bias = d->bias; g.colsum = d->ln_colsum;
    g.ln_stats = d->ln_stats; g.kimg = (const char*)d->k_image; g.vimg = (const char*)d->v_image;
    g.O = (half_t*)d->out;
    g.M = d->M; g.lda = d->ldx; g.ldw = d->ldw; g.ldo = d->ldo;
    g.rows_per_sample = d->rows_per_sample;
    g.samples_per_rep = d->M / d->rows_per_sample; g.n_keys = d->n_keys;
    FD_CHECK_ARG(d->ln_stats_parts == 0 || d->ln_stats_parts == 2 || d->ln_stats_parts == 4 || d->ln_stats_parts == 8, FD_ESHAPE,
                 "fd_xattn_q_f16: ln_stats_parts=%d (0, 2, 4 or 8)", d->ln_stats_parts);
    g.ln_parts = d->ln_stats_parts; g.ln_inv_n = 1.0f / (float)C; g.ln_eps = d->ln_fold_eps > 0.f ? d->ln_fold_eps : 1e-5f;
    hipStream_t st = (hipStream_t)stream;
    // priced like the two launches it replaces: the q projection (2 M C C, once) and the attention proper
    // (4 heads Nq Nk d per replica)
    const double flops = 2.0 * d->M * (double)C * C + 4.0 * (double)d->n_rep * d->M * d->n_keys * C;
    fd_prof_begin(FD_FAMILY_ATTENTION, st, flops, -1.0, fd_tag(5u, d->M, d->n_rep, d->head_dim, d->n_keys));
    const int rc = d->head_dim == 40 ? xa_launch<40>(g, d->n_rep, (unsigned)a_bytes, (unsigned)w_bytes, (unsigned)o_bytes, st)
                                     : xa_launch<80>(g, d->n_rep, (unsigned)a_bytes, (unsigned)w_bytes, (unsigned)o_bytes, st);
    fd_prof_end(FD_FAMILY_ATTENTION, st);
    if (rc != FD_OK) return rc;
    FD_CHECK_LAUNCH("k_xattn");
    return FD_OK;
}
