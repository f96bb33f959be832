// Flash-style attention forward for gfx950 (fp16 in, fp32 softmax/accumulate, MFMA
// v_mfma_f32_16x16x32_f16).  Serves UNet self-attention (N in {4096,1024,256,64}, head dims
// 40/80/160), UNet cross-attention against the 77 tweened text tokens, and the CLIP text
// (causal) / ViT towers (head dim 64).  The N x Nk score matrix never leaves registers.
//
// Layout co-design: both products are computed TRANSPOSED so that every softmax statistic
// is lane-local:
//     S^T = K . Q^T   (A operand = K rows from LDS, B operand = Q fragments held in VGPRs)
//     O^T = V^T . P^T (A operand = V^T rows from LDS, B operand = P from this lane's own S^T)
// A lane owns query column (lane & 15); its S^T registers are, with a fixed permutation of
// the key index inside each group of 32 keys, already the B operand of the second MFMA, so
// P goes from the softmax to the PV product without any cross-lane traffic or LDS round
// trip.  V is consumed as V^T[d][key]; the projection GEMM's transposed-store epilogue
// (gemm.hip) writes it in that form.  K and V^T tiles (64 keys) are staged through LDS with
// a padded row stride (conflict-free ds_read_b128); the next tile's global loads
// are issued before the current tile's MFMAs.
#include <stdlib.h>

#include "common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct AttnArgs {
    const half_t* Q;
    const half_t* K;
    const half_t* Vt;
    half_t* O;
    long long sQ, sK, sVt, sO;  // per-sample strides (elements)
    int ldq, ldk, ldvt, ldo;
    int Nq, Nk, d, heads;
    int causal;
    float scale_log2;  // softmax scale * log2(e)
};

// max over the 4 lanes {l, l^16, l^32, l^48} with the VALU lane-swap instructions
// (v_permlane16_swap / v_permlane32_swap) instead of two LDS-path ds_bpermute round trips.
__device__ __forceinline__ float xor_max_16_32(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned u = __float_as_uint(v);
    auto r16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const unsigned w = __float_as_uint(v);
    auto r32 = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    v = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
#endif
    return v;
}

// QT = 16-row query tiles per wave, NW = waves per workgroup (NW*QT*16 = 128 query rows).
// QT=1/NW=8 trades K/V fragment reuse (2x the LDS reads) for half the registers, i.e. more
// resident waves to hide the serial max -> exp -> PV dependency chain of the softmax.
template <int DQK, int DV, int QT = 2, int NW = 4>
__global__ __launch_bounds__(64 * NW, (DQK <= 96 ? 2 : 1)) void k_attention(AttnArgs a) {
    constexpr int KS = DQK / 32;  // MFMA k-steps of the QK^T product
    constexpr int NT = 64 * NW;   // threads
    constexpr int KCH = DQK / 8;  // 16-byte chunks per K row
    // row strides of KCH+2 / 8+2 sixteen-byte chunks: conflict-free for the ds_read_b128
    // lane groups {0-3,12-15,20-27},... (an odd chunk stride is still 2-way conflicted)
    constexpr int KSTR = (KCH + 2) * 16;
    constexpr int VSTR = 10 * 16;
    constexpr int VROWS = DV * 16;
    constexpr int KLD = (64 * KCH + NT - 1) / NT;  // K chunks per thread
    constexpr int VLD = (VROWS * 8 + NT - 1) / NT; // V^T chunks per thread
    constexpr int KBYTES = 64 * KSTR, VBYTES = VROWS * VSTR;
    __shared__ __attribute__((aligned(16))) char sKV[2 * (KBYTES + VBYTES)];  // two stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    // XCD-aware order (workgroups round-robin over the 8 XCDs): give each XCD a contiguous
    // range of the (batch, head, q-block) list so that all q-blocks of one head re-read its
    // K / V^T from the same 4 MiB L2 (without this every head's K/V is fetched by all 8 XCDs:
    // rocprofv3 FETCH_SIZE 680 MB vs 126 MB algorithmic for 16x8 heads of 4096x40)
    const int nqb = (a.Nq + NW * QT * 16 - 1) / (NW * QT * 16);
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int qb = id % nqb, h = (id / nqb) % a.heads, b = id / (nqb * a.heads);
    const int qblk0 = qb * (NW * QT * 16);
    const int q0 = qblk0 + wave * (QT * 16);
    const int d = a.d;
    const half_t* __restrict__ Qb = a.Q + (size_t)b * a.sQ + h * d;
    const half_t* __restrict__ Kb = a.K + (size_t)b * a.sK + h * d;
    const half_t* __restrict__ Vb = a.Vt + (size_t)b * a.sVt + (size_t)h * d * a.ldvt;
    half_t* __restrict__ Ob = a.O + (size_t)b * a.sO + h * d;

    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    // ---- Q fragments (B operand of S^T = K Q^T): lane = query fr, dims g*8.. of step ks --
    half8 qf[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int q = q0 + t * 16 + fr, d0 = (ks * 4 + g) * 8;
            uint4 v = zero4;
            if (q < a.Nq && d0 < d) v = *reinterpret_cast<const uint4*>(Qb + (size_t)q * a.ldq + d0);
            qf[t][ks] = *reinterpret_cast<half8*>(&v);
        }

    floatx4 o[QT][DV];
    float mrow[QT], lrow[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        mrow[t] = -INFINITY;
        lrow[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DV; ++dt) o[t][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }

    int ntiles = (a.Nk + 63) >> 6;
    if (a.causal) {
        const int qend = min(a.Nq, qblk0 + NW * QT * 16);
        ntiles = min(ntiles, (qend + 63) >> 6);
    }
    const int nk8 = (a.Nk + 7) & ~7;

    uint4 rk[KLD], rv[VLD];
    auto load_kv = [&](int j) {
        const int key0 = j * 64;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int e = tid + NT * i;
            const int row = e / KCH, c = e - row * KCH;
            rk[i] = zero4;
            if (e < 64 * KCH && key0 + row < a.Nk && c * 8 < d)
                rk[i] = *reinterpret_cast<const uint4*>(Kb + (size_t)(key0 + row) * a.ldk + c * 8);
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int e = tid + NT * i;
            const int row = e >> 3, c = e & 7;
            rv[i] = zero4;
            if (e < VROWS * 8 && row < d && key0 + c * 8 < nk8)
                rv[i] = *reinterpret_cast<const uint4*>(Vb + (size_t)row * a.ldvt + key0 + c * 8);
        }
    };
    auto store_kv = [&](int buf) {
        char* sK = sKV + buf * (KBYTES + VBYTES);
        char* sV = sK + KBYTES;
#pragma unroll
        for (int i = 0; i < KLD; ++i) {
            const int e = tid + NT * i;
            const int row = e / KCH, c = e - row * KCH;
            if (e < 64 * KCH) *reinterpret_cast<uint4*>(sK + row * KSTR + c * 16) = rk[i];
        }
#pragma unroll
        for (int i = 0; i < VLD; ++i) {
            const int e = tid + NT * i;
            const int row = e >> 3, c = e & 7;
            if (e < VROWS * 8) {
                // keys 8c..8c+7 of a 32-key group -> permuted so that a lane's 8 P slots
                // {4g..4g+3, 16+4g..16+4g+3} are contiguous
                const int grp = c >> 2, cc = c & 3;
                const int pos = grp * 32 + (cc & 1) * 16 + (cc >> 1) * 4;  // halfs
                char* dst = sV + row * VSTR + pos * 2;
                *reinterpret_cast<uint2*>(dst) = make_uint2(rv[i].x, rv[i].y);
                *reinterpret_cast<uint2*>(dst + 16) = make_uint2(rv[i].z, rv[i].w);
            }
        }
    };

    load_kv(0);
    store_kv(0);
    __syncthreads();
    const float c2 = a.scale_log2;

    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) load_kv(j + 1);
        const char* sK = sKV + (j & 1) * (KBYTES + VBYTES);
        const char* sV = sK + KBYTES;
        // masking is needed only on a ragged last tile or under the causal mask (wave-uniform)
        const bool need_mask = (j * 64 + 64 > a.Nk) || a.causal;
        // ---- S^T tile: 64 keys x (QT x 16) queries ---------------------------------------
        floatx4 s[4][QT];
        const floatx4 zacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(sK + (f * 16 + fr) * KSTR +
                                                                 (ks * 4 + g) * 16);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    s[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[t][ks],
                                                                     ks == 0 ? zacc : s[f][t], 0, 0, 0);
            }
        // ---- online softmax, all statistics lane-local (query = fr) -----------------------
        half8 p[2][QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int q = q0 + t * 16 + fr;
            if (need_mask) {
                // the key base goes through an empty asm INSIDE the branch: hipcc otherwise speculates the whole index /
                // compare arithmetic (16 v_add + 32 v_cmp per query block) above the branch, into every unmasked tile
                int kb = j * 64 + g * 4;
                asm volatile("" : "+v"(kb));
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb + f * 16 + r;
                        if (key >= a.Nk || (a.causal && key > q)) s[f][t][r] = -INFINITY;
                    }
            }
            float tmax = fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3]));
#pragma unroll
            for (int f = 1; f < 4; ++f)
                tmax = fmaxf(fmaxf(tmax, fmaxf(s[f][t][0], s[f][t][1])),
                             fmaxf(s[f][t][2], s[f][t][3]));
            tmax = xor_max_16_32(tmax);
            // running max kept in raw-score units; p = 2^(s*c2 - m*c2) as one fma + v_exp_f32
            const float mnew = fmaxf(mrow[t], tmax);
            const bool dead = mnew == -INFINITY;
            const float mc = dead ? 0.f : mnew * c2;
            const bool moved = mnew != mrow[t];
            if (__any(moved)) {   // wave-uniform: rescale only when some row's running max grew
                const float alpha = (dead || !moved) ? 1.f : __builtin_amdgcn_exp2f((mrow[t] - mnew) * c2);
                lrow[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DV; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
                mrow[t] = mnew;
            }
            // p = 2^(s*c2 - m*c2): packed fp32 fma / add (v_pk_fma_f32, v_pk_add_f32)
            floatx2 ps = {0.f, 0.f};
            const floatx2 c2v = {c2, c2}, mcv = {mc, mc};
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    floatx2 x = {s[f][t][r], s[f][t][r + 1]};
                    x = x * c2v - mcv;
                    floatx2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                    ps += e;
                    p[f >> 1][t][(f & 1) * 4 + r] = (half_t)e[0];
                    p[f >> 1][t][(f & 1) * 4 + r + 1] = (half_t)e[1];
                }
            lrow[t] += ps[0] + ps[1];
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
                const half8 vf = *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR +
                                                                 (kg * 4 + g) * 16);
#pragma unroll
                for (int t = 0; t < QT; ++t)
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[kg][t], o[t][dt], 0, 0, 0);
            }
        if (j + 1 < ntiles) store_kv((j + 1) & 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds O^T[d = dt*16 + g*4 + r][query fr] ---------------
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = lrow[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);   // once per block: the LDS-path shuffle is fine here
        const float inv = 1.0f / l;
        const int q = q0 + t * 16 + fr;
        if (q >= a.Nq) continue;
#pragma unroll
        for (int dt = 0; dt < DV; ++dt) {
            const int d0 = dt * 16 + g * 4;
            if (d0 >= d) continue;
            half4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[t][dt][r] * inv);
            *reinterpret_cast<half4*>(Ob + (size_t)q * a.ldo + d0) = v;
        }
    }
}


// ---------------------------------------------------------------------------------------
// 8-wave / 16-queries-per-wave variant with a VALU-lean softmax.  PMC on the N=4096 d=40
// self-attention showed the original loop VALU-issue-bound (7.4 VALU instructions per score,
// VALU busy ~80 %, MFMA pipe 30 %), so this one removes per-score work:
//  * LAZY running max: scores are exponentiated against a max that is only advanced when some
//    score exceeds it by more than 2^LAZY_THR (wave vote).  The usual tile only needs a
//    lane-local max for the vote: no cross-lane reduce, no rescale of O, no alpha.
//    P <= 2^LAZY_THR stays far inside fp16 range; accumulation is fp32.
//  * PRE: Q already carries scale*log2(e) (folded into the q projection); the running max is
//    then fed to the QK^T MFMA as its accumulator input (-m), so the MFMA result is the
//    exponent itself: no scale/subtract pass at all.
//  * ONES: when head_dim < 16*DV the first padding row of V^T is set to 1.0, so the PV MFMA
//    also produces the softmax denominator (from the same fp16 P): no row-sum adds.
//  * K / V^T tile loads are buffer loads (constant per-thread offsets, zero fill by the
//    bounds check) instead of per-tile 64-bit address math.
#define LAZY_THR 8.0f
#ifndef ATT_VPRE
#define ATT_VPRE 0
#endif
#ifndef ATT_SETPRIO
#define ATT_SETPRIO 0   // s_setprio(1) around the MFMA clusters of k_attention_w8q2 (measured: see profiles)
#endif
#ifndef ATT_W8_MINW
#define ATT_W8_MINW 8   // waves per SIMD the d <= 48 kernel is compiled for (8: 64 VGPRs, 6: 80 VGPRs)
#endif
template <int DQK, int DV, bool PRE, bool ONES>
__global__ __launch_bounds__(512, (DV <= 3 ? ATT_W8_MINW : DQK <= 96 ? 2 : 1)) void k_attention_w8(AttnArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = DQK / 32;
    constexpr int NT = 512;
    constexpr int KCH = DQK / 8;
    constexpr int KSTR = (KCH + 2) * 16;
    constexpr int VSTR = 10 * 16;
    constexpr int VROWS = DV * 16;
    constexpr int KLD = (64 * KCH + NT - 1) / NT;
    constexpr int VLD = (VROWS * 8 + NT - 1) / NT;
    constexpr int KBYTES = 64 * KSTR, VBYTES = VROWS * VSTR;
    constexpr unsigned OOB = 0x7fffffffu;
    __shared__ __attribute__((aligned(16))) char sKV[2 * (KBYTES + VBYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int nqb = (a.Nq + 127) >> 7;
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int qb = id % nqb, h = (id / nqb) % a.heads, b = id / (nqb * a.heads);
    const int qblk0 = qb * 128;
    const int q0 = qblk0 + wave * 16;
    const int d = a.d;
    const half_t* __restrict__ Qb = a.Q + (size_t)b * a.sQ + h * d;
    const half_t* Kb = a.K + (size_t)b * a.sK + h * d;
    const half_t* Vb = a.Vt + (size_t)b * a.sVt + (size_t)h * d * a.ldvt;
    half_t* __restrict__ Ob = a.O + (size_t)b * a.sO + h * d;
    const int nk8 = (a.Nk + 7) & ~7;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Kb, 0, (unsigned)(((size_t)(a.Nk - 1) * a.ldk + d) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Vb, 0, (unsigned)(((size_t)(d - 1) * a.ldvt + nk8) * 2), 0x00020000);

    half8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int q = q0 + fr, d0 = (ks * 4 + g) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < a.Nq && d0 < d) v = *reinterpret_cast<const u32x4*>(Qb + (size_t)q * a.ldq + d0);
        qf[ks] = __builtin_bit_cast(half8, v);
    }

    floatx4 o[DV];
#pragma unroll
    for (int dt = 0; dt < DV; ++dt) o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_used = 0.f;  // running (lazy) max, in base-2 logit units
    floatx4 init = {0.f, 0.f, 0.f, 0.f};  // PRE: -m_used, the QK^T accumulator input
    float lrow = 0.f;

    int ntiles = (a.Nk + 63) >> 6;
    if (a.causal) {
        const int qend = min(a.Nq, qblk0 + 128);
        ntiles = min(ntiles, (qend + 63) >> 6);
    }

    // per-thread constant load offsets / LDS store addresses
    unsigned kvo[KLD], vvo[VLD];
    // LDS store offsets of this thread's K chunk (low 16 bits) and V^T chunk (high 16 bits) in ONE
    // register: at the 64-VGPR budget of 8 waves/SIMD the separate V offset was spilled to scratch
    // and reloaded (scratch_load + vmcnt(0)) in every K/V tile iteration
    unsigned kvst[KLD > VLD ? KLD : VLD];
    static_assert(2 * (KBYTES + VBYTES) < 65536 + 65536, "LDS offsets must fit 16 bits");
    bool vones[VLD], vlive[VLD];
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
        const int e = tid + NT * i;
        const int row = e / KCH, c = e - row * KCH;
        kvo[i] = (e < 64 * KCH && c * 8 < d) ? (unsigned)(row * a.ldk + c * 8) * 2u : OOB;
        kvst[i] = (unsigned)(row * KSTR + c * 16);
    }
#pragma unroll
    for (int i = KLD; i < VLD; ++i) kvst[i] = 0u;
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
        const int e = tid + NT * i;
        const int row = e >> 3, c = e & 7;
        vvo[i] = (e < VROWS * 8 && row < d) ? (unsigned)(row * a.ldvt + c * 8) * 2u : OOB;
        vones[i] = ONES && row == d;
        vlive[i] = e < VROWS * 8 && row < d;
        const int grp = c >> 2, cc = c & 3;
        const int pos = grp * 32 + (cc & 1) * 16 + (cc >> 1) * 4;  // halfs (see k_attention)
        kvst[i] |= (unsigned)(KBYTES + row * VSTR + pos * 2) << 16;
        asm volatile("" : "+v"(kvst[i]));   // opaque: keeps the optimizer from un-packing the two halves again
    }

    u32x4 rk[KLD], rv[VLD];
#define ATT_LOAD(J)                                                                           \
    {                                                                                         \
        const int key0 = (J) * 64;                                                            \
        if (key0 + 64 <= a.Nk) { /* full tile: constant per-lane offsets + scalar tile offset */ \
            _Pragma("unroll") for (int i = 0; i < KLD; ++i)                                   \
                rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[i], key0 * a.ldk * 2, 0); \
            _Pragma("unroll") for (int i = 0; i < VLD; ++i)                                   \
                rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[i], key0 * 2, 0);      \
        } else { /* ragged last tile: rows / key chunks past the end read zero */             \
            _Pragma("unroll") for (int i = 0; i < KLD; ++i) {                                 \
                const unsigned vo = (key0 + (tid + NT * i) / KCH < a.Nk) ? kvo[i] + (unsigned)key0 * a.ldk * 2u : OOB; /* row recomputed: rare path */ \
                rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, vo, 0, 0);                 \
            }                                                                                 \
            _Pragma("unroll") for (int i = 0; i < VLD; ++i) {                                 \
                const unsigned vo = (key0 + ((tid + NT * i) & 7) * 8 < nk8) ? vvo[i] + (unsigned)key0 * 2u : OOB; \
                rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, 0, 0);                 \
            }                                                                                 \
        }                                                                                     \
    }
#define ATT_STORE(BUF)                                                                        \
    {                                                                                         \
        char* sb = sKV + (BUF) * (KBYTES + VBYTES);                                           \
        _Pragma("unroll") for (int i = 0; i < KLD; ++i)                                       \
            if (KLD * NT == 64 * KCH || tid + NT * i < 64 * KCH)                              \
                *reinterpret_cast<u32x4*>(sb + (kvst[i] & 0xffffu)) = rk[i];                               \
        _Pragma("unroll") for (int i = 0; i < VLD; ++i)                                       \
            if (vlive[i]) {                                                                   \
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16)) = u32x2{rv[i][0], rv[i][1]};  \
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16) + 16) = u32x2{rv[i][2], rv[i][3]}; \
            }                                                                                 \
    }

    // V^T rows >= head_dim never change: zero padding, and with ONES row `d` = 1.0 so that the
    // PV MFMA also accumulates the softmax denominator.  Written once, to both stages.
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
        const int e = tid + NT * i;
        if (e < VROWS * 8 && !vlive[i]) {
            const unsigned w = vones[i] ? 0x3C003C00u : 0u;
#pragma unroll
            for (int buf = 0; buf < 2; ++buf) {
                char* sb = sKV + buf * (KBYTES + VBYTES);
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16)) = u32x2{w, w};
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16) + 16) = u32x2{w, w};
            }
        }
    }
    ATT_LOAD(0);
    ATT_STORE(0);
    __syncthreads();
    const float c2 = a.scale_log2;
    const floatx2 c2v = {c2, c2};

    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) ATT_LOAD(j + 1);
        const char* sK = sKV + (j & 1) * (KBYTES + VBYTES);
        const char* sV = sK + KBYTES;
        const bool need_mask = (j * 64 + 64 > a.Nk) || a.causal;
        floatx4 s[4];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(sK + (f * 16 + fr) * KSTR +
                                                                 (ks * 4 + g) * 16);
                s[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], ks == 0 ? init : s[f], 0, 0, 0);
            }
#if ATT_VPRE
        // issue the V^T fragment reads before the softmax so their latency hides under it
        constexpr int VP = DV <= 5 ? DV : 0;
        half8 vfp[2][VP > 0 ? VP : 1];
        if (VP > 0) {
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int dt = 0; dt < VP; ++dt)
                    vfp[kg][dt] = *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR +
                                                                  (kg * 4 + g) * 16);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (!PRE) {
            const floatx2 mv = {m_used, m_used};
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    floatx2 x = {s[f][r], s[f][r + 1]};
                    x = x * c2v - mv;
                    s[f][r] = x[0];
                    s[f][r + 1] = x[1];
                }
        }
        if (need_mask) {
            const int q = q0 + fr;
            int kb = j * 64 + g * 4;   // opaque inside the branch: see k_attention
            asm volatile("" : "+v"(kb));
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kb + f * 16 + r;
                    if (key >= a.Nk || (a.causal && key > q)) s[f][r] = -INFINITY;
                }
        }
        // s = base-2 logits relative to the lazy max.  Lane-local max as 8 v_max3_f32: fmaxf()
        // makes the compiler canonicalise every MFMA result first (16 extra v_max per tile)
        float tmax;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[0][3]), "v"(s[1][0]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[1][1]), "v"(s[1][2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[1][3]), "v"(s[2][0]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[2][1]), "v"(s[2][2]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[2][3]), "v"(s[3][0]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[3][1]), "v"(s[3][2]));
        asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(tmax), "v"(s[3][3]));
        if (j == 0 || __any(tmax > LAZY_THR)) {
            // advance the running max to the exact row max (rare after the first tiles)
            const float t = xor_max_16_32(tmax);
            float delta = (j == 0) ? t : fmaxf(t, 0.f);
            if (delta == -INFINITY) delta = 0.f;
            const float alpha = (j == 0) ? 1.f : __builtin_amdgcn_exp2f(-delta);
            m_used += delta;
            if (PRE) init = floatx4{-m_used, -m_used, -m_used, -m_used};
            lrow *= alpha;
#pragma unroll
            for (int dt = 0; dt < DV; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[f][r] -= delta;
        }
        half8 p[2];
        floatx2 ps = {0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const floatx2 e = {__builtin_amdgcn_exp2f(s[f][r]), __builtin_amdgcn_exp2f(s[f][r + 1])};
                if (!ONES) ps += e;
                const half2v eh = __builtin_convertvector(e, half2v);
                p[f >> 1][(f & 1) * 4 + r] = eh[0];
                p[f >> 1][(f & 1) * 4 + r + 1] = eh[1];
            }
        if (!ONES) lrow += ps[0] + ps[1];
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
#if ATT_VPRE
                const half8 vf = VP > 0 ? vfp[kg][VP > 0 ? dt : 0]
                                        : *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR +
                                                                          (kg * 4 + g) * 16);
#else
                const half8 vf = *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR +
                                                                 (kg * 4 + g) * 16);
#endif
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[kg], o[dt], 0, 0, 0);
            }
        if (j + 1 < ntiles) ATT_STORE((j + 1) & 1);
        __syncthreads();
    }
#undef ATT_LOAD
#undef ATT_STORE

    float l;
    if (ONES) {
        // denominator = O^T row d: lane group g = (d>>2)&3, fragment d>>4, element 0
        float lv = 0.f;
#pragma unroll
        for (int dt = 0; dt < DV; ++dt)
            if (dt == (d >> 4)) lv = o[dt][0];
        l = __shfl(lv, ((d >> 2) & 3) * 16 + fr, 64);
    } else {
        l = lrow;
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    const int q = q0 + fr;
    if (q < a.Nq) {
#pragma unroll
        for (int dt = 0; dt < DV; ++dt) {
            const int d0 = dt * 16 + g * 4;
            if (d0 >= d) continue;
            half4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[dt][r] * inv);
            *reinterpret_cast<half4*>(Ob + (size_t)q * a.ldo + d0) = v;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------
// Two 16-query blocks per wave (256 queries per workgroup).  On this chip the instructions of
// a SIMD's waves ADD UP (tools/micro/pipe_overlap.hip: no overlap between v_mfma_f32_16x16x32
// and VALU / DS issue), so the tile loop is bound by its instruction count per score.  Both
// query blocks share every K and V^T fragment read and the K / V^T tile staging: DS
// instructions and global loads per score halve; MFMA, exp and convert counts per score are
// unchanged.  ~125 VGPRs (4 waves per SIMD: the kernel is not sensitive to occupancy,
// tools/ab_attn_occ.py).  Same arithmetic per query as k_attention_w8.
template <int DQK, int DV, bool PRE, bool ONES>
__global__ __launch_bounds__(512, (DV <= 3 ? 2 : 4)) void k_attention_w8q2(AttnArgs a) {   // (second bound = waves per SIMD; DV 4: keep two workgroups per CU)
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = DQK / 32;
    constexpr int NT = 512;
    constexpr int KCH = DQK / 8;
    constexpr int KSTR = (KCH + 2) * 16;
    constexpr int VSTR = 10 * 16;
    constexpr int VROWS = DV * 16;
    constexpr int KLD = (64 * KCH + NT - 1) / NT;
    constexpr int VLD = (VROWS * 8 + NT - 1) / NT;
    constexpr int KBYTES = 64 * KSTR, VBYTES = VROWS * VSTR;
    constexpr unsigned OOB = 0x7fffffffu;
    __shared__ __attribute__((aligned(16))) char sKV[2 * (KBYTES + VBYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int nqb = (a.Nq + 255) >> 8;
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int qb = id % nqb, h = (id / nqb) % a.heads, b = id / (nqb * a.heads);
    const int qblk0 = qb * 256;
    const int q0 = qblk0 + wave * 32;   // this wave's queries: q0 .. q0+31 (two fragments of 16)
    const int d = a.d;
    const half_t* __restrict__ Qb = a.Q + (size_t)b * a.sQ + h * d;
    const half_t* Kb = a.K + (size_t)b * a.sK + h * d;
    const half_t* Vb = a.Vt + (size_t)b * a.sVt + (size_t)h * d * a.ldvt;
    half_t* __restrict__ Ob = a.O + (size_t)b * a.sO + h * d;
    const int nk8 = (a.Nk + 7) & ~7;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Kb, 0, (unsigned)(((size_t)(a.Nk - 1) * a.ldk + d) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Vb, 0, (unsigned)(((size_t)(d - 1) * a.ldvt + nk8) * 2), 0x00020000);

    half8 qf[2][KS];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int q = q0 + t * 16 + fr, d0 = (ks * 4 + g) * 8;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (q < a.Nq && d0 < d) v = *reinterpret_cast<const u32x4*>(Qb + (size_t)q * a.ldq + d0);
            qf[t][ks] = __builtin_bit_cast(half8, v);
        }

    floatx4 o[2][DV];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < DV; ++dt) o[t][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_used[2] = {0.f, 0.f};  // running (lazy) max per query block, in base-2 logit units
    floatx4 init[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // PRE: -m_used, the QK^T accumulator input
    float lrow[2] = {0.f, 0.f};

    int ntiles = (a.Nk + 63) >> 6;
    if (a.causal) {
        const int qend = min(a.Nq, qblk0 + 256);
        ntiles = min(ntiles, (qend + 63) >> 6);
    }

    // per-thread constant load offsets / LDS store addresses
    unsigned kvo[KLD], vvo[VLD];
    // LDS store offsets of this thread's K chunk (low 16 bits) and V^T chunk (high 16 bits) in ONE
    // register: at the 64-VGPR budget of 8 waves/SIMD the separate V offset was spilled to scratch
    // and reloaded (scratch_load + vmcnt(0)) in every K/V tile iteration
    unsigned kvst[KLD > VLD ? KLD : VLD];
    static_assert(2 * (KBYTES + VBYTES) < 65536 + 65536, "LDS offsets must fit 16 bits");
    bool vones[VLD], vlive[VLD];
#pragma unroll
    for (int i = 0; i < KLD; ++i) {
        const int e = tid + NT * i;
        const int row = e / KCH, c = e - row * KCH;
        kvo[i] = (e < 64 * KCH && c * 8 < d) ? (unsigned)(row * a.ldk + c * 8) * 2u : OOB;
        kvst[i] = (unsigned)(row * KSTR + c * 16);
    }
#pragma unroll
    for (int i = KLD; i < VLD; ++i) kvst[i] = 0u;
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
        const int e = tid + NT * i;
        const int row = e >> 3, c = e & 7;
        vvo[i] = (e < VROWS * 8 && row < d) ? (unsigned)(row * a.ldvt + c * 8) * 2u : OOB;
        vones[i] = ONES && row == d;
        vlive[i] = e < VROWS * 8 && row < d;
        const int grp = c >> 2, cc = c & 3;
        const int pos = grp * 32 + (cc & 1) * 16 + (cc >> 1) * 4;  // halfs (see k_attention)
        kvst[i] |= (unsigned)(KBYTES + row * VSTR + pos * 2) << 16;
        asm volatile("" : "+v"(kvst[i]));   // opaque: keeps the optimizer from un-packing the two halves again
    }

    u32x4 rk[KLD], rv[VLD];
#define ATT_LOAD(J)                                                                           \
    {                                                                                         \
        const int key0 = (J) * 64;                                                            \
        if (key0 + 64 <= a.Nk) { /* full tile: constant per-lane offsets + scalar tile offset */ \
            _Pragma("unroll") for (int i = 0; i < KLD; ++i)                                   \
                rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo[i], key0 * a.ldk * 2, 0); \
            _Pragma("unroll") for (int i = 0; i < VLD; ++i)                                   \
                rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo[i], key0 * 2, 0);      \
        } else { /* ragged last tile: rows / key chunks past the end read zero */             \
            _Pragma("unroll") for (int i = 0; i < KLD; ++i) {                                 \
                const unsigned vo = (key0 + (tid + NT * i) / KCH < a.Nk) ? kvo[i] + (unsigned)key0 * a.ldk * 2u : OOB; /* row recomputed: rare path */ \
                rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, vo, 0, 0);                 \
            }                                                                                 \
            _Pragma("unroll") for (int i = 0; i < VLD; ++i) {                                 \
                const unsigned vo = (key0 + ((tid + NT * i) & 7) * 8 < nk8) ? vvo[i] + (unsigned)key0 * 2u : OOB; \
                rv[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, 0, 0);                 \
            }                                                                                 \
        }                                                                                     \
    }
#define ATT_STORE(BUF)                                                                        \
    {                                                                                         \
        char* sb = sKV + (BUF) * (KBYTES + VBYTES);                                           \
        _Pragma("unroll") for (int i = 0; i < KLD; ++i)                                       \
            if (KLD * NT == 64 * KCH || tid + NT * i < 64 * KCH)                              \
                *reinterpret_cast<u32x4*>(sb + (kvst[i] & 0xffffu)) = rk[i];                               \
        _Pragma("unroll") for (int i = 0; i < VLD; ++i)                                       \
            if (vlive[i]) {                                                                   \
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16)) = u32x2{rv[i][0], rv[i][1]};  \
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16) + 16) = u32x2{rv[i][2], rv[i][3]}; \
            }                                                                                 \
    }

    // V^T rows >= head_dim never change: zero padding, and with ONES row `d` = 1.0 so that the
    // PV MFMA also accumulates the softmax denominator.  Written once, to both stages.
#pragma unroll
    for (int i = 0; i < VLD; ++i) {
        const int e = tid + NT * i;
        if (e < VROWS * 8 && !vlive[i]) {
            const unsigned w = vones[i] ? 0x3C003C00u : 0u;
#pragma unroll
            for (int buf = 0; buf < 2; ++buf) {
                char* sb = sKV + buf * (KBYTES + VBYTES);
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16)) = u32x2{w, w};
                *reinterpret_cast<u32x2*>(sb + (kvst[i] >> 16) + 16) = u32x2{w, w};
            }
        }
    }
    ATT_LOAD(0);
    ATT_STORE(0);
    __syncthreads();
    const float c2 = a.scale_log2;
    const floatx2 c2v = {c2, c2};

    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) ATT_LOAD(j + 1);
        const char* sK = sKV + (j & 1) * (KBYTES + VBYTES);
        const char* sV = sK + KBYTES;
        const bool need_mask = (j * 64 + 64 > a.Nk) || a.causal;
        floatx4 s[2][4];
#if ATT_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(sK + (f * 16 + fr) * KSTR +
                                                                 (ks * 4 + g) * 16);
                s[0][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][ks], ks == 0 ? init[0] : s[0][f], 0, 0, 0);
                s[1][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][ks], ks == 0 ? init[1] : s[1][f], 0, 0, 0);
            }
#if ATT_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        half8 p[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!PRE) {
                const floatx2 mv = {m_used[t], m_used[t]};
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        floatx2 x = {s[t][f][r], s[t][f][r + 1]};
                        x = x * c2v - mv;
                        s[t][f][r] = x[0];
                        s[t][f][r + 1] = x[1];
                    }
            }
            if (need_mask) {
                const int q = q0 + t * 16 + fr;
                int kb = j * 64 + g * 4;   // opaque inside the branch: see k_attention
                asm volatile("" : "+v"(kb));
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kb + f * 16 + r;
                        if (key >= a.Nk || (a.causal && key > q)) s[t][f][r] = -INFINITY;
                    }
            }
            float tmax;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(s[t][0][0]), "v"(s[t][0][1]), "v"(s[t][0][2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][0][3]), "v"(s[t][1][0]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][1][1]), "v"(s[t][1][2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][1][3]), "v"(s[t][2][0]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][2][1]), "v"(s[t][2][2]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][2][3]), "v"(s[t][3][0]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[t][3][1]), "v"(s[t][3][2]));
            asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(tmax), "v"(s[t][3][3]));
            if (j == 0 || __any(tmax > LAZY_THR)) {
                // advance the running max to the exact row max (rare after the first tiles)
                const float tm = xor_max_16_32(tmax);
                float delta = (j == 0) ? tm : fmaxf(tm, 0.f);
                if (delta == -INFINITY) delta = 0.f;
                const float alpha = (j == 0) ? 1.f : __builtin_amdgcn_exp2f(-delta);
                m_used[t] += delta;
                if (PRE) init[t] = floatx4{-m_used[t], -m_used[t], -m_used[t], -m_used[t]};
                lrow[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DV; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha;
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[t][f][r] -= delta;
            }
            floatx2 ps = {0.f, 0.f};
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const floatx2 e = {__builtin_amdgcn_exp2f(s[t][f][r]), __builtin_amdgcn_exp2f(s[t][f][r + 1])};
                    if (!ONES) ps += e;
                    const half2v eh = __builtin_convertvector(e, half2v);
                    p[t][f >> 1][(f & 1) * 4 + r] = eh[0];
                    p[t][f >> 1][(f & 1) * 4 + r + 1] = eh[1];
                }
            if (!ONES) lrow[t] += ps[0] + ps[1];
        }
#if ATT_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
                const half8 vf = *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR +
                                                                 (kg * 4 + g) * 16);
                o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[0][kg], o[0][dt], 0, 0, 0);
                o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[1][kg], o[1][dt], 0, 0, 0);
            }
#if ATT_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (j + 1 < ntiles) ATT_STORE((j + 1) & 1);
        __syncthreads();
    }
#undef ATT_LOAD
#undef ATT_STORE

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float l;
        if (ONES) {
            // denominator = O^T row d: lane group g = (d>>2)&3, fragment d>>4, element 0
            float lv = 0.f;
#pragma unroll
            for (int dt = 0; dt < DV; ++dt)
                if (dt == (d >> 4)) lv = o[t][dt][0];
            l = __shfl(lv, ((d >> 2) & 3) * 16 + fr, 64);
        } else {
            l = lrow[t];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
        }
        const float inv = 1.0f / l;
        const int q = q0 + t * 16 + fr;
        if (q < a.Nq) {
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
                const int d0 = dt * 16 + g * 4;
                if (d0 >= d) continue;
                half4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[t][dt][r] * inv);
                *reinterpret_cast<half4*>(Ob + (size_t)q * a.ldo + d0) = v;
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------
// head_dim 40, Q pre-scaled: k_attention_w8q2 with the QK^T product on v_mfma_f32_32x32x16_f16.
// The 16x16x32 form contracts 32 head-dim columns per instruction, so head_dim 40 costs 64
// columns (37.5 % zeros); the 32x32x16 form contracts 16, so 40 costs 48 -- 6 MFMAs of 32
// passes per 64-key tile and 32 queries instead of 16 of 16 passes (-25 % matrix-pipe time on
// QK^T, -14 % on the whole tile loop's MFMA work).  What changes around it:
//  * a wave's 32 queries are ONE B operand (lane l: query l & 31, head-dim chunk 2 ks + (l >> 5));
//    S^T comes out with query l & 31 on lane l and 32 of its 64 keys (the other 32 on lane l ^ 32);
//  * the running max rides through the MFMA as a head-dim column instead of the accumulator
//    input (a 32x32 accumulator input would be 16 more VGPRs): K's first padding column (40) is
//    1.0 in LDS, the same column of Q holds -m.  m is kept fp16-representable so that the shift
//    applied is exactly the shift tracked (softmax is invariant to the shift, not to a mismatch);
//  * P for the PV product (still 16x16x32: O^T has 48 rows, a 32-row form would pad to 64):
//    packed-fp16 P registers of query rows 16..31 and 0..15 are exchanged between lane rows with
//    v_permlane16_swap (8 per tile), which turns the 32x32 accumulator layout into two 16-query B
//    operands; the key order inside each 32-key group that this produces is baked into the LDS
//    image of V^T (k slot (g, e) <-> key 8 ((g & 1) + 2 (e >> 2)) + 4 (g >> 1) + (e & 3)).
// Same lazy max, same fused denominator (ones row of V^T), same staging as k_attention_w8q2.
__global__ __launch_bounds__(512, 4) void k_attention_w8q2m(AttnArgs a) {   // (HIP: the second bound is waves per SIMD: <= 128 VGPRs, two workgroups per CU)
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 512;
    constexpr int DV = 3;
    constexpr int KSTR = 7 * 16;        // 6 chunks of 8 head-dim columns + 16 B pad: 16 rows x 112 B tile the 64 banks
    constexpr int VSTR = 10 * 16;
    constexpr int VROWS = DV * 16;
    constexpr int KBYTES = 64 * KSTR, VBYTES = VROWS * VSTR;
    constexpr unsigned OOB = 0x7fffffffu;
    __shared__ __attribute__((aligned(16))) char sKV[2 * (KBYTES + VBYTES)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;      // PV / O^T domain
    const int qi = lane & 31, hb = lane >> 5;     // QK^T / S^T domain
    const int nqb = (a.Nq + 255) >> 8;
    const int nwg = gridDim.x;
    int id = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int qb = id % nqb, h = (id / nqb) % a.heads, b = id / (nqb * a.heads);
    const int qblk0 = qb * 256;
    const int q0 = qblk0 + wave * 32;
    constexpr int d = 40;
    const half_t* __restrict__ Qb = a.Q + (size_t)b * a.sQ + h * d;
    const half_t* Kb = a.K + (size_t)b * a.sK + h * d;
    const half_t* Vb = a.Vt + (size_t)b * a.sVt + (size_t)h * d * a.ldvt;
    half_t* __restrict__ Ob = a.O + (size_t)b * a.sO + h * d;
    const int nk8 = (a.Nk + 7) & ~7;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Kb, 0, (unsigned)(((size_t)(a.Nk - 1) * a.ldk + d) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Vb, 0, (unsigned)(((size_t)(d - 1) * a.ldvt + nk8) * 2), 0x00020000);

    half8 qf[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        const int q = q0 + qi, d0 = (ks * 2 + hb) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q < a.Nq && d0 < d) v = *reinterpret_cast<const u32x4*>(Qb + (size_t)q * a.ldq + d0);
        qf[ks] = __builtin_bit_cast(half8, v);
    }

    floatx4 o[2][DV];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < DV; ++dt) o[t][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_used = 0.f;   // running (lazy) max of query qi, base-2 logit units, always an fp16 value

    int ntiles = (a.Nk + 63) >> 6;
    if (a.causal) {
        const int qend = min(a.Nq, qblk0 + 256);
        ntiles = min(ntiles, (qend + 63) >> 6);
    }

    // K: 64 rows x 5 chunks of 16 B (threads 0..319); V^T: 40 live rows x 8 chunks (threads 0..319), rows 40..47 constant
    const bool kld = tid < 64 * 5;
    const int krow = tid / 5, kc = tid - krow * 5;
    const unsigned kvo = kld ? (unsigned)(krow * a.ldk + kc * 8) * 2u : OOB;
    const unsigned kst = (unsigned)(krow * KSTR + kc * 16);
    const int vrow = tid >> 3, vc = tid & 7;
    const bool vlive = tid < VROWS * 8 && vrow < d;
    const unsigned vvo = vlive ? (unsigned)(vrow * a.ldvt + vc * 8) * 2u : OOB;
    // 8 keys of chunk vc = keys 8 cc + {0..3} (k slot row g = cc & 1) and 8 cc + {4..7} (g = 2 + (cc & 1)) of 32-key group vc >> 2
    const unsigned vst = (unsigned)(KBYTES + vrow * VSTR + ((vc >> 2) * 32 + (vc & 1) * 8 + ((vc >> 1) & 1) * 4) * 2);

    u32x4 rk, rv;
#define ATT_LOAD(J)                                                                           \
    {                                                                                         \
        const int key0 = (J) * 64;                                                            \
        if (key0 + 64 <= a.Nk) {                                                              \
            rk = __builtin_amdgcn_raw_buffer_load_b128(rsK, kvo, key0 * a.ldk * 2, 0);        \
            rv = __builtin_amdgcn_raw_buffer_load_b128(rsV, vvo, key0 * 2, 0);                \
        } else { /* ragged last tile: rows / key chunks past the end read zero */             \
            const unsigned ko = (kld && key0 + krow < a.Nk) ? kvo + (unsigned)key0 * a.ldk * 2u : OOB; \
            rk = __builtin_amdgcn_raw_buffer_load_b128(rsK, ko, 0, 0);                        \
            const unsigned vo = (key0 + vc * 8 < nk8) ? vvo + (unsigned)key0 * 2u : OOB;      \
            rv = __builtin_amdgcn_raw_buffer_load_b128(rsV, vo, 0, 0);                        \
        }                                                                                     \
    }
#define ATT_STORE(BUF)                                                                        \
    {                                                                                         \
        char* sb = sKV + (BUF) * (KBYTES + VBYTES);                                           \
        if (kld) *reinterpret_cast<u32x4*>(sb + kst) = rk;                                    \
        if (vlive) {                                                                          \
            *reinterpret_cast<u32x2*>(sb + vst) = u32x2{rv[0], rv[1]};                        \
            *reinterpret_cast<u32x2*>(sb + vst + 32) = u32x2{rv[2], rv[3]};                   \
        }                                                                                     \
    }

    // constants of both stages: K column 40 = 1.0 (carries -m through QK^T), columns 41..47 = 0;
    // V^T row 40 = 1.0 (the PV product accumulates the softmax denominator), rows 41..47 = 0
    if (tid < 128) {
        char* sb = sKV + (tid >> 6) * (KBYTES + VBYTES);
        *reinterpret_cast<u32x4*>(sb + (tid & 63) * KSTR + 5 * 16) = u32x4{0x00003C00u, 0u, 0u, 0u};
    }
    if (tid < VROWS * 8 && !vlive) {
        const unsigned w = vrow == d ? 0x3C003C00u : 0u;
#pragma unroll
        for (int buf = 0; buf < 2; ++buf) {
            char* sb = sKV + buf * (KBYTES + VBYTES);
            *reinterpret_cast<u32x2*>(sb + vst) = u32x2{w, w};
            *reinterpret_cast<u32x2*>(sb + vst + 32) = u32x2{w, w};
        }
    }
    ATT_LOAD(0);
    ATT_STORE(0);
    __syncthreads();
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) ATT_LOAD(j + 1);
        const char* sK = sKV + (j & 1) * (KBYTES + VBYTES);
        const char* sV = sK + KBYTES;
        const bool need_mask = (j * 64 + 64 > a.Nk) || a.causal;
        floatx16 s[2];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const half8 kf = *reinterpret_cast<const half8*>(sK + (kh * 32 + qi) * KSTR + (ks * 2 + hb) * 16);
                s[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? zero16 : s[kh], 0, 0, 0);
            }
        if (need_mask) {
            const int q = q0 + qi;
            int kb = j * 64 + hb * 4;
            asm volatile("" : "+v"(kb));
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = kb + kh * 32 + (i >> 2) * 8 + (i & 3);
                    if (key >= a.Nk || (a.causal && key > q)) s[kh][i] = -INFINITY;
                }
        }
        // P = 2^s as packed fp16; R[kh][c][w]: keys kh*32 + 8c + 4hb + 2w + {0, 1} of query qi.  Exponentiated BEFORE the
        // lazy-max vote: the vote is then a max over 16 packed registers (v_pk_maximum3_f16: four values per instruction,
        // against two for v_max3_f32 on the fp32 logits), P > 2^LAZY_THR <=> s > LAZY_THR; the rare path redoes the tile's P.
        unsigned R[2][4][2];
#define ATT_EXP()                                                                              \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                           \
        _Pragma("unroll") for (int c = 0; c < 4; ++c)                                          \
            _Pragma("unroll") for (int w = 0; w < 2; ++w) {                                    \
                const floatx2 e = {__builtin_amdgcn_exp2f(s[kh][4 * c + 2 * w]), __builtin_amdgcn_exp2f(s[kh][4 * c + 2 * w + 1])}; \
                R[kh][c][w] = __builtin_bit_cast(unsigned, __builtin_convertvector(e, half2v)); \
            }
        bool over = true;      // first tile: the running max is set from its row max
        if (j != 0) {
            ATT_EXP();
            half2v pm = __builtin_bit_cast(half2v, R[0][0][0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) pm = __builtin_elementwise_maximum(pm, __builtin_bit_cast(half2v, R[i >> 3][(i >> 1) & 3][i & 1]));
            pm = __builtin_elementwise_maximum(pm, half2v{pm[1], pm[0]});
            // P >= 0 (or +inf / NaN): fp16 bit patterns order as integers; 0x5C00 = 2^8 = 2^LAZY_THR
            static_assert(LAZY_THR == 8.0f, "the packed vote compares against fp16 2^8");
            over = __builtin_bit_cast(unsigned, pm) > 0x5C005C00u;
        }
        if (__any(over)) {
            // advance the running max to the exact row max (rare after the first tiles)
            float tmax;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]));
#pragma unroll
            for (int i = 3; i < 15; i += 2)
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[0][i]), "v"(s[0][i + 1]));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[0][15]), "v"(s[1][0]));
#pragma unroll
            for (int i = 1; i < 15; i += 2)
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(s[1][i]), "v"(s[1][i + 1]));
            asm("v_max_f32 %0, %1, %2" : "=v"(tmax) : "v"(tmax), "v"(s[1][15]));
            const unsigned tu = __float_as_uint(tmax);
            auto r32 = __builtin_amdgcn_permlane32_swap(tu, tu, false, false);
            const float tm = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
            float delta = (j == 0) ? tm : fmaxf(tm, 0.f);
            if (delta == -INFINITY) delta = 0.f;
            const float m_new = (float)(half_t)fminf(fmaxf(m_used + delta, -60000.f), 60000.f);
            delta = m_new - m_used;      // the shift actually applied from the next tile on
            const float alpha = (j == 0) ? 1.f : __builtin_amdgcn_exp2f(-delta);
            m_used = m_new;
            if (hb) qf[2][0] = (half_t)(-m_used);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float at = __shfl(alpha, t * 16 + fr, 64);
#pragma unroll
                for (int dt = 0; dt < DV; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[t][dt][r] *= at;
            }
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[kh][i] -= delta;
            ATT_EXP();
        }
#undef ATT_EXP
        // lane rows 1, 3 (queries 16..31) of the even chunk <-> lane rows 0, 2 (queries 0..15) of the odd chunk
        half8 p[2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int c = 0; c < 4; c += 2)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    auto r = __builtin_amdgcn_permlane16_swap(R[kh][c][w], R[kh][c + 1][w], false, false);
                    R[kh][c][w] = r[0];
                    R[kh][c + 1][w] = r[1];
                }
            p[0][kh] = __builtin_bit_cast(half8, u32x4{R[kh][0][0], R[kh][0][1], R[kh][2][0], R[kh][2][1]});
            p[1][kh] = __builtin_bit_cast(half8, u32x4{R[kh][1][0], R[kh][1][1], R[kh][3][0], R[kh][3][1]});
        }
#pragma unroll
        for (int kg = 0; kg < 2; ++kg)
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
                const half8 vf = *reinterpret_cast<const half8*>(sV + (dt * 16 + fr) * VSTR + (kg * 4 + g) * 16);
                o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[0][kg], o[0][dt], 0, 0, 0);
                o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, p[1][kg], o[1][dt], 0, 0, 0);
            }
        if (j + 1 < ntiles) ATT_STORE((j + 1) & 1);
        __syncthreads();
    }
#undef ATT_LOAD
#undef ATT_STORE

#pragma unroll
    for (int t = 0; t < 2; ++t) {
        // denominator = O^T row 40: fragment 2, lane group g = 2, element 0
        const float l = __shfl(o[t][2][0], 2 * 16 + fr, 64);
        const float inv = 1.0f / l;
        const int q = q0 + t * 16 + fr;
        if (q < a.Nq) {
#pragma unroll
            for (int dt = 0; dt < DV; ++dt) {
                const int d0 = dt * 16 + g * 4;
                if (d0 >= d) continue;
                half4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (half_t)(o[t][dt][r] * inv);
                *reinterpret_cast<half4*>(Ob + (size_t)q * a.ldo + d0) = v;
            }
        }
    }
#endif
}

extern "C" int fd_attention_f16(const fd_attention_desc* d, void* stream) {
    if (fd_plan_recording() && d) {
        const fd_attention_desc dc_ = *d;
        fd_plan_push([dc_](void* fd_s_) -> int { return fd_attention_f16(&dc_, fd_s_); });
    }
    FD_CHECK_ARG(d && d->Q && d->K && d->Vt && d->O, FD_EINVAL, "fd_attention_f16: null pointer");
    FD_CHECK_ARG(d->batch > 0 && d->heads > 0 && d->n_q > 0 && d->n_k > 0, FD_EINVAL,
                 "fd_attention_f16: non-positive dimension");
    FD_CHECK_ARG(d->head_dim % 8 == 0 && d->head_dim <= 160 && d->head_dim >= 8, FD_ESHAPE,
                 "fd_attention_f16: head_dim=%d unsupported (multiple of 8, <= 160)", d->head_dim);
    FD_CHECK_ARG(d->ldq % 8 == 0 && d->ldk % 8 == 0 && d->ldvt % 8 == 0 && d->ldo % 4 == 0,
                 FD_ESHAPE, "fd_attention_f16: leading dimensions must be multiples of 8");
    FD_CHECK_ARG(d->ldvt >= ((d->n_k + 7) & ~7), FD_ESHAPE,
                 "fd_attention_f16: ldvt=%d < n_k rounded up to 8", d->ldvt);
    AttnArgs a;
    a.Q = (const half_t*)d->Q; a.K = (const half_t*)d->K; a.Vt = (const half_t*)d->Vt;
    a.O = (half_t*)d->O;
    a.sQ = d->q_sample_stride; a.sK = d->k_sample_stride; a.sVt = d->vt_sample_stride;
    a.sO = d->o_sample_stride;
    a.ldq = d->ldq; a.ldk = d->ldk; a.ldvt = d->ldvt; a.ldo = d->ldo;
    a.Nq = d->n_q; a.Nk = d->n_k; a.d = d->head_dim; a.heads = d->heads;
    a.causal = d->causal;
    const float scale = d->scale > 0.f ? d->scale : 1.0f / sqrtf((float)d->head_dim);
    a.scale_log2 = scale * 1.4426950408889634f;
    if (d->q_prescaled) a.scale_log2 = 1.0f;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(fd_cdiv(d->n_q, 128) * d->heads * d->batch);
    const double flops = 4.0 * (double)d->batch * d->heads * (double)d->n_q * d->n_k * d->head_dim *
                         (d->causal ? 0.5 : 1.0);
    fd_prof_begin(FD_FAMILY_ATTENTION, st, flops, -1.0, fd_tag(6u, d->batch * d->heads, d->n_q, d->n_k, d->head_dim));
    const int hd = d->head_dim;
    // FD_ATTN_QT1: 0 = 4-wave / 32-query kernel, 1 = 8-wave kernel with the exact running max,
    // 2 (default) = 8-wave kernel with the VALU-lean softmax (lazy max, fused denominator)
    static const int wide = getenv("FD_ATTN_QT1") ? atoi(getenv("FD_ATTN_QT1")) : 2;
    const bool pre = d->q_prescaled != 0;
    // two query blocks per wave for the long self-attention rows (FD_ATTN_Q2=0: one block, A/B)
    static const int q2 = getenv("FD_ATTN_Q2") ? atoi(getenv("FD_ATTN_Q2")) : 1;
    FD_CHECK_ARG(!pre || (wide == 2 && (hd <= 80 || hd > 128)), FD_ESHAPE,
                 "fd_attention_f16: q_prescaled is not supported for head_dim=%d", hd);
#define ATT_W8(DQK, DV, ONES)                                                                  \
    {                                                                                          \
        if (pre) hipLaunchKernelGGL((k_attention_w8<DQK, DV, true, ONES>), grid, dim3(512), 0, st, a); \
        else hipLaunchKernelGGL((k_attention_w8<DQK, DV, false, ONES>), grid, dim3(512), 0, st, a);    \
    }
    // also on the short text-context rows (n_k = 77): K / V^T staging per query halves (38.6 -> 36.9 us at 16x4096 queries)
    const int q2_mink = 64;
    // head dims 49..64 (SD2.1: 64): the same two-block kernel with four O^T fragments (FD_ATTN_Q2_64=0: one block per wave, A/B)
    static const int q2_64 = getenv("FD_ATTN_Q2_64") ? atoi(getenv("FD_ATTN_Q2_64")) : 1;
    if (hd > 48 && hd <= 64 && wide == 2 && q2 && q2_64 && d->n_q >= 2048 && d->n_k >= 1024) {
        dim3 grid2(fd_cdiv(d->n_q, 256) * d->heads * d->batch);
        if (pre) hipLaunchKernelGGL((k_attention_w8q2<64, 4, true, false>), grid2, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention_w8q2<64, 4, false, false>), grid2, dim3(512), 0, st, a);
    } else if (hd <= 48 && wide == 2 && q2 && d->n_q >= 2048 && d->n_k >= q2_mink) {
        dim3 grid2(fd_cdiv(d->n_q, 256) * d->heads * d->batch);
        // FD_ATTN_M32=0: the 16x16x32 QK^T form for head_dim 40 too (A/B)
        static const int m32 = getenv("FD_ATTN_M32") ? atoi(getenv("FD_ATTN_M32")) : 1;
        if (hd == 40 && pre && m32) {
            hipLaunchKernelGGL(k_attention_w8q2m, grid2, dim3(512), 0, st, a);
        } else if (hd <= 40) {
            if (pre) hipLaunchKernelGGL((k_attention_w8q2<64, 3, true, true>), grid2, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_attention_w8q2<64, 3, false, true>), grid2, dim3(512), 0, st, a);
        } else {
            if (pre) hipLaunchKernelGGL((k_attention_w8q2<64, 3, true, false>), grid2, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_attention_w8q2<64, 3, false, false>), grid2, dim3(512), 0, st, a);
        }
    } else if (hd <= 48) {
        if (wide == 2 && hd <= 40) ATT_W8(64, 3, true)
        else if (wide == 2) ATT_W8(64, 3, false)
        else if (wide) hipLaunchKernelGGL((k_attention<64, 3, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<64, 3>), grid, dim3(256), 0, st, a);
    } else if (hd <= 64) {
        if (wide == 2) ATT_W8(64, 4, false)
        else if (wide) hipLaunchKernelGGL((k_attention<64, 4, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<64, 4>), grid, dim3(256), 0, st, a);
    } else if (hd <= 80) {
        if (wide == 2) ATT_W8(96, 5, false)
        else if (wide) hipLaunchKernelGGL((k_attention<96, 5, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<96, 5>), grid, dim3(256), 0, st, a);
    } else if (hd <= 96) {
        hipLaunchKernelGGL((k_attention<96, 6>), grid, dim3(256), 0, st, a);
    } else if (hd <= 128) {
        hipLaunchKernelGGL((k_attention<128, 8>), grid, dim3(256), 0, st, a);
    } else {
        if (wide == 2) ATT_W8(160, 10, false)
        else if (wide) hipLaunchKernelGGL((k_attention<160, 10, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<160, 10>), grid, dim3(256), 0, st, a);
    }
#undef ATT_W8
    fd_prof_end(FD_FAMILY_ATTENTION, st);
    FD_CHECK_LAUNCH("k_attention");
    return FD_OK;
}
