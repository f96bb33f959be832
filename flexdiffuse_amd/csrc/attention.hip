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
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = j * 64 + f * 16 + g * 4 + r;
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

extern "C" int fd_attention_f16(const fd_attention_desc* d, void* stream) {
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
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(fd_cdiv(d->n_q, 128) * d->heads * d->batch);
    const double flops = 4.0 * (double)d->batch * d->heads * (double)d->n_q * d->n_k * d->head_dim *
                         (d->causal ? 0.5 : 1.0);
    fd_prof_begin(FD_FAMILY_ATTENTION, st, flops);
    const int hd = d->head_dim;
    static const int wide = getenv("FD_ATTN_QT1") ? atoi(getenv("FD_ATTN_QT1")) : 1;  // measured: -5...-20 %
    if (hd <= 48) {
        if (wide) hipLaunchKernelGGL((k_attention<64, 3, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<64, 3>), grid, dim3(256), 0, st, a);
    } else if (hd <= 64) {
        if (wide) hipLaunchKernelGGL((k_attention<64, 4, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<64, 4>), grid, dim3(256), 0, st, a);
    } else if (hd <= 80) {
        if (wide) hipLaunchKernelGGL((k_attention<96, 5, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<96, 5>), grid, dim3(256), 0, st, a);
    } else if (hd <= 96) {
        hipLaunchKernelGGL((k_attention<96, 6>), grid, dim3(256), 0, st, a);
    } else if (hd <= 128) {
        hipLaunchKernelGGL((k_attention<128, 8>), grid, dim3(256), 0, st, a);
    } else {
        if (wide) hipLaunchKernelGGL((k_attention<160, 10, 1, 8>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((k_attention<160, 10>), grid, dim3(256), 0, st, a);
    }
    fd_prof_end(FD_FAMILY_ATTENTION, st);
    FD_CHECK_LAUNCH("k_attention");
    return FD_OK;
}
