// Guidance stage on gfx950: CLIP image<->text token alignment ("map") and the
// Linear / Clustered / Threshold tween.  Replaces the Python/.item() loops of the
// reference guidance.py:23-272 with two launches per batch of prompts:
//
//   k_guidance_sim   grid (ceil(N/32), B): fp32 cosine logits on the f32-input MFMA
//                    (v_mfma_f32_32x32x2_f32: exact fp32 fma chain), K split over the
//                    4 waves of the workgroup, fixed-order combine in LDS, 77-way
//                    softmax with wave shuffles -> sim[B][N][L] (fp32, L2-resident).
//   k_guidance_tween grid (B): the similarity table of one prompt staged in LDS
//                    (257x77 f32 = 79 KB of the CU's 160 KB), greedy assignment,
//                    weights (one lane, float64 compares like the reference's Python
//                    floats), then a float4-vectorised blend of the (L, D) tokens.
//
// Bit-level notes: decisions are taken on float64 images of fp32
// similarities; weight updates are fp32 op fp32(scalar); the blend is
// base + (alt - base) * fp32(iw) with separately rounded sub/mul/add (no FMA).
#include "common.h"

#define SIM_CT 3  // column tiles of 32 -> L <= 96

__global__ __launch_bounds__(256) void k_guidance_sim(const float* __restrict__ alt,
                                                      const float* __restrict__ txt,
                                                      float* __restrict__ sim, int alt_batched,
                                                      int N, int L, int D) {
    __shared__ float norms[32 + SIM_CT * 32];
    __shared__ float part[4][32][SIM_CT * 32];
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* altb = alt + (alt_batched ? (size_t)b * N * D : 0);
    const float* txtb = txt + (size_t)b * L * D;

    // ---- L2 norms of the 32 guide rows of this tile and of all text rows ----------
    for (int q = wave; q < 32 + SIM_CT * 32; q += 4) {
        const float* row = nullptr;
        if (q < 32) {
            if (r0 + q < N) row = altb + (size_t)(r0 + q) * D;
        } else if (q - 32 < L) {
            row = txtb + (size_t)(q - 32) * D;
        }
        float acc = 0.f;
        if (row) {
            for (int k = lane * 4; k < D; k += 256) {
                const float4 v = *reinterpret_cast<const float4*>(row + k);
                acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        acc = fd_wave_sum(acc);
        if (lane == 0) norms[q] = row ? sqrtf(acc) : 1.0f;
    }
    __syncthreads();

    // ---- logits tile: 32 guide rows x 96 text cols, K slice [wave*D/4, (wave+1)*D/4) --
    floatx16 acc[SIM_CT];
#pragma unroll
    for (int c = 0; c < SIM_CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    const int dq = D >> 2;
    const bool a_ok = (r0 + r) < N;
    const float* arow = altb + (size_t)(a_ok ? r0 + r : 0) * D;
    const float an = norms[r];
    const float* brow[SIM_CT];
    float bn[SIM_CT];
    bool b_ok[SIM_CT];
#pragma unroll
    for (int c = 0; c < SIM_CT; ++c) {
        const int j = c * 32 + r;
        b_ok[c] = j < L;
        brow[c] = txtb + (size_t)(b_ok[c] ? j : 0) * D;
        bn[c] = norms[32 + j];
    }
    for (int k0 = wave * dq; k0 < (wave + 1) * dq; k0 += 8) {
        const int k = k0 + h * 4;
        float4 av = *reinterpret_cast<const float4*>(arow + k);
        if (!a_ok) av = make_float4(0.f, 0.f, 0.f, 0.f);
        const float a4[4] = {av.x / an, av.y / an, av.z / an, av.w / an};
        float b4[SIM_CT][4];
#pragma unroll
        for (int c = 0; c < SIM_CT; ++c) {
            float4 bv = *reinterpret_cast<const float4*>(brow[c] + k);
            if (!b_ok[c]) bv = make_float4(0.f, 0.f, 0.f, 0.f);
            b4[c][0] = bv.x / bn[c];
            b4[c][1] = bv.y / bn[c];
            b4[c][2] = bv.z / bn[c];
            b4[c][3] = bv.w / bn[c];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int c = 0; c < SIM_CT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[c][s], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < SIM_CT; ++c)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            part[wave][row][c * 32 + r] = acc[c][q];
        }
    __syncthreads();

    // ---- fixed-order combine, x100, softmax over the L text tokens -----------------
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr;
        if (r0 + row >= N) break;
        float x[2];
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < L) {
                const float d = ((part[0][row][j] + part[1][row][j]) + part[2][row][j]) +
                                part[3][row][j];
                x[u] = 100.0f * d;
                m = fmaxf(m, x[u]);
            } else {
                x[u] = -INFINITY;
            }
        }
        m = fd_wave_max(m);
        float e[2], sum = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            e[u] = (lane + 64 * u < L) ? expf(x[u] - m) : 0.f;
            sum += e[u];
        }
        sum = fd_wave_sum(sum);
        float* dst = sim + ((size_t)b * N + r0 + row) * L;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = lane + 64 * u;
            if (j < L) dst[j] = e[u] / sum;
        }
    }
}

// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long key_of(float v, unsigned lo) {
    return ((unsigned long long)(__float_as_uint(v) + 1u) << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o, 64);
        k = other > k ? other : k;
    }
    return k;
}

// a.max() >= 0 ? (b.max() >= 0 ? max(a,b) : a+b) : min(a,b)     (guidance.py:175-193)
__device__ void blend_weights_dev(float* a, const float* b, int n) {
    float amax = a[0], bmax = b[0];
    for (int k = 1; k < n; ++k) {
        amax = fmaxf(amax, a[k]);
        bmax = fmaxf(bmax, b[k]);
    }
    if (amax >= 0.f) {
        if (bmax >= 0.f) {
            for (int k = 0; k < n; ++k) a[k] = fmaxf(a[k], b[k]);
        } else {
            for (int k = 0; k < n; ++k) a[k] = __fadd_rn(a[k], b[k]);
        }
    } else {
        for (int k = 0; k < n; ++k) a[k] = fminf(a[k], b[k]);
    }
}

#define TW_MAXL 96

struct TweenShared {
    int idx[TW_MAXL];
    float sv[TW_MAXL];
    float w[TW_MAXL];
    float tmp[TW_MAXL];
    int peaks[TW_MAXL];
    int valleys[TW_MAXL + 2];
    int mode[TW_MAXL];
    float iwf[TW_MAXL];
    unsigned long long red[4];
    int status;
};

__global__ __launch_bounds__(256) void k_guidance_tween(
    const float* __restrict__ sim, const float* __restrict__ base, const float* __restrict__ alt,
    const float* __restrict__ lin_w, float* __restrict__ out, float* __restrict__ weights,
    int* __restrict__ idx_out, float* __restrict__ s_out, int* __restrict__ status_out,
    int alt_batched, int N, int L, int D, fd_tween_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TweenShared& sh = *reinterpret_cast<TweenShared*>(smem_raw);
    float* S = reinterpret_cast<float*>(smem_raw + ((sizeof(TweenShared) + 15) & ~15));
    const int LS = L | 1;  // odd row stride: conflict-free row- and column-wise scans
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NC = L - 1;  // similarity columns 1..L-1 (header column dropped)

    const float* simb = sim + (size_t)b * N * L;
    for (int e = tid; e < N * L; e += 256) {
        const int i = e / L, j = e - i * L;
        S[i * LS + j] = simb[e];
    }
    if (tid < TW_MAXL) {
        sh.idx[tid] = 0;
        sh.sv[tid] = 0.f;
    }
    if (tid == 0) sh.status = 0;
    __syncthreads();

    // ---------------- assignment (guidance.py:57-85) -------------------------------
    if (p.order == FD_ORDER_DIRECT) {
        if (tid < NC && tid < N) {
            sh.idx[tid] = tid;
            sh.sv[tid] = S[tid * LS + tid + 1];
        }
    } else if (p.reuse) {
        // best guide token per text column; ties -> lowest index.  A column whose best
        // similarity underflowed to 0 never "locks": the last candidate (N-1) stays.
        if (tid < NC) {
            float best = -1.f;
            int bi = 0;
            for (int i = 0; i < N; ++i) {
                const float v = S[i * LS + tid + 1];
                if (v > best) {
                    best = v;
                    bi = i;
                }
            }
            if (!(best > 0.f)) {
                best = 0.f;
                bi = N - 1;
            }
            sh.idx[tid] = bi;
            sh.sv[tid] = best;
        }
    } else if (p.order == FD_ORDER_TEXT) {
        // text order, each guide token used once: one wave walks the columns
        if (wave == 0) {
            unsigned used = 0;  // bit m <-> row lane + 64*m   (N <= 2048)
            for (int j = 0; j < NC; ++j) {
                unsigned long long k = 0;
                for (int i = lane, m = 0; i < N; i += 64, ++m)
                    if (!((used >> m) & 1u)) {
                        const unsigned long long c = key_of(S[i * LS + j + 1], 0xFFFFFFFFu - i);
                        k = c > k ? c : k;
                    }
                k = wave_max_u64(k);
                if (k == 0) break;  // nothing unused
                const float v = __uint_as_float((unsigned)(k >> 32) - 1u);
                if (v > 0.f) {
                    const int i = (int)(0xFFFFFFFFu - (unsigned)k);
                    if (lane == (i & 63)) used |= 1u << (i >> 6);
                    if (lane == 0) {
                        sh.idx[j] = i;
                        sh.sv[j] = v;
                    }
                } else {
                    // only zero-similarity candidates left: they never lock the slot and
                    // each one consumes its guide token -> the slot ends on the highest
                    // unused index and nothing is left for later columns
                    int hi = -1;
                    for (int i = lane, m = 0; i < N; i += 64, ++m)
                        if (!((used >> m) & 1u)) hi = i;
                    for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_xor(hi, o, 64));
                    if (lane == 0) {
                        sh.idx[j] = hi;
                        sh.sv[j] = 0.f;
                    }
                    break;
                }
            }
        }
    } else {
        // alignment order, each guide token used once: repeated global arg-max over the
        // unlocked columns x unused rows, ties -> lowest column then lowest row.
        unsigned lock[3] = {0u, 0u, 0u};
        const int RM = (N + 255) >> 8;  // rows per thread (<= 8)
        unsigned usedm = 0;
        unsigned long long cache[8];
        bool stale[8];
        for (int m = 0; m < 8; ++m) {
            cache[m] = 0;
            stale[m] = true;
        }
        for (int it = 0; it < NC; ++it) {
            unsigned long long k = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int i = tid + 256 * m;
                if (m >= RM || i >= N || ((usedm >> m) & 1u)) continue;
                if (stale[m]) {
                    unsigned long long c = 0;
                    for (int j = 0; j < NC; ++j) {
                        if ((lock[j >> 5] >> (j & 31)) & 1u) continue;
                        const unsigned long long cj =
                            key_of(S[i * LS + j + 1],
                                   ((unsigned)(0xFFFF - j) << 16) | (unsigned)(0xFFFF - i));
                        c = cj > c ? cj : c;
                    }
                    cache[m] = c;
                    stale[m] = false;
                }
                k = cache[m] > k ? cache[m] : k;
            }
            k = wave_max_u64(k);
            if (lane == 0) sh.red[wave] = k;
            __syncthreads();
            unsigned long long g = sh.red[0];
            g = sh.red[1] > g ? sh.red[1] : g;
            g = sh.red[2] > g ? sh.red[2] : g;
            g = sh.red[3] > g ? sh.red[3] : g;
            __syncthreads();
            if (g == 0) break;
            const float v = __uint_as_float((unsigned)(g >> 32) - 1u);
            const int j = 0xFFFF - (int)(((unsigned)g >> 16) & 0xFFFFu);
            const int i = 0xFFFF - (int)((unsigned)g & 0xFFFFu);
            if (v > 0.f) {
                if (tid == 0) {
                    sh.idx[j] = i;
                    sh.sv[j] = v;
                }
                lock[j >> 5] |= 1u << (j & 31);
                if ((i & 255) == tid) usedm |= 1u << (i >> 8);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int cj = 0xFFFF - (int)(((unsigned)cache[m] >> 16) & 0xFFFFu);
                    if (cache[m] != 0 && cj == j) stale[m] = true;
                }
            } else {
                // zero-similarity tail: first unlocked column swallows every unused row
                int hi = -1;
                for (int m = 0; m < RM; ++m) {
                    const int r = tid + 256 * m;
                    if (r < N && !((usedm >> m) & 1u)) hi = r;
                }
                for (int o = 32; o > 0; o >>= 1) hi = max(hi, __shfl_xor(hi, o, 64));
                if (lane == 0) sh.red[wave] = (unsigned long long)(hi + 1);
                __syncthreads();
                if (tid == 0) {
                    unsigned long long mx = sh.red[0];
                    for (int q = 1; q < 4; ++q) mx = sh.red[q] > mx ? sh.red[q] : mx;
                    int j0 = 0;
                    while (j0 < NC && ((lock[j0 >> 5] >> (j0 & 31)) & 1u)) ++j0;
                    if (j0 < NC && mx > 0) {
                        sh.idx[j0] = (int)mx - 1;
                        sh.sv[j0] = 0.f;
                    }
                }
                break;
            }
        }
    }
    __syncthreads();
    if (tid < L) {
        idx_out[(size_t)b * L + tid] = sh.idx[tid];
        s_out[(size_t)b * L + tid] = sh.sv[tid];
    }
    if (base == nullptr) return;  // map only

    // ---------------- weights (guidance.py:219-254), one lane ------------------------
    if (tid == 0) {
        double mean = 0.0;
        for (int k = 0; k < L; ++k) mean += (double)sh.sv[k];
        mean /= (double)L;
        for (int k = 0; k < L; ++k) sh.w[k] = lin_w[k];
        if (p.clustered != 0.0) {
            int np = 0;
            for (int k = 1; k < L - 1; ++k) {
                const double s = (double)sh.sv[k];
                if (s < mean) continue;
                if ((double)sh.sv[k - 1] <= s && s >= (double)sh.sv[k + 1]) sh.peaks[np++] = k;
            }
            bool plateau = false;
            for (int q = 0; q + 1 < np; ++q)
                if (sh.peaks[q + 1] == sh.peaks[q] + 1) plateau = true;
            if (plateau) {
                sh.status = 1;  // reference: ZeroDivisionError (guidance.py:112)
            } else if (np > 0) {
                int nv = 0;
                sh.valleys[nv++] = 0;
                for (int q = 0; q + 1 < np; ++q) {
                    const int d = sh.peaks[q + 1] - sh.peaks[q];
                    sh.valleys[nv++] = sh.peaks[q] + (d + 1) / 2;
                }
                sh.valleys[nv++] = L - 1;
                float* cw = sh.tmp;
                for (int k = 0; k < L; ++k) cw[k] = 1.0f;
                cw[0] = __fsub_rn(cw[0], 1.0f);
                int vi = 0;
                for (int q = 0; q < np; ++q) {
                    const int a = sh.peaks[q];
                    int v = sh.valleys[vi];
                    if (v < a) {
                        const double gs = 1.0 / (double)(a - v);
                        for (int k = 1; k < a - v; ++k)
                            cw[a - k] = __fsub_rn(cw[a - k], (float)(gs * (double)k));
                        ++vi;
                    }
                    if (vi >= nv) break;
                    v = sh.valleys[vi];
                    const double gs = 1.0 / (double)(v - a);
                    for (int k = 1; k <= v - a; ++k)
                        cw[a + k] = __fsub_rn(cw[a + k], (float)(gs * (double)k));
                }
                const float gain = (float)p.clustered;
                for (int k = 0; k < L; ++k) cw[k] = __fmul_rn(cw[k], gain);
                blend_weights_dev(sh.w, cw, L);
            }
        }
        if (p.threshold_mult != 0.0) {
            const float mult = __fmul_rn(1.0f, (float)p.threshold_mult);
            for (int k = 0; k < L; ++k)
                sh.tmp[k] = ((double)sh.sv[k] < p.threshold_floor) ? 0.f : mult;
            blend_weights_dev(sh.w, sh.tmp, L);
        }
        if (p.header_max < 1.0) {
            const double hw = (double)sh.w[0];
            const double c = hw >= 0.0 ? (p.header_max < hw ? p.header_max : hw)
                                       : (-p.header_max > hw ? -p.header_max : hw);
            sh.w[0] = (float)c;
        }
    }
    __syncthreads();
    if (tid < L) {
        weights[(size_t)b * L + tid] = sh.w[tid];
        const double wj = (double)sh.w[tid];
        const double iw = p.max_guidance < wj ? p.max_guidance : wj;
        int mode;
        if (iw == 0.0) mode = 0;
        else if (fabs(iw) >= 1.0 - (double)sh.sv[tid]) mode = 1;
        else mode = 2;
        if (sh.status != 0) mode = 0;
        sh.mode[tid] = mode;
        sh.iwf[tid] = (float)iw;
    }
    if (tid == 0) status_out[b] = sh.status;
    __syncthreads();

    // ---------------- blend (guidance.py:258-271) -------------------------------------
    const float* altb = alt + (alt_batched ? (size_t)b * N * D : 0);
    const float* baseb = base + (size_t)b * L * D;
    float* outb = out + (size_t)b * L * D;
    const int D4 = D >> 2;
    for (int e = tid; e < L * D4; e += 256) {
        const int j = e / D4, d = (e - j * D4) * 4;
        const int mode = sh.mode[j];
        const float4 bv = *reinterpret_cast<const float4*>(baseb + (size_t)j * D + d);
        float4 r = bv;
        if (mode != 0) {
            const float4 av =
                *reinterpret_cast<const float4*>(altb + (size_t)sh.idx[j] * D + d);
            if (mode == 1) {
                r = av;
            } else {
                const float iw = sh.iwf[j];
                r.x = __fadd_rn(bv.x, __fmul_rn(__fsub_rn(av.x, bv.x), iw));
                r.y = __fadd_rn(bv.y, __fmul_rn(__fsub_rn(av.y, bv.y), iw));
                r.z = __fadd_rn(bv.z, __fmul_rn(__fsub_rn(av.z, bv.z), iw));
                r.w = __fadd_rn(bv.w, __fmul_rn(__fsub_rn(av.w, bv.w), iw));
            }
        }
        *reinterpret_cast<float4*>(outb + (size_t)j * D + d) = r;
    }
}

__global__ void k_concept_override(const float* __restrict__ guide, const int* __restrict__ cm_idx,
                                   const int* __restrict__ ct_idx, const float* __restrict__ ct_s,
                                   float* __restrict__ out, int N, int L, int D) {
    const int j = blockIdx.x;  // text row j -> token j + 1
    if (j + 1 >= L) return;
    const int c = ct_idx[j];
    if (c - 1 < 0) return;
    if (!(ct_s[j] > 0.9f)) return;
    const int gi = cm_idx[c - 1];
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        out[(size_t)(j + 1) * D + d] = guide[(size_t)gi * D + d];
}

__global__ void k_header_pull(float* __restrict__ out, const float* __restrict__ hdr, int L,
                              int D) {
    float* row = out + (size_t)blockIdx.x * L * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float x = row[d];
        row[d] = __fadd_rn(x, __fmul_rn(__fsub_rn(hdr[d], x), 0.85f));
    }
}

// ------------------------------------------------------------------------------------
extern "C" int64_t fd_guidance_workspace_floats(int B, int N, int L) {
    return (int64_t)B * N * L;
}

static int check_shapes(int B, int N, int L, int D, const char* who) {
    FD_CHECK_ARG(B > 0 && N > 0 && L > 1 && D > 0, FD_EINVAL, "%s: non-positive dimension", who);
    FD_CHECK_ARG(D % 32 == 0, FD_ESHAPE, "%s: D=%d must be a multiple of 32", who, D);
    FD_CHECK_ARG(L <= 32 * SIM_CT && L <= TW_MAXL, FD_ESHAPE, "%s: L=%d > %d", who, L,
                 32 * SIM_CT);
    FD_CHECK_ARG(N <= 2048 && N <= 0xFFFF, FD_ESHAPE, "%s: N=%d too large", who, N);
    const size_t lds = ((sizeof(TweenShared) + 15) & ~15) + (size_t)N * (L | 1) * 4;
    FD_CHECK_ARG(lds <= 150 * 1024, FD_ESHAPE,
                 "%s: similarity table %zu B exceeds the 150 KiB LDS budget", who, lds);
    return FD_OK;
}

static int launch_tween(const float* sim, const float* base, const float* alt, const float* lin_w,
                        float* out, float* weights, int32_t* idx, float* s, int32_t* status,
                        int B, int alt_batched, int N, int L, int D, const fd_tween_params& p,
                        hipStream_t st) {
    const size_t lds = ((sizeof(TweenShared) + 15) & ~15) + (size_t)N * (L | 1) * 4;
    static size_t configured = 0;
    if (lds > configured) {
        FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_guidance_tween),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = lds;
    }
    hipLaunchKernelGGL(k_guidance_tween, dim3(B), dim3(256), lds, st, sim, base, alt, lin_w, out,
                       weights, idx, s, status, alt_batched, N, L, D, p);
    FD_CHECK_LAUNCH("k_guidance_tween");
    return FD_OK;
}

extern "C" int fd_guidance_map(const float* alt, const float* txt, float* ws, int32_t* idx,
                               float* s, int B, int alt_batched, int N, int L, int D, int order,
                               int reuse, void* stream) {
    FD_PLAN(fd_guidance_map(alt, txt, ws, idx, s, B, alt_batched, N, L, D, order, reuse, fd_s_));
    int rc = check_shapes(B, N, L, D, "fd_guidance_map");
    if (rc) return rc;
    FD_CHECK_ARG(alt && txt && ws && idx && s, FD_EINVAL, "fd_guidance_map: null pointer");
    FD_CHECK_ARG(order >= 0 && order <= 2, FD_EINVAL, "fd_guidance_map: order=%d", order);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_guidance_sim, dim3(fd_cdiv(N, 32), B), dim3(256), 0, st, alt, txt, ws,
                       alt_batched, N, L, D);
    FD_CHECK_LAUNCH("k_guidance_sim");
    fd_tween_params p;
    memset(&p, 0, sizeof(p));
    p.order = order;
    p.reuse = reuse;
    return launch_tween(ws, nullptr, nullptr, nullptr, nullptr, nullptr, idx, s, nullptr, B,
                        alt_batched, N, L, D, p, st);
}

extern "C" int fd_guidance_tween(const float* base, const float* alt, const float* lin_w,
                                 float* ws, float* out, float* weights, int32_t* idx, float* s,
                                 int32_t* status, int B, int alt_batched, int N, int L, int D,
                                 const fd_tween_params* p, void* stream) {
    if (fd_plan_recording() && p) {
        const fd_tween_params pc_ = *p;
        fd_plan_push([=](void* fd_s_) -> int {
            return fd_guidance_tween(base, alt, lin_w, ws, out, weights, idx, s, status, B, alt_batched, N, L, D, &pc_, fd_s_);
        });
    }
    int rc = check_shapes(B, N, L, D, "fd_guidance_tween");
    if (rc) return rc;
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(3u, __LINE__));
    FD_CHECK_ARG(base && alt && lin_w && ws && out && weights && idx && s && status && p,
                 FD_EINVAL, "fd_guidance_tween: null pointer");
    FD_CHECK_ARG(p->order >= 0 && p->order <= 2, FD_EINVAL, "fd_guidance_tween: order=%d",
                 p->order);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_guidance_sim, dim3(fd_cdiv(N, 32), B), dim3(256), 0, st, alt, base, ws,
                       alt_batched, N, L, D);
    FD_CHECK_LAUNCH("k_guidance_sim");
    return launch_tween(ws, base, alt, lin_w, out, weights, idx, s, status, B, alt_batched, N, L,
                        D, *p, st);
}

extern "C" int fd_guidance_concept_override(const float* guide, const int32_t* cm_idx,
                                            const int32_t* ct_idx, const float* ct_s, float* out,
                                            int N, int L, int D, void* stream) {
    FD_PLAN(fd_guidance_concept_override(guide, cm_idx, ct_idx, ct_s, out, N, L, D, fd_s_));
    FD_CHECK_ARG(guide && cm_idx && ct_idx && ct_s && out, FD_EINVAL,
                 "fd_guidance_concept_override: null pointer");
    FD_CHECK_ARG(N > 0 && L > 1 && D > 0, FD_EINVAL, "fd_guidance_concept_override: bad dims");
    hipLaunchKernelGGL(k_concept_override, dim3(L - 1), dim3(256), 0, (hipStream_t)stream, guide,
                       cm_idx, ct_idx, ct_s, out, N, L, D);
    FD_CHECK_LAUNCH("k_concept_override");
    return FD_OK;
}

extern "C" int fd_guidance_header_pull(float* out, const float* hdr, int B, int L, int D,
                                       void* stream) {
    FD_PLAN(fd_guidance_header_pull(out, hdr, B, L, D, fd_s_));
    FD_CHECK_ARG(out && hdr && B > 0 && L > 0 && D > 0, FD_EINVAL, "fd_guidance_header_pull");
    hipLaunchKernelGGL(k_header_pull, dim3(B), dim3(256), 0, (hipStream_t)stream, out, hdr, L, D);
    FD_CHECK_LAUNCH("k_header_pull");
    return FD_OK;
}
