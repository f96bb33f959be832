// HBM-bound normalisation kernels for gfx950: GroupNorm(+SiLU) on NHWC fp16 activations,
// LayerNorm over the channel dimension, and a row softmax (VAE single-head attention).
// All loads/stores are 16-byte vectors (8 halfs per lane); statistics in fp32/fp64.
//
// GroupNorm is two launches over an L2/MALL-resident tensor:
//   k_gn_stats : grid (chunks, B).  Thread t owns the fixed 8-channel chunk t % (C/8) and
//                walks pixels t / (C/8), +PL, ... of its slab, so per-channel sums stay in
//                registers; per-(sample, chunk, group) partial (sum, sumsq) go to a small
//                fp32 workspace (deterministic, no atomics).
//   k_gn_apply : grid (chunks, B).  Combines the partials in fp64, folds mean/rstd/gamma/
//                beta into per-channel (scale, shift) in LDS, then streams the slab:
//                y = x*scale + shift, optional SiLU.
// Algorithmic HBM bytes: 2 B read (stats) + 2 B read + 2 B write (apply) per element.
#include <stdlib.h>

#include "common.h"
#include "gn_slab.h"

#define GN_MAX_CHUNKS 256

__global__ void k_gn_stats(const half_t* __restrict__ x, float* __restrict__ part, int HW, int C,
                           int G, int PL, int pix_per_chunk, int ldx) {
    extern __shared__ float sm[];  // [PL][C][2]
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c8 = C >> 3;
    const int tid = threadIdx.x;
    const int cc = tid % c8, pl = tid / c8;
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(HW, p0 + pix_per_chunk);
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = q[k] = 0.f;
    if (pl < PL) {
        const half_t* base = x + ((size_t)b * HW) * ldx + cc * 8;   // ldx: row stride of x (>= C)
        int p = p0 + pl;
        // 4 independent 16-byte loads per group, and the NEXT group issued before this one is consumed (8 in flight):
        // with one workgroup per CU the kernel is latency-bound otherwise
        uint4 raw[4], nxt[4];
        bool have = p + 3 * PL < p1;
        if (have) {
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (size_t)(p + u * PL) * ldx);
        }
        while (have) {
            const int pn = p + 4 * PL;
            const bool more = pn + 3 * PL < p1;
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) nxt[u] = *reinterpret_cast<const uint4*>(base + (size_t)(pn + u * PL) * ldx);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const half8 v = *reinterpret_cast<const half8*>(&raw[u]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float f = (float)v[k];
                    s[k] += f;
                    q[k] = fmaf(f, f, q[k]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = nxt[u];
            p = pn;
            have = more;
        }
        for (; p < p1; p += PL) {
            const uint4 raw = *reinterpret_cast<const uint4*>(base + (size_t)p * ldx);
            const half8 v = *reinterpret_cast<const half8*>(&raw);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float f = (float)v[k];
                s[k] += f;
                q[k] = fmaf(f, f, q[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sm[((size_t)pl * C + cc * 8 + k) * 2 + 0] = s[k];
            sm[((size_t)pl * C + cc * 8 + k) * 2 + 1] = q[k];
        }
    }
    __syncthreads();
    // (1) per-channel sums over the PL pixel lanes, one channel per thread; (2) per-group sums
    // with a wavefront reduction (group g handled by wave g % nwaves).  Fixed order, no atomics.
    const int nt = blockDim.x;
    for (int c = tid; c < C; c += nt) {
        float a = 0.f, bq = 0.f;
#pragma unroll 4
        for (int l = 0; l < PL; ++l) {
            a += sm[((size_t)l * C + c) * 2 + 0];
            bq += sm[((size_t)l * C + c) * 2 + 1];
        }
        sm[(size_t)c * 2 + 0] = a;     // row l = 0 is only read by its own thread: safe in place
        sm[(size_t)c * 2 + 1] = bq;
    }
    __syncthreads();
    const int cpg = C / G, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    for (int g = wave; g < G; g += nw) {
        float a = 0.f, bq = 0.f;
        for (int i = lane; i < cpg; i += 64) {
            a += sm[(size_t)(g * cpg + i) * 2 + 0];
            bq += sm[(size_t)(g * cpg + i) * 2 + 1];
        }
        a = fd_wave_sum(a);
        bq = fd_wave_sum(bq);
        if (lane == 0) {
            float* dst = part + (((size_t)b * gridDim.x + chunk) * G + g) * 2;
            dst[0] = a;
            dst[1] = bq;
        }
    }
}

// y = x*scale + shift (+SiLU).  Prologue: all threads combine the per-chunk partials (fp64,
// fixed order, loads unrolled so they are in flight together).  Then thread t owns the fixed
// 8-channel chunk t % (C/8): its 16 coefficients live in registers and the loop is pure
// 16-byte streaming with 4 loads in flight (same thread -> address map as k_gn_stats).
__global__ void k_gn_apply(const half_t* __restrict__ x, half_t* __restrict__ y,
                           const float* __restrict__ part, const float* __restrict__ gamma,
                           const float* __restrict__ beta, int HW, int C, int G,
                           int nchunk_stats, int PL, int pix_per_chunk, float eps, int silu, int ldx) {
    extern __shared__ float sm[];   // [2][nsub*G] doubles, then [G][2] floats (mean, rstd)
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int nt = blockDim.x, nsub = nt / G;
    double* red = reinterpret_cast<double*>(sm);
    float* gst = reinterpret_cast<float*>(red + 2 * nsub * G);
    {
        const int g = tid % G, sub = tid / G;
        if (sub < nsub) {
            double s = 0.0, q = 0.0;
            const float2* src = reinterpret_cast<const float2*>(part) + ((size_t)b * nchunk_stats) * G + g;
            int k = sub;
            for (; k + 3 * nsub < nchunk_stats; k += 4 * nsub) {
                float2 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(k + u * nsub) * G];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s += (double)v[u].x;
                    q += (double)v[u].y;
                }
            }
            for (; k < nchunk_stats; k += nsub) {
                const float2 v = src[(size_t)k * G];
                s += (double)v.x;
                q += (double)v.y;
            }
            red[sub * G + g] = s;
            red[nsub * G + sub * G + g] = q;
        }
    }
    __syncthreads();
    const int cpg = C / G;
    if (tid < G) {
        double s = 0.0, q = 0.0;
        for (int u = 0; u < nsub; ++u) {
            s += red[u * G + tid];
            q += red[nsub * G + u * G + tid];
        }
        const double n = (double)HW * cpg;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        gst[tid * 2 + 0] = (float)mean;
        gst[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int c8 = C >> 3;
    const int cc = tid % c8, pl = tid / c8;
    if (pl >= PL) return;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cc * 8 + k, g = c / cpg;
        const float a = gst[g * 2 + 1] * gamma[c];
        sc[k] = a;
        sh[k] = beta[c] - gst[g * 2 + 0] * a;
    }
    const int p0 = chunk * pix_per_chunk;
    const int p1 = min(HW, p0 + pix_per_chunk);
    const half_t* xb = x + ((size_t)b * HW) * ldx + cc * 8;
    half_t* yb = y + ((size_t)b * HW) * C + cc * 8;
    int p = p0 + pl;
    // software-pipelined: the NEXT group's 4 loads are issued before this group's stores.  Loads issued after
    // the stores could only be waited for with a vmcnt that also drains those stores (one counter, in order): a
    // store round trip plus a load round trip per group with only 2 waves per SIMD to hide them
    uint4 raw[4], nxt[4];
    bool have = p + 3 * PL < p1;
    if (have) {
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4*>(xb + (size_t)(p + u * PL) * ldx);
    }
    while (have) {
        const int pn = p + 4 * PL;
        const bool more = pn + 3 * PL < p1;
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; ++u) nxt[u] = *reinterpret_cast<const uint4*>(xb + (size_t)(pn + u * PL) * ldx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const half8 v = *reinterpret_cast<const half8*>(&raw[u]);
            half8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (half_t)gn_act(fmaf((float)v[k], sc[k], sh[k]), silu);
            *reinterpret_cast<uint4*>(yb + (size_t)(p + u * PL) * C) = *reinterpret_cast<uint4*>(&o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = nxt[u];
        p = pn;
        have = more;
    }
    for (; p < p1; p += PL) {
        const uint4 raw = *reinterpret_cast<const uint4*>(xb + (size_t)p * ldx);
        const half8 v = *reinterpret_cast<const half8*>(&raw);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)gn_act(fmaf((float)v[k], sc[k], sh[k]), silu);
        *reinterpret_cast<uint4*>(yb + (size_t)p * C) = *reinterpret_cast<uint4*>(&o);
    }
}

// Slab GroupNorm(+SiLU) in ONE launch and ONE read of x: a workgroup of NT threads owns one
// sample x GB consecutive groups; its [HW][GB*cpg] slab is read once into registers, statistics are reduced through
// LDS in a fixed order, then the same registers are normalised and stored (gn_slab.h: the body is shared with the
// split-K finish of gemm.hip).  4 B of HBM traffic per element instead of 6, no partials, no second
// launch.  <256,16> serves the 16x16 / 8x8 UNet levels, <1024,22> slabs up to ~400 KB (the
// 64x64 and 32x32 levels whenever (HW * cpg*GB/8) / 1024 <= 22).
template <int NT, int NV>
__global__ __launch_bounds__(NT) void k_gn_slab(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int HW, int C,
                                                int G, int GB, float eps, int silu, int ldx) {
    extern __shared__ float sm[];
    const int cpg = C / G, CB = cpg * GB, c8 = CB >> 3;
    const int PL = NT / c8;
    // grid (B, G / GB): consecutive workgroup ids go round-robin over the 8 XCDs, so with the SAMPLE as the fast index all group
    // blocks of one sample -- whose 40..160-byte row segments share 128-byte lines -- run on one XCD and meet in its L2
    const int b = blockIdx.x, ch0 = blockIdx.y * CB, tid = threadIdx.x;
    const int cc = tid % c8, pl = tid / c8;
    // uniform 64-bit base + 32-bit per-lane offsets (keeps the address math out of VGPR pairs)
    const half_t* xb = x + (size_t)b * HW * ldx + ch0;
    const unsigned xoff0 = (unsigned)(pl * ldx + cc * 8), xstep = (unsigned)(PL * ldx);
    const bool active = pl < PL;
    gn_slab_body<NT, NV>(
        [&](uint4(&v)[NV]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int p = pl + PL * i;
                v[i] = make_uint4(0u, 0u, 0u, 0u);
                if (active && p < HW) v[i] = *reinterpret_cast<const uint4*>(xb + (xoff0 + xstep * i));
            }
        },
        y + (size_t)b * HW * C + ch0, gamma + ch0, beta + ch0, HW, C, cpg, GB, eps, silu, sm);
}

template <int NT, int NV>
static bool gn_try_slab(const void* x, void* y, const float* gamma, const float* beta, int B,
                        int HW, int C, int G, float eps, int silu, hipStream_t st, int* rc, int ldx) {
    size_t lds = 0;
    const int GB = gn_slab_pick<NT, NV>(HW, C, G, &lds);
    if (!GB) return false;
    static std::atomic<unsigned long long> attr_set{0};   // per instantiation and device
    if (fd_first_on_device(&attr_set)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_gn_slab<NT, NV>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            fd_set_error("fd_groupnorm_nhwc_f16: cannot raise the LDS limit");
            *rc = FD_EHIP;
            return true;
        }
    }
    fd_prof_begin(FD_FAMILY_GROUPNORM, st, (double)B * HW * C * 4.0, -1.0, fd_tag(7u, B, HW, C, NT));
    // grid (B, G / GB), sample fastest (round 4): -10..-20 % on every slab shape against (G / GB, B)
    // (profiles/r04_session_ab.txt sec. 8: 16x1024x640 17.6 -> 14.0 us, 16x256x1280 10.2 -> 8.3 us)
    hipLaunchKernelGGL((k_gn_slab<NT, NV>), dim3(B, G / GB), dim3(NT), lds, st, (const half_t*)x,
                       (half_t*)y, gamma, beta, HW, C, G, GB, eps, silu, ldx);
    fd_prof_end(FD_FAMILY_GROUPNORM, st);
    *rc = hipGetLastError() == hipSuccess ? FD_OK : FD_EHIP;
    if (*rc != FD_OK) fd_set_error("k_gn_slab: launch failed");
    return true;
}

// ---------------------------------------------------------------------------------------
// GroupNorm folded into the linear layer that consumes it (fd_groupnorm_fold_linear_f16): after the statistics pass the
// per-sample weights w_out[b][n][c] = wg[n][c] * rstd[b][g(c)] and biases
// bias_out[b][n] = biasf[n] - sum_g mean[b][g] rstd[b][g] S[n][g] are all the "apply" there is -- the consumer GEMM reads
// the un-normalised activation with sample b's weights.  grid (N / ROWS, B); 16-byte chunks of the weight rows.
// ---------------------------------------------------------------------------------------
#define GNF_ROWS 16
__global__ __launch_bounds__(256) void k_gn_fold_linear(const float* __restrict__ part, int nchunk_stats, int HW, int C, int G, float eps,
                                                        const half_t* __restrict__ wg,
                                                        const float* __restrict__ biasf, int N, half_t* __restrict__ w_out,
                                                        float* __restrict__ bias_out) {
    __shared__ float rstd_s[64], mean_s[64];   // G <= 64
    __shared__ float gsum_s[GNF_ROWS][64];
    __shared__ double red_s[2][256];           // [stat][sub * G + g]
    extern __shared__ __align__(16) unsigned char gnf_lds[];   // the block's ROUNDED output weights, fp16 [GNF_ROWS][C]
    half_t* wr = reinterpret_cast<half_t*>(gnf_lds);
    const int b = blockIdx.y, tid = threadIdx.x;
    // the combine of the partial sums, fp64, in a fixed order -- spread over all 256 threads (sub-sums of every nsub-th chunk, then nsub values
    // per group): with one thread per group walking 16-32 chunks the serial loop of dependent-latency loads was most of this 12-us kernel
    const int nsub = 256 / G;
    {
        const int g = tid % G, sub = tid / G;
        if (sub < nsub) {
            double s = 0.0, q = 0.0;
            const float2* src = reinterpret_cast<const float2*>(part) + ((size_t)b * nchunk_stats) * G + g;
            for (int k = sub; k < nchunk_stats; k += nsub) {
                const float2 v = src[(size_t)k * G];
                s += (double)v.x;
                q += (double)v.y;
            }
            red_s[0][sub * G + g] = s;
            red_s[1][sub * G + g] = q;
        }
    }
    __syncthreads();
    if (tid < G) {
        double s = 0.0, q = 0.0;
        for (int u = 0; u < nsub; ++u) {
            s += red_s[0][u * G + tid];
            q += red_s[1][u * G + tid];
        }
        const double n = (double)HW * (C / G);
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        rstd_s[tid] = (float)rstd;
        mean_s[tid] = (float)mean;
    }
    __syncthreads();
    const int n0 = blockIdx.x * GNF_ROWS, rows = min(GNF_ROWS, N - n0);
    const int c8 = C >> 3, cpg = C / G;
    for (int i = tid; i < rows * c8; i += 256) {
        const int r = i / c8, cc = i - r * c8;
        const half8 v = *reinterpret_cast<const half8*>(wg + (size_t)(n0 + r) * C + cc * 8);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)((float)v[k] * rstd_s[(cc * 8 + k) / cpg]);
        *reinterpret_cast<half8*>(w_out + ((size_t)b * N + n0 + r) * C + cc * 8) = o;
        *reinterpret_cast<half8*>(wr + (size_t)r * C + cc * 8) = o;
    }
    __syncthreads();
    // The consumer GEMM computes sum_c w_out[c] x[c] + bias with the RE-ROUNDED w_out, so the mean term must be taken
    // over those same fp16 values -- bias = biasf - sum_g mean_g * (sum_{c in g} w_out[c]) -- for the group mean to
    // cancel exactly whatever its size relative to the spread (a mean of 10 sigma would otherwise leave 10 * 2^-11 of
    // sigma per weight instead of 2^-11).  Fixed summation order: deterministic.
    for (int i = tid; i < rows * G; i += 256) {
        const int r = i / G, g = i - r * G;
        const half_t* src = wr + (size_t)r * C + g * cpg;
        float acc = 0.f;
        for (int c = 0; c < cpg; ++c) acc += (float)src[c];
        gsum_s[r][g] = acc;
    }
    __syncthreads();
    if (tid < rows) {
        float acc = biasf[n0 + tid];
        for (int g = 0; g < G; ++g) acc = fmaf(-mean_s[g], gsum_s[tid][g], acc);
        bias_out[(size_t)b * N + n0 + tid] = acc;
    }
}

// launch shape of the streaming statistics pass (shared by fd_groupnorm_nhwc_ld_f16's two-pass form and the fold)
static void gn_stats_shape(int B, int HW, int C, int* PL, int* threads, int* nchunk, int* ppc) {
    const int c8 = C / 8;
    int pl = 512 / c8;
    if (pl < 1) pl = 1;
    if (pl > HW) pl = HW;
    int nc = 256 / B;
    if (nc < 1) nc = 1;
    if (nc > GN_MAX_CHUNKS) nc = GN_MAX_CHUNKS;
    int pp = fd_cdiv(HW, nc);
    if (pp < pl) pp = pl;
    *PL = pl;
    *threads = ((c8 * pl + 63) / 64) * 64;
    *ppc = pp;
    *nchunk = fd_cdiv(HW, pp);
}

extern "C" int fd_groupnorm_fold_linear_f16(const void* x, int ldx, float* ws, int B, int HW, int C, int G, float eps,
                                            const void* wg, const float* biasf, int N,
                                            void* w_out, float* bias_out, void* stream) {
    FD_PLAN(fd_groupnorm_fold_linear_f16(x, ldx, ws, B, HW, C, G, eps, wg, biasf, N, w_out, bias_out, fd_s_));
    FD_CHECK_ARG(x && ws && wg && biasf && w_out && bias_out, FD_EINVAL, "fd_groupnorm_fold_linear_f16: null pointer");
    FD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && N > 0, FD_EINVAL, "fd_groupnorm_fold_linear_f16: bad dims");
    FD_CHECK_ARG(C % 8 == 0 && C % G == 0 && G <= 64 && C / 8 <= 1024, FD_ESHAPE,
                 "fd_groupnorm_fold_linear_f16: C=%d must be a multiple of 8 and of G=%d (G <= 64)", C, G);
    FD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && ((uintptr_t)x | (uintptr_t)wg | (uintptr_t)w_out) % 16 == 0, FD_ESHAPE,
                 "fd_groupnorm_fold_linear_f16: ldx=%d must be >= C=%d, a multiple of 8; x / wg / w_out 16-byte aligned", ldx, C);
    FD_CHECK_ARG((long long)B * HW * ldx < 0x7fffffffLL, FD_ESHAPE, "fd_groupnorm_fold_linear_f16: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    int PL, threads, nchunk, ppc;
    gn_stats_shape(B, HW, C, &PL, &threads, &nchunk, &ppc);
    const size_t lds1 = (size_t)PL * C * 2 * sizeof(float);
    FD_CHECK_ARG(lds1 <= 64 * 1024, FD_ESHAPE, "fd_groupnorm_fold_linear_f16: stats LDS too large");
    const size_t lds2 = (size_t)GNF_ROWS * C * sizeof(half_t);
    FD_CHECK_ARG(lds2 <= 48 * 1024, FD_ESHAPE, "fd_groupnorm_fold_linear_f16: C=%d too wide for the fold kernel's weight tile", C);
    // priced as the statistics read only (2 B/element): the apply pass it replaces is gone
    fd_prof_begin(FD_FAMILY_GROUPNORM, st, (double)B * HW * C * 2.0, -1.0, fd_tag(8u, B, HW, C, N));
    hipLaunchKernelGGL(k_gn_stats, dim3(nchunk, B), dim3(threads), lds1, st, (const half_t*)x, ws, HW, C, G, PL, ppc, ldx);
    hipLaunchKernelGGL(k_gn_fold_linear, dim3(fd_cdiv(N, GNF_ROWS), B), dim3(256), lds2, st, (const float*)ws, nchunk, HW, C, G, eps,
                       (const half_t*)wg, biasf, N, (half_t*)w_out, bias_out);
    fd_prof_end(FD_FAMILY_GROUPNORM, st);
    FD_CHECK_LAUNCH("k_gn_stats/k_gn_fold_linear");
    return FD_OK;
}

extern "C" int fd_groupnorm_fold_linear_parts_f16(const float* parts, int chunks, int B, int HW, int C, int G, float eps,
                                                  const void* wg, const float* biasf, int N, void* w_out, float* bias_out, void* stream) {
    FD_PLAN(fd_groupnorm_fold_linear_parts_f16(parts, chunks, B, HW, C, G, eps, wg, biasf, N, w_out, bias_out, fd_s_));
    FD_CHECK_ARG(parts && wg && biasf && w_out && bias_out, FD_EINVAL, "fd_groupnorm_fold_linear_parts_f16: null pointer");
    FD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && N > 0 && chunks > 0, FD_EINVAL, "fd_groupnorm_fold_linear_parts_f16: bad dims");
    FD_CHECK_ARG(C % 8 == 0 && C % G == 0 && G <= 64 && ((uintptr_t)wg | (uintptr_t)w_out) % 16 == 0, FD_ESHAPE,
                 "fd_groupnorm_fold_linear_parts_f16: C=%d must be a multiple of 8 and of G=%d (G <= 64); wg / w_out 16-byte aligned", C, G);
    const size_t lds2 = (size_t)GNF_ROWS * C * sizeof(half_t);
    FD_CHECK_ARG(lds2 <= 48 * 1024, FD_ESHAPE, "fd_groupnorm_fold_linear_parts_f16: C=%d too wide for the fold kernel's weight tile", C);
    hipStream_t st = (hipStream_t)stream;
    fd_prof_begin(FD_FAMILY_GROUPNORM, st, 0.0, -1.0, fd_tag(12u, B, HW, C, N));   // no pass over the activation at all
    hipLaunchKernelGGL(k_gn_fold_linear, dim3(fd_cdiv(N, GNF_ROWS), B), dim3(256), lds2, st, parts, chunks, HW, C, G, eps,
                       (const half_t*)wg, biasf, N, (half_t*)w_out, bias_out);
    fd_prof_end(FD_FAMILY_GROUPNORM, st);
    FD_CHECK_LAUNCH("k_gn_fold_linear");
    return FD_OK;
}

extern "C" int64_t fd_groupnorm_workspace_floats(int B, int G) {
    return (int64_t)B * GN_MAX_CHUNKS * G * 2;
}

extern "C" int fd_groupnorm_nhwc_f16(const void* x, void* y, const float* gamma,
                                     const float* beta, float* ws, int B, int HW, int C, int G,
                                     float eps, int silu, void* stream) {
    return fd_groupnorm_nhwc_ld_f16(x, C, y, gamma, beta, ws, B, HW, C, G, eps, silu, stream);
}

extern "C" int fd_groupnorm_nhwc_ld_f16(const void* x, int ldx, void* y, const float* gamma,
                                        const float* beta, float* ws, int B, int HW, int C, int G,
                                        float eps, int silu, void* stream) {
    FD_PLAN(fd_groupnorm_nhwc_ld_f16(x, ldx, y, gamma, beta, ws, B, HW, C, G, eps, silu, fd_s_));
    FD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (uintptr_t)x % 16 == 0, FD_ESHAPE,
                 "fd_groupnorm_nhwc_ld_f16: ldx=%d must be >= C=%d, a multiple of 8, x 16-byte aligned", ldx, C);
    FD_CHECK_ARG((long long)B * HW * ldx < 0x7fffffffLL, FD_ESHAPE, "fd_groupnorm_nhwc_ld_f16: tensor too large");
    FD_CHECK_ARG(x && y && gamma && beta && ws, FD_EINVAL, "fd_groupnorm_nhwc_f16: null pointer");
    FD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0, FD_EINVAL, "fd_groupnorm_nhwc_f16: bad dims");
    FD_CHECK_ARG(C % 8 == 0 && C % G == 0 && G <= 64, FD_ESHAPE,
                 "fd_groupnorm_nhwc_f16: C=%d must be a multiple of 8 and of G=%d (G<=64)", C, G);
    hipStream_t st = (hipStream_t)stream;
    const int c8 = C / 8;
    FD_CHECK_ARG(c8 <= 1024, FD_ESHAPE, "fd_groupnorm_nhwc_f16: C=%d too large", C);
    // slabs that fit a workgroup's registers: one launch, one read
    {
        const int off = 0;
        int rc = FD_OK;
        if (gn_try_slab<256, 16>(x, y, gamma, beta, B, HW, C, G, eps, silu, st, &rc, ldx)) return rc;
        // 1024-thread slabs pay off up to 32x32 maps; at 64x64 the 80-byte rows of a narrow slab
        // waste cache lines and the two streaming passes below are as fast or faster (measured)
        if (off == 0 && HW <= 1024 && gn_try_slab<1024, 22>(x, y, gamma, beta, B, HW, C, G, eps, silu, st, &rc, ldx)) return rc;
    }
    int PL = 512 / c8;
    if (PL < 1) PL = 1;
    if (PL > HW) PL = HW;
    const int threads = ((c8 * PL + 63) / 64) * 64;
    // one workgroup per CU: with 4x more (smaller) chunks the statistics pass was latency-bound
    // (16x4096x320: 36 -> 29 us for the pair; 128/B, 192/B, 384/B ... 4096/B all measured slower)
    int nchunk = 256 / B;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
    int ppc = fd_cdiv(HW, nchunk);
    if (ppc < PL) ppc = PL;
    nchunk = fd_cdiv(HW, ppc);
    const size_t lds1 = (size_t)PL * C * 2 * sizeof(float);
    FD_CHECK_ARG(lds1 <= 64 * 1024, FD_ESHAPE, "fd_groupnorm_nhwc_f16: stats LDS too large");
    // priced at SURVEY 8(d)'s ALGORITHMIC 4 B/element (fp16 read + write); the streaming pair really
    // moves 6 B/element (the statistics pass reads x once more, mostly from the Infinity Cache)
    const double bytes = (double)B * HW * C * 4.0;
    fd_prof_begin(FD_FAMILY_GROUPNORM, st, bytes, -1.0, fd_tag(9u, B, HW, C, ldx));
    hipLaunchKernelGGL(k_gn_stats, dim3(nchunk, B), dim3(threads), lds1, st, (const half_t*)x, ws,
                       HW, C, G, PL, ppc, ldx);
    const size_t lds2 = (size_t)2 * threads * sizeof(double) + (size_t)2 * G * sizeof(float);
    hipLaunchKernelGGL(k_gn_apply, dim3(nchunk, B), dim3(threads), lds2, st, (const half_t*)x,
                       (half_t*)y, (const float*)ws, gamma, beta, HW, C, G, nchunk, PL, ppc, eps, silu, ldx);
    fd_prof_end(FD_FAMILY_GROUPNORM, st);
    FD_CHECK_LAUNCH("k_gn_stats/k_gn_apply");
    return FD_OK;
}

// The apply pass alone: the statistics come as partial sums written by the PRODUCER of x (fd_gemm_desc.gn_part_out: the lean
// epilogue of the convolution that wrote x), [B][chunks][G][2] -- the layout k_gn_stats writes.
extern "C" int fd_groupnorm_apply_parts_f16(const void* x, int ldx, void* y, const float* gamma, const float* beta, const float* parts,
                                            int chunks, int B, int HW, int C, int G, float eps, int silu, void* stream) {
    FD_PLAN(fd_groupnorm_apply_parts_f16(x, ldx, y, gamma, beta, parts, chunks, B, HW, C, G, eps, silu, fd_s_));
    FD_CHECK_ARG(x && y && gamma && beta && parts, FD_EINVAL, "fd_groupnorm_apply_parts_f16: null pointer");
    FD_CHECK_ARG(B > 0 && HW > 0 && C > 0 && G > 0 && chunks > 0, FD_EINVAL, "fd_groupnorm_apply_parts_f16: bad dims");
    FD_CHECK_ARG(ldx >= C && ldx % 8 == 0 && (uintptr_t)x % 16 == 0 && C % 8 == 0 && C % G == 0 && G <= 64 && C / 8 <= 1024, FD_ESHAPE,
                 "fd_groupnorm_apply_parts_f16: ldx=%d >= C=%d, both multiples of 8, C a multiple of G=%d (G <= 64), x 16-byte aligned", ldx, C, G);
    FD_CHECK_ARG((long long)B * HW * ldx < 0x7fffffffLL, FD_ESHAPE, "fd_groupnorm_apply_parts_f16: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    const int c8 = C / 8;
    int PL = 512 / c8;
    if (PL < 1) PL = 1;
    if (PL > HW) PL = HW;
    const int threads = ((c8 * PL + 63) / 64) * 64;
    int nchunk = 256 / B;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > GN_MAX_CHUNKS) nchunk = GN_MAX_CHUNKS;
    int ppc = fd_cdiv(HW, nchunk);
    if (ppc < PL) ppc = PL;
    nchunk = fd_cdiv(HW, ppc);
    const size_t lds2 = (size_t)2 * threads * sizeof(double) + (size_t)2 * G * sizeof(float);
    fd_prof_begin(FD_FAMILY_GROUPNORM, st, (double)B * HW * C * 4.0, -1.0, fd_tag(13u, B, HW, C, ldx));
    hipLaunchKernelGGL(k_gn_apply, dim3(nchunk, B), dim3(threads), lds2, st, (const half_t*)x, (half_t*)y, parts, gamma, beta, HW, C, G,
                       chunks, PL, ppc, eps, silu, ldx);
    fd_prof_end(FD_FAMILY_GROUPNORM, st);
    FD_CHECK_LAUNCH("k_gn_apply");
    return FD_OK;
}

// ---------------------------------------------------------------------------------------
// LayerNorm over the last dim: one wavefront per row, the row lives in registers, exact
// two-pass statistics (mean, then sum of squared deviations) with wave shuffles.
template <int VPL, int RPW, bool STATS = false>  // uint4 vectors per lane: C <= 512*VPL; RPW rows per wavefront;
// STATS: write (rstd, -mean * rstd) per row to y instead of the normalised row (LayerNorm fold of fd_gemm_f16)
__global__ __launch_bounds__(256) void k_layernorm(const half_t* __restrict__ x, void* __restrict__ y,
                                                   const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, int rows, int C,
                                                   int ldx, int ldy, float eps, int out_f32) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    // all loads of the wave's RPW rows are issued before the first reduction (one 16-byte load
    // per lane per row otherwise: the kernel lives on occupancy alone)
    half8 v[RPW][VPL];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const half_t* xr = x + (size_t)min(row0 + r, rows - 1) * ldx;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 8;
            uint4 raw = make_uint4(0u, 0u, 0u, 0u);
            if (c < C) raw = *reinterpret_cast<const uint4*>(xr + c);
            v[r][i] = *reinterpret_cast<half8*>(&raw);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += (float)v[r][i][k];
        const float mean = fd_wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c < C) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float dlt = (float)v[r][i][k] - mean;
                    sq = fmaf(dlt, dlt, sq);
                }
            }
        }
        const float rstd = rsqrtf(fd_wave_sum(sq) / (float)C + eps);
        if constexpr (STATS) {
            if (lane == 0) reinterpret_cast<float2*>(y)[row] = make_float2(rstd, -mean * rstd);
            continue;
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = (lane + 64 * i) * 8;
            if (c >= C) continue;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                o[k] = ((float)v[r][i][k] - mean) * rstd * gamma[c + k] + beta[c + k];
            if (out_f32) {
                float* yr = reinterpret_cast<float*>(y) + (size_t)row * ldy + c;
                *reinterpret_cast<float4*>(yr) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(yr + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
                half8 h;
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = (half_t)o[k];
                *reinterpret_cast<uint4*>(reinterpret_cast<half_t*>(y) + (size_t)row * ldy + c) =
                    *reinterpret_cast<uint4*>(&h);
            }
        }
    }
}

extern "C" int fd_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta,
                                int rows, int C, int ldx, int ldy, float eps, int out_f32,
                                void* stream) {
    FD_PLAN(fd_layernorm_f16(x, y, gamma, beta, rows, C, ldx, ldy, eps, out_f32, fd_s_));
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(2u, __LINE__));
    FD_CHECK_ARG(x && y && gamma && beta && rows > 0 && C > 0, FD_EINVAL, "fd_layernorm_f16: args");
    FD_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && C <= 2048, FD_ESHAPE,
                 "fd_layernorm_f16: C=%d must be a multiple of 8 and <= 2048", C);
    hipStream_t st = (hipStream_t)stream;
    // two rows per wavefront for the narrow, tall case (level-0 hidden states, 65536 x 320:
    // 28.9 -> 19.0 us; four rows 20.7 us; at C = 640 two rows are 4 % slower than one)
    if (C <= 512 && rows >= 8192)
        hipLaunchKernelGGL((k_layernorm<1, 2>), dim3(fd_cdiv(rows, 8)), dim3(256), 0, st, (const half_t*)x, y,
                           gamma, beta, rows, C, ldx, ldy, eps, out_f32);
    else if (C <= 512)
        hipLaunchKernelGGL((k_layernorm<1, 1>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x, y,
                           gamma, beta, rows, C, ldx, ldy, eps, out_f32);
    else if (C <= 1024)
        hipLaunchKernelGGL((k_layernorm<2, 1>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x, y,
                           gamma, beta, rows, C, ldx, ldy, eps, out_f32);
    else
        hipLaunchKernelGGL((k_layernorm<4, 1>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x, y,
                           gamma, beta, rows, C, ldx, ldy, eps, out_f32);
    FD_CHECK_LAUNCH("k_layernorm");
    return FD_OK;
}

extern "C" int fd_ln_row_stats_f16(const void* x, float* stats, int rows, int C, int ldx, float eps, void* stream) {
    FD_PLAN(fd_ln_row_stats_f16(x, stats, rows, C, ldx, eps, fd_s_));
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(2u, __LINE__));
    FD_CHECK_ARG(x && stats && rows > 0 && C > 0, FD_EINVAL, "fd_ln_row_stats_f16: args");
    FD_CHECK_ARG(C % 8 == 0 && ldx % 8 == 0 && C <= 2048, FD_ESHAPE,
                 "fd_ln_row_stats_f16: C=%d must be a multiple of 8 and <= 2048", C);
    hipStream_t st = (hipStream_t)stream;
    const float* nul = nullptr;
    if (C <= 512 && rows >= 8192)
        hipLaunchKernelGGL((k_layernorm<1, 2, true>), dim3(fd_cdiv(rows, 8)), dim3(256), 0, st, (const half_t*)x,
                           (void*)stats, nul, nul, rows, C, ldx, 0, eps, 0);
    else if (C <= 512)
        hipLaunchKernelGGL((k_layernorm<1, 1, true>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x,
                           (void*)stats, nul, nul, rows, C, ldx, 0, eps, 0);
    else if (C <= 1024)
        hipLaunchKernelGGL((k_layernorm<2, 1, true>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x,
                           (void*)stats, nul, nul, rows, C, ldx, 0, eps, 0);
    else
        hipLaunchKernelGGL((k_layernorm<4, 1, true>), dim3(fd_cdiv(rows, 4)), dim3(256), 0, st, (const half_t*)x,
                           (void*)stats, nul, nul, rows, C, ldx, 0, eps, 0);
    FD_CHECK_LAUNCH("k_layernorm<stats>");
    return FD_OK;
}

// ---------------------------------------------------------------------------------------
// Row softmax in place on fp16 [rows][ld] (first N columns), fp32 math, one block per row.
__global__ __launch_bounds__(256) void k_softmax_rows(half_t* __restrict__ x, int N, int ld,
                                                      float scale) {
    __shared__ float red[4];
    half_t* row = x + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -INFINITY;
    for (int c = tid * 8; c < N; c += 2048) {
        const uint4 raw = *reinterpret_cast<const uint4*>(row + c);
        const half8 v = *reinterpret_cast<const half8*>(&raw);
#pragma unroll
        for (int k = 0; k < 8; ++k) m = fmaxf(m, (float)v[k] * scale);
    }
    m = fd_wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = tid * 8; c < N; c += 2048) {
        const uint4 raw = *reinterpret_cast<const uint4*>(row + c);
        const half8 v = *reinterpret_cast<const half8*>(&raw);
#pragma unroll
        for (int k = 0; k < 8; ++k) s += __expf((float)v[k] * scale - m);
    }
    s = fd_wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = tid * 8; c < N; c += 2048) {
        const uint4 raw = *reinterpret_cast<const uint4*>(row + c);
        const half8 v = *reinterpret_cast<const half8*>(&raw);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)(__expf((float)v[k] * scale - m) * inv);
        *reinterpret_cast<uint4*>(row + c) = *reinterpret_cast<uint4*>(&o);
    }
}

// ---- LayerNorm statistics from per-n-tile partial sums (fd_gemm_desc.ln_stats_out with N > 320) -----------------
// parts [n_tiles][M][2] = (sum, sum of squares) of each row over one 160-column tile of the fp16 values the producing
// GEMM stored; combined in a fixed order -> stats [M][2] = (rstd, -mean rstd), the form fd_gemm_desc.ln_stats takes.
// NT = compile-time slab count (the loads of all slabs are issued together; a run-time trip count compiles to one
// dependent load / wait / add round trip per slab: 5 us for 8 slabs), 0 = any count.
template <int NT>
__global__ void k_ln_finalize(const float* __restrict__ parts, float* __restrict__ stats, int M, int nt, float inv_n, float eps) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float s1 = 0.f, s2 = 0.f;
    if constexpr (NT > 0) {
        floatx2 p[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t] = *reinterpret_cast<const floatx2*>(parts + 2 * ((size_t)t * M + m));
#pragma unroll
        for (int t = 0; t < NT; ++t) {   // fixed order: the same sums whatever the launch shape
            s1 += p[t][0];
            s2 += p[t][1];
        }
    } else {
        for (int t = 0; t < nt; ++t) {
            const floatx2 p = *reinterpret_cast<const floatx2*>(parts + 2 * ((size_t)t * M + m));
            s1 += p[0];
            s2 += p[1];
        }
    }
    const float mean = s1 * inv_n;
    const float var = fmaxf(fmaf(-mean, mean, s2 * inv_n), 0.f);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<floatx2*>(stats + 2 * (size_t)m) = floatx2{rstd, -mean * rstd};
}

extern "C" int fd_ln_finalize_stats_f32(const float* partials, float* stats, int M, int N, int n_tiles, float eps, void* stream) {
    FD_PLAN(fd_ln_finalize_stats_f32(partials, stats, M, N, n_tiles, eps, fd_s_));
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(2u, __LINE__));
    FD_CHECK_ARG(partials && stats && M > 0 && N > 0 && n_tiles >= 1 && n_tiles <= 64, FD_EINVAL, "fd_ln_finalize_stats_f32: args");
    const dim3 grid((M + 63) / 64), block(64);   // one wave per workgroup: 16384 rows spread over all 256 CUs
    const float inv_n = 1.0f / (float)N, e = eps > 0.f ? eps : 1e-5f;
    hipStream_t st = (hipStream_t)stream;
    switch (n_tiles) {
        case 2: hipLaunchKernelGGL(k_ln_finalize<2>, grid, block, 0, st, partials, stats, M, n_tiles, inv_n, e); break;
        case 3: hipLaunchKernelGGL(k_ln_finalize<3>, grid, block, 0, st, partials, stats, M, n_tiles, inv_n, e); break;
        case 4: hipLaunchKernelGGL(k_ln_finalize<4>, grid, block, 0, st, partials, stats, M, n_tiles, inv_n, e); break;
        case 8: hipLaunchKernelGGL(k_ln_finalize<8>, grid, block, 0, st, partials, stats, M, n_tiles, inv_n, e); break;
        default: hipLaunchKernelGGL(k_ln_finalize<0>, grid, block, 0, st, partials, stats, M, n_tiles, inv_n, e); break;
    }
    FD_CHECK_LAUNCH("k_ln_finalize");
    return FD_OK;
}

extern "C" int fd_softmax_rows_f16(void* x, int rows, int N, int ld, float scale, void* stream) {
    FD_PLAN(fd_softmax_rows_f16(x, rows, N, ld, scale, fd_s_));
    FdProfScope fd_prof_(FD_FAMILY_OTHER, stream, 0.0, fd_tag(2u, __LINE__));
    FD_CHECK_ARG(x && rows > 0 && N > 0, FD_EINVAL, "fd_softmax_rows_f16: args");
    FD_CHECK_ARG(N % 8 == 0 && ld % 8 == 0, FD_ESHAPE, "fd_softmax_rows_f16: N, ld %% 8");
    hipLaunchKernelGGL(k_softmax_rows, dim3(rows), dim3(256), 0, (hipStream_t)stream, (half_t*)x, N,
                       ld, scale);
    FD_CHECK_LAUNCH("k_softmax_rows");
    return FD_OK;
}
