// Slab GroupNorm(+SiLU): the body shared by k_gn_slab (norm.hip: the slab is read from an fp16 activation) and
// k_splitk_finish_gn (gemm.hip: the slab is PRODUCED in registers by the split-K finish -- fp32 partial slabs summed in a fixed
// order + bias + per-sample bias (+ residual), rounded to fp16 -- so the convolution's output never makes the round trip through
// HBM between the finish launch and the GroupNorm launch it used to feed).  Same thread -> (pixel, channel) map, same reduction
// order: for the same <NT, NV> and GB the two kernels produce the same bits.  Device code; include once per translation unit.
#pragma once
#include "common.h"

__device__ __forceinline__ float gn_act(float f, int silu) {
    // SiLU with the hardware reciprocal (1 ulp; the result is rounded to fp16 anyway)
    return silu ? f * __builtin_amdgcn_rcpf(1.0f + __expf(-f)) : f;
}


// A workgroup of NT threads owns one sample x GB consecutive groups; its [HW][GB*cpg] slab lives in registers (thread t keeps
// the fixed 8-channel chunk t % c8 of pixels t / c8, +PL, ... -- at most NV 16-byte vectors = 8 halfs), statistics are reduced
// through LDS in a fixed order, then the same registers are normalised and stored.
//   fill(v) -> v[i] = the 8 fp16 values (packed) of pixel p = pl + PL * i at this thread's channel chunk, zeros for p >= HW and
//              for the threads beyond the last pixel lane (pl >= PL)
//   yb: output base of (sample, first channel of the block); gamma / beta already offset to the block's first channel
template <int NT, int NV, typename Fill>
__device__ __forceinline__ void gn_slab_body(Fill fill, half_t* __restrict__ yb, const float* __restrict__ gamma,
                                             const float* __restrict__ beta, int HW, int C, int cpg, int GB, float eps,
                                             int silu, float* sm) {
    const int CB = cpg * GB, c8 = CB >> 3;
    const int PL = NT / c8;
    const int J = NT / (CB >> 1);     // second-level partial rows
    float* part = sm;                 // [PL][CB/2][2]  (sum, sumsq) per channel pair
    float* part2 = sm + PL * CB;      // [J][CB/2][2]
    float* gst = part2 + J * CB;      // [GB][2] mean, rstd
    const int tid = threadIdx.x;
    const int cc = tid % c8, pl = tid / c8;
    const bool active = pl < PL;
    const unsigned off0 = (unsigned)(pl * C + cc * 8), ostep = (unsigned)(PL * C);
    // statistics per channel PAIR with v_dot2_f32_f16 (packed fp16 in, fp32 accumulate): cpg is
    // even (checked by the launcher), so a pair never straddles two groups
    uint4 v[NV];
    float s[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = q[k] = 0.f;
    fill(v);
    const half2v one2 = {(half_t)1.0f, (half_t)1.0f};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const unsigned w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};   // zeros beyond HW add nothing
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half2v h = __builtin_bit_cast(half2v, w[k]);
            s[k] = __builtin_amdgcn_fdot2(h, one2, s[k], false);
            q[k] = __builtin_amdgcn_fdot2(h, h, q[k], false);
        }
    }
    const int CP = CB >> 1;           // channel pairs
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            part[(pl * CP + cc * 4 + k) * 2 + 0] = s[k];
            part[(pl * CP + cc * 4 + k) * 2 + 1] = q[k];
        }
    }
    __syncthreads();
    {   // level 1: thread (j, c) sums pixel lanes j, j+J, ... of channel pair c
        const int c = tid % CP, j = tid / CP;
        if (j < J) {
            float a = 0.f, bq = 0.f;
            for (int l = j; l < PL; l += J) {
                a += part[(l * CP + c) * 2 + 0];
                bq += part[(l * CP + c) * 2 + 1];
            }
            part2[(j * CP + c) * 2 + 0] = a;
            part2[(j * CP + c) * 2 + 1] = bq;
        }
    }
    __syncthreads();
    {   // level 2: one wavefront per group sums its J x cpg values (fp64, fixed order)
        const int lane = tid & 63, wave = tid >> 6;
        for (int g = wave; g < GB; g += NT / 64) {
            double a = 0.0, bq = 0.0;
            const int ppg = cpg >> 1;   // pairs per group
            for (int i = lane; i < J * ppg; i += 64) {
                const int j = i / ppg, c = g * ppg + (i - j * ppg);
                a += (double)part2[(j * CP + c) * 2 + 0];
                bq += (double)part2[(j * CP + c) * 2 + 1];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                bq += __shfl_xor(bq, o, 64);
            }
            if (lane == 0) {
                const double n = (double)HW * cpg;
                const double mean = a / n;
                double var = bq / n - mean * mean;
                if (var < 0.0) var = 0.0;
                gst[g * 2 + 0] = (float)mean;
                gst[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
            }
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cc * 8 + k, g = c / cpg;
        const float a = gst[g * 2 + 1] * gamma[c];
        sc[k] = a;
        sh[k] = beta[c] - gst[g * 2 + 0] * a;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int p = pl + PL * i;
        if (p >= HW) break;
        const half8 h = *reinterpret_cast<const half8*>(&v[i]);
        half8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (half_t)gn_act(fmaf((float)h[k], sc[k], sh[k]), silu);
        *reinterpret_cast<uint4*>(yb + (off0 + ostep * i)) = *reinterpret_cast<uint4*>(&o);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The launch shape of a slab kernel <NT, NV> for a [B][HW][C] map with G groups: the number of groups per workgroup (the first
// of 1, 2, 4, 8 whose slab fits the kernel's registers and LDS), 0 = this kernel cannot run the shape.  *lds = its dynamic LDS.
template <int NT, int NV>
static inline int gn_slab_pick(int HW, int C, int G, size_t* lds) {
    const int cpg = C / G;
    if (C % G || (cpg & 1)) return 0;
    for (int GB = 1; GB <= 8 && GB <= G; GB *= 2) {
        if (G % GB || (cpg * GB) % 8) continue;
        const int CB = cpg * GB, cb8 = CB / 8;
        if (cb8 > NT || CB / 2 > NT) break;
        const int pl = NT / cb8, J = NT / (CB / 2);
        if ((HW + pl - 1) / pl > NV) break;
        const size_t bytes = ((size_t)pl * CB + (size_t)J * CB + GB * 2) * sizeof(float);
        if (bytes > 160 * 1024) break;
        *lds = bytes;
        return GB;
    }
    return 0;
}
