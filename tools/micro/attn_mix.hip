// Would v_mfma_f32_32x32x16_f16 pay in the head-dim-40 self-attention tile loop?  
// One iteration = the MFMA + VALU instruction mix of ONE 64-key K/V tile for 32 queries of one wave of
// k_attention_w8q2 (registers only: no LDS, no loads; operands are random fp16 so the power / clock effect of
// real toggle rates is included), 16 waves per CU (4 per SIMD) like the product kernel at 128 VGPRs:
//   MODE 0  product form:  S^T = K Q^T as 16 x v_mfma_16x16x32 (2 query blocks x 4 key blocks x 2 k-steps, head dim 40
//           padded to 64), 32 v_exp_f32 + 16 v_cvt_pk + 16 v_max3 per lane, PV as 12 x v_mfma_16x16x32 (3 d-tiles of 16)
//   MODE 1  QK^T on 32x32x16: 6 x v_mfma_32x32x16 (2 key blocks of 32 x 3 k-steps of 16: head dim padded to 48), same
//           VALU, + 8 v_permlane16_swap to re-pair P for the 16x16x32 PV form, PV 12 x v_mfma_16x16x32
//   MODE 2  everything on 32x32x16: QK^T 6, PV 8 (d padded to 64 = 2 row tiles x 4 key steps of 16), no swaps
//   MODE 3  MFMAs of MODE 0 only;  MODE 4  MFMAs of MODE 1 only;  MODE 5  the VALU part only
// Exponent inputs are kept in a harmless range; every result feeds the final checksum.
// hipcc --offload-arch=gfx950 -O3 attn_mix.hip -o attn_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ half8 rnd8(unsigned seed) {
    half8 v;
    for (int i = 0; i < 8; ++i) v[i] = (half_t)(((int)(hash(seed * 8 + i) & 0xffff) - 32768) * (1.0f / 32768.f));
    return v;
}

template <int MODE, int NT = 1024, int SYNC = 0>
__global__ __launch_bounds__(NT) void k(float* out, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned tid = blockIdx.x * NT + threadIdx.x;
    // few distinct fragment registers (the product kernel re-reads K / V^T fragments from LDS, they are transient there)
    half8 kf[4], qf[4], vf[3];
    for (int i = 0; i < 4; ++i) kf[i] = rnd8(tid * 31 + i);
    for (int i = 0; i < 4; ++i) qf[i] = rnd8(tid * 37 + 100 + i);
    for (int i = 0; i < 3; ++i) vf[i] = rnd8(tid * 41 + 200 + i);
    floatx4 o[2][3];
    for (int t = 0; t < 2; ++t) for (int d = 0; d < 3; ++d) o[t][d] = floatx4{0, 0, 0, 0};
    floatx16 o32[2];
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) o32[d][r] = 0.f;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        // SYNC 1: one workgroup barrier per tile, as the product kernel has for its shared K / V^T stage: the waves of a
        // workgroup then run every tile in the SAME phase (all in QK^T, all in the softmax, all in PV)
        if (SYNC == 1) __syncthreads();
        // SYNC 2: the same barrier, but the odd waves of each SIMD pair enter the loop half a tile late (their first
        // iteration skips the QK^T + softmax half), so that a SIMD always has one wave in MFMAs while the other does VALU
        float sc[32];   // the 32 scores a lane owns per 64-key tile: 2 query blocks x 16, or 1 query column x 2 x 16
        if (MODE == 0 || MODE == 3) {
            floatx4 s[2][4];
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[f], qf[2 * t], floatx4{-1.f, -1.f, -1.f, -1.f}, 0, 0, 0);
                    s[t][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[(f + 1) & 3], qf[2 * t + 1], s[t][f], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sc[(t * 4 + f) * 4 + r] = s[t][f][r];
        } else if (MODE == 1 || MODE == 2 || MODE == 4) {
            floatx16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                floatx16 a;
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = -1.f;
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb], qf[0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb + 1], qf[1], a, 0, 0, 0);
                s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kb + 2], qf[2], a, 0, 0, 0);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[kb * 16 + r] = s[kb][r];
        } else {
#pragma unroll
            for (int i = 0; i < 32; ++i) sc[i] = -0.01f * i - 0.001f * (threadIdx.x & 63) + keep * 1e-9f;
        }
        half8 p[4];
        if (MODE != 3 && MODE != 4) {
            // lane-local max (v_max3 chain), exp2, packed convert: the per-score VALU work of the product kernel
            float tmax = sc[0];
#pragma unroll
            for (int i = 1; i + 1 < 32; i += 2) asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(tmax), "v"(sc[i]), "v"(sc[i + 1]));
            keep += tmax * 1e-6f;
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const floatx2 e = {__builtin_amdgcn_exp2f(sc[i]), __builtin_amdgcn_exp2f(sc[i + 1])};
                const half2v eh = __builtin_convertvector(e, half2v);
                p[i >> 3][(i & 7)] = eh[0];
                p[i >> 3][(i & 7) + 1] = eh[1];
            }
            if (MODE == 1) {   // re-pair P from the 32-column layout to the two 16-column B operands: 8 swaps of packed pairs
#pragma unroll
                for (int i = 0; i < 4; i += 2)
#pragma unroll
                    for (int h = 0; h < 8; h += 2) {
                        unsigned x = __builtin_bit_cast(unsigned, half2v{p[i][h], p[i][h + 1]});
                        unsigned y = __builtin_bit_cast(unsigned, half2v{p[i + 1][h], p[i + 1][h + 1]});
                        auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
                        const half2v a = __builtin_bit_cast(half2v, (unsigned)r[0]), b = __builtin_bit_cast(half2v, (unsigned)r[1]);
                        p[i][h] = a[0]; p[i][h + 1] = a[1]; p[i + 1][h] = b[0]; p[i + 1][h + 1] = b[1];
                    }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) p[i][j] = (half_t)(sc[i * 8 + j] * 0.01f);
        }
        if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 4; ++i) keep += (float)p[i][0] + (float)p[i][7];
        } else if (MODE == 2) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) o32[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[(dt + ks) % 3], p[ks], o32[dt], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dt = 0; dt < 3; ++dt)
#pragma unroll
                    for (int kg = 0; kg < 2; ++kg) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[(dt + kg) % 3], p[t * 2 + kg], o[t][dt], 0, 0, 0);
        }
        // keep the accumulators bounded without touching the instruction mix much
        if ((it & 63) == 63) {
            for (int t = 0; t < 2; ++t) for (int d = 0; d < 3; ++d) o[t][d] *= 1e-3f;
            for (int d = 0; d < 2; ++d) o32[d] *= 1e-3f;
        }
    }
    float sum = keep;
    for (int t = 0; t < 2; ++t) for (int d = 0; d < 3; ++d) sum += o[t][d][0] + o[t][d][1] + o[t][d][2] + o[t][d][3];
    for (int d = 0; d < 2; ++d) for (int r = 0; r < 16; ++r) sum += o32[d][r];
    out[tid] = sum;
#endif
}

template <int MODE, int NT = 1024, int SYNC = 0>
void run(float* out, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;   // 64 tiles x 64 query-block passes: the order of one level-0 launch per wave
    const int grid = 256 * (1024 / NT);
    hipLaunchKernelGGL((k<MODE, NT, SYNC>), dim3(grid), dim3(NT), 0, 0, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NT, SYNC>), dim3(grid), dim3(NT), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d nt %4d sync %d %-64s %8.3f ms  %7.1f ns / tile-iteration\n", MODE, NT, SYNC, what, ms, 1e6 * ms / iters);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, "product mix: QK 16 x m16x16x32, softmax VALU, PV 12 x m16x16x32");
        run<1>(out, "QK 6 x m32x32x16, softmax VALU + 8 permlane16_swap, PV 12 x m16x16x32");
        run<2>(out, "QK 6 x m32x32x16, softmax VALU, PV 8 x m32x32x16");
        run<3>(out, "MFMAs of mode 0 only");
        run<4>(out, "MFMAs of mode 1 only");
        run<5>(out, "softmax VALU only");
        // does the per-tile workgroup barrier (lockstep phases) expose the softmax VALU?
        run<0, 1024, 1>(out, "product mix, 16-wave workgroups, barrier per tile");
        run<0, 512, 0>(out, "product mix, 2 x 8-wave workgroups per CU, no barrier");
        run<0, 512, 1>(out, "product mix, 2 x 8-wave workgroups per CU, barrier per tile");
        run<0, 256, 1>(out, "product mix, 4 x 4-wave workgroups per CU, barrier per tile");
        run<3, 512, 1>(out, "MFMAs only, 2 x 8-wave workgroups per CU, barrier per tile");
    }
    return 0;
}
