// Is v_mfma_f32_16x16x16_f16 half the cost of v_mfma_f32_16x16x32_f16 on gfx950?  If so, a
// head_dim-40 QK^T can contract over 32 + 16 = 48 instead of 64 padded columns.
// 256 workgroups x 1024 threads, 8 independent accumulators per wave, registers only.
// hipcc --offload-arch=gfx950 -O3 mfma_k16.hip -o mfma_k16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: 8x k32   1: 8x k16   2: 8x (k32 + k16)
__global__ __launch_bounds__(1024) void k(float* out, int n) {
    const int lane = threadIdx.x & 63;
    half8 a8, b8; half4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (half_t)(0.01f * ((lane * 7 + i) % 13)); b8[i] = (half_t)(0.02f * ((lane * 5 + i) % 11)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            if (MODE == 1 || MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE>
void run(float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 4 waves per SIMD, 8 (or 16) MFMAs per iteration each
    const double per_simd = 4.0 * n * 8;
    printf("mode %d: %.3f ms  -> %.1f cycles per loop step (one k32, one k16, or the pair) per SIMD at 2.4 GHz\n", MODE, ms,
           ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) { run<0>(out); run<1>(out); run<2>(out); }
    return 0;
}
