// Pure-MFMA ceiling on this GPU: W waves per CU issuing independent v_mfma_f32_16x16x32_f16
// from registers (no LDS, no memory).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float* out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    floatx4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 16 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep)
    for (int waves : {4, 8, 16}) {
        for (int blocks : {256, 512}) {
            hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(64 * waves), 0, 0, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * waves * iters * 10 * 16 * 16 * 32 * 2;
            printf("waves/block %2d blocks %3d: %.2f ms  %.0f TFLOP/s\n", waves, blocks, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
