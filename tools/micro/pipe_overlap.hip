// Do the matrix pipe and the transcendental VALU ops of one SIMD overlap?  Each CU runs 16
// waves (4 per SIMD).  MODE 0: every wave issues NM independent v_mfma_f32_16x16x32_f16 per
// iteration; MODE 1: every wave issues NE v_exp_f32; MODE 2: waves alternate roles per SIMD (2 MFMA
// waves + 2 exp waves on each SIMD); MODE 3: every wave does BOTH in program order (MFMA block
// then exp block, like an attention tile).  If the pipes overlap, MODE 2 ~ max(MODE 0, MODE 1)/2
// of the summed work; if they serialise, ~ their sum.  Also v_fma_f32 (full-rate VALU) as MODE 4/5.
// hipcc --offload-arch=gfx950 -O3 pipe_overlap.hip -o pipe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int NM = 14, NE = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(((threadIdx.x * 7 + i * 13) % 31) * 0.06f - 0.9f); b[i] = (_Float16)(((threadIdx.x * 3 + i * 5) % 29) * 0.07f - 1.0f); }
    floatx4 acc[NM];
    for (int i = 0; i < NM; ++i) acc[i] = floatx4{0, 0, 0, 0};
    float e[NE];
    for (int i = 0; i < NE; ++i) e[i] = -0.001f * (threadIdx.x % 64) - 0.01f * i;
    // waves 0..3 land on SIMD 0..3 (cyclic), so wave >> 2 alternates roles within each SIMD
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && ((wave >> 2) & 1) == 0) || (MODE == 5 && ((wave >> 2) & 1) == 0);
    const bool do_e = MODE == 1 || MODE == 3 || (MODE == 2 && ((wave >> 2) & 1) == 1);
    const bool do_f = MODE == 4 || (MODE == 5 && ((wave >> 2) & 1) == 1);
    const bool do_m32 = MODE == 6 || MODE == 8 || (MODE == 7 && ((wave >> 2) & 1) == 0);
    const bool do_e2 = MODE == 8 || (MODE == 7 && ((wave >> 2) & 1) == 1);
    floatx16 acc32[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();   // s_memtime: shader clock ticks
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        if (do_e) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.0009765625f;
        }
        if (do_m32) {
#pragma unroll
            for (int i = 0; i < 7; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i & 3], 0, 0, 0);
        }
        if (do_e2) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]) - 1.0009765625f;
        }
        if (MODE == 9) {   // one wave, fine interleave: 1 MFMA 32x32x16 then 2 exp + 2 sub, 7 times (+2 exp)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc32[i & 3], 0, 0, 0);
                e[2 * i] = __builtin_amdgcn_exp2f(e[2 * i]) - 1.0009765625f;
                e[2 * i + 1] = __builtin_amdgcn_exp2f(e[2 * i + 1]) - 1.0009765625f;
            }
            e[14] = __builtin_amdgcn_exp2f(e[14]) - 1.0009765625f;
            e[15] = __builtin_amdgcn_exp2f(e[15]) - 1.0009765625f;
        }
        if (MODE == 10) {  // the same with 16x16x32: 2 MFMA then 2 exp
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[2 * i + 1], 0, 0, 0);
                e[2 * i] = __builtin_amdgcn_exp2f(e[2 * i]) - 1.0009765625f;
                e[2 * i + 1] = __builtin_amdgcn_exp2f(e[2 * i + 1]) - 1.0009765625f;
            }
            e[14] = __builtin_amdgcn_exp2f(e[14]) - 1.0009765625f;
            e[15] = __builtin_amdgcn_exp2f(e[15]) - 1.0009765625f;
        }
        if (do_f) {
#pragma unroll
            for (int i = 0; i < NE; ++i) e[i] = fmaf(e[i], 0.999f, -0.001f);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = c1 - c0;
    float s = 0;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < NE; ++i) s += e[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc32[i][r];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* what) {
    static unsigned long long* cyc = nullptr;
    if (!cyc) hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d %-46s %.2f ms  %.1f shader clocks / iteration  (clock %.2f GHz)\n", MODE, what, ms, (double)c / iters, c / (ms * 1e6));
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, "16 waves: 14 MFMA / iter");
        run<1>(out, "16 waves: 16 v_exp_f32 (+1 sub) / iter");
        run<2>(out, "8 waves MFMA + 8 waves exp (2+2 per SIMD)");
        run<3>(out, "16 waves: MFMA block then exp block");
        run<4>(out, "16 waves: 16 v_fma_f32 / iter");
        run<5>(out, "8 waves MFMA + 8 waves fma");
        run<6>(out, "16 waves: 7 MFMA 32x32x16 / iter");
        run<7>(out, "8 waves MFMA32 + 8 waves exp");
        run<8>(out, "16 waves: MFMA32 block then exp block");
        run<9>(out, "16 waves: MFMA32 / exp finely interleaved");
        run<10>(out, "16 waves: MFMA16 / exp finely interleaved");
    }
    return 0;
}
