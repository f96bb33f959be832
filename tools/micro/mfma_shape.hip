// Which MFMA shape costs less energy per FLOP on MI355X with realistic (random fp16) operands?
// The product GEMM loop is power-limited (profiles/r01_micro_gemm_loop.txt): with random data the
// chip clocks down, so instructions-per-FLOP and register-file reads per FLOP matter.  Same wave
// tile (64 x 64 per wave, 16 waves on a 256 x 256 block, BK = 64), same LDS bytes per FLOP, only
// the instruction differs:
//   SHAPE 16: v_mfma_f32_16x16x32_f16   16 MFMAs + 8 ds_read_b128 per k32 step
//   SHAPE 32: v_mfma_f32_32x32x16_f16    8 MFMAs + 8 ds_read_b128 per k32 step
// MODE 0: MFMAs only (fragments loaded once), MODE 1: + fragment reads from LDS every K-tile,
// MODE 2: + one s_barrier per K-tile.
// hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 256, BN = 256, STAGE = (BM + BN) * 128;

template <int SHAPE, int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int nk, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    for (int i = tid; i < 2 * STAGE / 4; i += 1024) {
        unsigned h = (i + blockIdx.x * 7919u) * 2654435761u;
        const unsigned lo = 0x3800u | ((h >> 3) & 0x87ffu), hi = 0x3800u | ((h >> 17) & 0x87ffu);
        reinterpret_cast<unsigned*>(smem)[i] = rnd ? (lo | (hi << 16)) : 0x3c003c00u;
    }
    __syncthreads();
    float s = 0;
    int cur = 0;
    if constexpr (SHAPE == 16 || SHAPE == 17) {
        const int fr = lane & 15, fq = lane >> 4;
        const int frag_a = (wm * 64 + fr) * 128, frag_b = BM * 128 + (wn * 64 + fr) * 128;
        floatx4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0, 0, 0, 0};
        half8 fa[4], fb[4];
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const half8*>(smem + frag_a + i * 2048 + ((fq ^ (fr & 7)) << 4));
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const half8*>(smem + frag_b + j * 2048 + ((fq ^ (fr & 7)) << 4));
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + cur * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int sw = ((ks * 4 + fq) ^ (fr & 7)) << 4;
                if (MODE >= 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if constexpr (SHAPE == 17)   // same bits reinterpreted as bf16: same LDS traffic, bf16 multipliers
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            if (MODE >= 2) __syncthreads();
            if (MODE >= 1) cur ^= 1;
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        // 32x32x16: lane supplies row (lane & 31), k = (lane >> 5) * 8 .. + 7 of a 16-deep step.
        // LDS rows of 128 B (8 chunks); slot = chunk ^ ((row >> 1) & 7): conflict-free ds_read_b128
        const int fr = lane & 31, fh = lane >> 5;
        const int frag_a = (wm * 64 + fr) * 128, frag_b = BM * 128 + (wn * 64 + fr) * 128;
        const int rs = (fr >> 1) & 7;
        floatx16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        half8 fa[2], fb[2];
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const half8*>(smem + frag_a + i * 4096 + ((fh ^ rs) << 4));
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const half8*>(smem + frag_b + j * 4096 + ((fh ^ rs) << 4));
        for (int kt = 0; kt < nk; ++kt) {
            const char* st = smem + cur * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int sw = ((ks * 2 + fh) ^ rs) << 4;
                if (MODE >= 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 4096 + sw);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 4096 + sw);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
            if (MODE >= 2) __syncthreads();
            if (MODE >= 1) cur ^= 1;
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 1024 + tid] = s;
}

template <int SHAPE, int MODE>
void run(float* out, int rnd) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nk = 4000;
    hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(1024), 2 * STAGE, 0, out, 10, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(256), dim3(1024), 2 * STAGE, 0, out, nk, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * nk * 2.0 * BM * BN * 64;
    printf("%s shape %2d mode %d: %.2f ms  %.0f TFLOP/s  (%.0f cycles per K-tile at 2.4 GHz)\n", rnd ? "random" : "const ",
           SHAPE, MODE, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / nk);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int rnd = 0; rnd < 2; ++rnd) {
            run<16, 0>(out, rnd); run<17, 0>(out, rnd); run<32, 0>(out, rnd);
            run<16, 1>(out, rnd); run<17, 1>(out, rnd); run<32, 1>(out, rnd);
            run<16, 2>(out, rnd); run<32, 2>(out, rnd);
        }
    return 0;
}
