// HBM write throughput by store shape: every lane stores 16 bytes; a wave instruction covers
// R rows x (1024/R) contiguous bytes of a row-major fp16 matrix [M][N] (the GEMM epilogue writes
// R = 16: 64-byte runs; a row-contiguous epilogue would write R = 4 or 2: 256 / 512-byte runs).
// Also a copy (read + write) with the same shapes.
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int R, bool COPY>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int M, int N16) {
    // N16 = row length in 16-byte units.  A wave handles a [16 rows][N] strip piece by piece.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int CW = 64 / R;                       // 16-byte units per row per instruction
    const int r = lane / CW, c = lane % CW;
    const int strips = M / R;
    for (int s = blockIdx.x * 4 + wave; s < strips; s += gridDim.x * 4) {
        const size_t row = (size_t)s * R + r;
        for (int c0 = 0; c0 < N16; c0 += CW) {
            u32x4 v = {(unsigned)s, (unsigned)c0, 1u, 2u};
            if (COPY) v = src[row * N16 + c0 + c];
            dst[row * N16 + c0 + c] = v;
        }
    }
}
template <int R, bool COPY>
static void run(const u32x4* src, u32x4* dst, int M, int N) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N16 = N * 2 / 16;
    hipLaunchKernelGGL((k<R, COPY>), dim3(2048), dim3(256), 0, 0, src, dst, M, N16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<R, COPY>), dim3(2048), dim3(256), 0, 0, src, dst, M, N16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double bytes = (double)M * N * 2 * (COPY ? 2 : 1);
    printf("M=%d N=%d rows/instr=%2d (%4d-byte runs) %s: %.1f us  %.2f TB/s\n", M, N, R, 1024 / R,
           COPY ? "copy " : "store", ms * 1e3, bytes / ms / 1e9);
}
// out = a + r: two streamed reads and one streamed write of [M][N] fp16 -- the HBM traffic mix of the residual GEMMs
// (attention / feed-forward output projections: A and the residual in, the sum out), the floor a GEMM of that shape
// with negligible MFMA time could reach.  16 bytes per lane, 4 independent loads in flight per lane.
__global__ __launch_bounds__(256) void k_r2w1(const u32x4* __restrict__ a, const u32x4* __restrict__ r, u32x4* __restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 x0 = a[i], x1 = a[i + stride], x2 = a[i + 2 * stride], x3 = a[i + 3 * stride];
        u32x4 y0 = r[i], y1 = r[i + stride], y2 = r[i + 2 * stride], y3 = r[i + 3 * stride];
        out[i] = x0 + y0; out[i + stride] = x1 + y1; out[i + 2 * stride] = x2 + y2; out[i + 3 * stride] = x3 + y3;
    }
    for (; i < n16; i += stride) out[i] = a[i] + r[i];
}
static void run_r2w1(const u32x4* a, const u32x4* r, u32x4* out, int M, int N, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n16 = (size_t)M * N * 2 / 16;
    hipLaunchKernelGGL(k_r2w1, dim3(blocks), dim3(256), 0, 0, a, r, out, n16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_r2w1, dim3(blocks), dim3(256), 0, 0, a, r, out, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("M=%d N=%d out = a + r (2 reads + 1 write), %d blocks: %.1f us  %.2f TB/s over %.0f MB\n", M, N, blocks, ms * 1e3,
           3.0 * M * N * 2 / ms / 1e9, 3.0 * M * N * 2 / 1e6);
}
int main() {
    const int M = 65536;
    u32x4 *src, *dst;
    hipMalloc(&src, (size_t)M * 1280 * 2); hipMalloc(&dst, (size_t)M * 1280 * 2);
    hipMemset(src, 1, (size_t)M * 1280 * 2);
    for (int N : {320, 1280}) {
        run<16, false>(src, dst, M, N); run<8, false>(src, dst, M, N); run<4, false>(src, dst, M, N); run<2, false>(src, dst, M, N);
        run<16, true>(src, dst, M, N); run<8, true>(src, dst, M, N); run<4, true>(src, dst, M, N); run<2, true>(src, dst, M, N);
    }
    u32x4* res;
    hipMalloc(&res, (size_t)M * 1280 * 2); hipMemset(res, 2, (size_t)M * 1280 * 2);
    for (int N : {320, 640, 1280})
        for (int blocks : {1024, 2048, 4096, 8192}) run_r2w1(src, res, dst, N == 320 ? M : (N == 640 ? 16384 : 4096) * 1 , N, blocks);
    return 0;
}
