// Where does the LDS-DMA GEMM loop lose MFMA issue slots?  The tile-13 inner loop (256x160,
// 16 waves, wave tile 32x80, BK=64) rebuilt feature by feature:
//   MODE 0: MFMAs only (operands loaded once)          MODE 1: + the 14 ds_read_b128 per K-tile
//   MODE 2: + s_barrier per K-tile                      MODE 3: + LDS-DMA of the next K-tile (L2-resident
//   source shared by all workgroups)                    MODE 4: the same from distinct HBM rows per workgroup
//   MODE 7: mode 5 + ~8 % of the lines streamed from HBM; NS = 3: three LDS stages (prefetch 2 ahead)
//   MODE 5 / 6: distinct rows per workgroup, footprint 3.4 MB / 13.6 MB per XCD (L2- / MALL-resident)
// hipcc --offload-arch=gfx950 -O3 gemm_loop.hip -o gemm_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int BM = 256, BN = 160, WM = 8, NW = 16, MI = 2, NI = 5, STAGE = (BM + BN) * 128;

template <int MODE, int NS = 2>
__global__ __launch_bounds__(1024) void k(const half_t* A, float* out, int nk, unsigned a_bytes, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    for (int i = tid; i < NS * STAGE / 4; i += 1024) {
        unsigned h = (i + blockIdx.x * 7919u) * 2654435761u;   // rnd: random fp16 in [-2, 2): realistic MFMA toggle rate / power
        const unsigned lo = 0x3800u | ((h >> 3) & 0x87ffu), hi = 0x3800u | ((h >> 17) & 0x87ffu);
        reinterpret_cast<unsigned*>(smem)[i] = rnd ? (lo | (hi << 16)) : 0x3c003c00u;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
    const int frag_a = (wm * 32 + fr) * 128, frag_b = BM * 128 + (wn * 80 + fr) * 128;
    const int sw0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, sw1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;
    const unsigned voff = (unsigned)(((MODE >= 4 ? blockIdx.x : 0) * 416 + wave * 8 + (lane >> 3)) * 5760 + (lane & 7) * 16);
    floatx4 acc[MI][NI];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0, 0, 0, 0};
    half8 fa[MI], fb[NI];
    for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8*>(smem + frag_a + i * 2048 + sw0);
    for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const half8*>(smem + frag_b + j * 2048 + sw0);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (MODE >= 3) {
            if (NS == 3) {   // K-tile kt has landed when at most one younger tile is in flight
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            char* stage = smem + (NS == 3 ? (cur == 0 ? 2 : cur - 1) : (cur ^ 1)) * STAGE;
            const int soff = (MODE == 5 || MODE == 7 ? (kt & 1) : MODE == 6 ? (kt & 7) : (kt & 31)) * 128;
#pragma unroll
            for (int i = 0; i < 3; ++i) {  // 52 groups of 8 rows over 16 waves = 3.25 per wave
                unsigned vo = voff + i * 16 * 8 * 5760;
                int so = soff;
                if (MODE == 7 && i == 2 && wave < 4) {   // ~8 % of the lines stream from HBM (never reused)
                    vo = voff + (unsigned)(256 * 416 * 5760) + (lane & 7) * 0;
                    so = (kt * 128) % 5632;
                    vo += (unsigned)((kt / 44) % 64) * 4u * 8u * 5760u * 256u / 64u;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(stage + (i * NW + wave) * 1024), 16, vo, so, 0, 0);
            }
        }
        const char* st = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
#pragma unroll
                for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 3 && NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE >= 2 && !(MODE >= 3 && NS == 3)) __syncthreads();
        if (MODE >= 1) cur = NS == 3 ? (cur == 2 ? 0 : cur + 1) : (cur ^ 1);
    }
    float s = 0;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NI; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 1024 + tid] = s;
}

template <int MODE, int NS = 2>
void run(const half_t* A, float* out, unsigned a_bytes, int rnd) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nk = 4000;
    hipLaunchKernelGGL((k<MODE, NS>), dim3(256), dim3(1024), NS * STAGE, 0, A, out, 10, a_bytes, rnd);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NS>), dim3(256), dim3(1024), NS * STAGE, 0, A, out, nk, a_bytes, rnd);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 256.0 * nk * 2.0 * BM * BN * 64;
    printf("%s mode %d NS %d: %.2f ms  %.0f TFLOP/s  (%.0f cycles per K-tile at 2.4 GHz)\n", rnd ? "random" : "const ", MODE, NS, ms, fl / ms / 1e9, ms * 1e-3 * 2.4e9 / nk);
}
__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        p[i] = (0x3800u | ((h >> 3) & 0x87ffu)) | ((0x3800u | ((h >> 17) & 0x87ffu)) << 16);
    }
}
int main() {
    const size_t bytes = (size_t)2 * 256 * 416 * 5760 + (1 << 20);
    half_t* A; hipMalloc(&A, bytes); hipMemset(A, 0, bytes);
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int rnd = 0; rnd < 2; ++rnd) {
        if (rnd) { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (unsigned*)A, bytes / 4); hipDeviceSynchronize(); }
        const unsigned b = (unsigned)bytes;
        run<0>(A, out, b, rnd); run<1>(A, out, b, rnd); run<2>(A, out, b, rnd); run<3>(A, out, b, rnd);
        run<4>(A, out, b, rnd); run<5>(A, out, b, rnd); run<6>(A, out, b, rnd);
        run<7>(A, out, b, rnd); run<7, 3>(A, out, b, rnd); run<5, 3>(A, out, b, rnd); run<6, 3>(A, out, b, rnd);
    }
    return 0;
}
