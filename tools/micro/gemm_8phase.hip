// Deep-pipelined fp16 GEMM main loop for gfx950: 256x256x64 block tile, 8 wavefronts (2 along M x 4 along N), 128x64
// wave tiles, 128 KiB of LDS as a ring of eight 16-KiB pieces (two K-tiles x {A rows 0-63 of each wave row, W rows 0-31 of
// each wave column, W rows 32-63, A rows 64-127}), every piece filled by LDS-DMA (buffer_load_dwordx4 ... lds) and kept in
// flight ACROSS barriers with counted s_waitcnt vmcnt(8) -- four pieces (one K-tile) are always on their way, nothing in
// the loop ever drains the queue.  A K-tile is four phases, one 64x32 quadrant of the wave tile each (16 MFMA 16x16x32):
//   phase 0: ds_read A0 (8 x b128) + W0 (4)   quadrant (0,0)        phase 1: ds_read W1 (4)   quadrant (0,1)
//   phase 2: ds_read A1 (8)                   quadrant (1,1)        phase 3: (W0 still in registers)   quadrant (1,0)
// and every phase stages one piece, six pieces ahead of the one it reads.  A phase is  [reads + DMA issue + vmcnt] s_barrier
// [16 MFMAs under s_setprio 1] s_barrier.  The two wave rows run ONE BARRIER APART (wave row 1 executes an extra s_barrier
// before the loop, wave row 0 after it), so on every SIMD one wave is in its MFMA section while its partner issues the reads
// and DMAs of its next section: the matrix pipe never sees both waves asking at once and never none.
//
// Ordering rules the schedule obeys (MI355X_MICROARCH.md, two waves per SIMD, item 7):
//   RAW  a piece is read one phase AFTER the phase whose vmcnt(8) retired it in every wave (wait, barrier, [staggered rows:
//        one more barrier], read);  piece q is staged in phase q - 6 and first read in phase >= q - 2.
//   WAR  piece q overwrites piece q - 8, whose last ds_read was issued >= 2 phases before (phase of the read + 2 <= q - 6).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_8phase.hip -o gemm_8phase
// Run:    ./gemm_8phase            refcheck 256/512 vs fp64 host, 4096 vs a naive device GEMM, 20-run race screen, timing
//                                  of the 8-phase loop and of the product's 2-barrier loop (16 waves, 64x64 wave tiles,
//                                  vmcnt(0) + __syncthreads per K-tile) on the same operands, interleaved.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PIECE = 128 * 128;   // bytes: 128 rows x 64 halfs

__device__ __forceinline__ void tile_of_block(int id, int tiles_m, int tiles_n, int remap, int& tile_m, int& tile_n) {
    const int nb = tiles_m * tiles_n;
    if (remap == 2 && tiles_m % 4 == 0 && tiles_n % 8 == 0) {
        // each XCD (blocks id % 8) owns 4 x 8 blocks of tiles: 4 A panels + 8 W panels per 32 tiles
        const int xcd = id & 7, slot = id >> 3;
        const int per_row = tiles_n / 8;                 // 4x8 super-blocks per super-row
        const int sb = xcd + 8 * (slot / 32);            // super-block index
        const int s = slot % 32;
        tile_m = (sb / per_row) * 4 + s / 8;
        tile_n = (sb % per_row) * 8 + s % 8;
        return;
    }
    if (remap) {
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    tile_n = id % tiles_n;
    tile_m = id / tiles_n;
}

// VAR bit 0: s_setprio around the MFMA clusters; bit 1: wave rows one barrier apart; bit 2: vmcnt(0) instead of vmcnt(8)
// bit 3: per-phase s_memtime stamps (LOAD start, LOAD end, MFMA-issue end) of the last main-loop K-tile, written to g_stamps
__device__ unsigned long long g_stamps[256 * 8 * 12];
template <int VAR>
__global__ __launch_bounds__(512) void k_gemm_8phase(const half_t* __restrict__ A, const half_t* __restrict__ W, half_t* __restrict__ C,
                                                      int M, int N, int K, int tiles_m, int tiles_n, int remap) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool PRIO = VAR & 1, STAG = VAR & 2, DRAIN = VAR & 4, TIMING = VAR & 8;
    unsigned long long ts_a[4] = {0, 0, 0, 0}, ts_b[4] = {0, 0, 0, 0}, ts_d[4] = {0, 0, 0, 0};
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    int tile_m, tile_n;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, remap, tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((size_t)N * K * 2), 0x00020000);

    // ---- LDS-DMA lane state: a wave instruction fills 8 rows x 128 B; lane slot p of row r holds source chunk p ^ (r & 7)
    const int rsub = lane >> 3, ck = (lane & 7) ^ rsub;
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        // piece row r = wave * 16 + u * 8 + rsub;  A piece: wave row r >> 6, local row r & 63;  W piece: wave column r >> 5, r & 31
        const int arow = m0 + (wave >> 2) * 128 + (wave & 3) * 16 + u * 8 + rsub;
        const int brow = n0 + (wave >> 1) * 64 + (wave & 1) * 16 + u * 8 + rsub;
        a_voff[u] = (unsigned)(arow * K + ck * 8) * 2u;
        b_voff[u] = (unsigned)(brow * K + ck * 8) * 2u;
    }
    const int a_half = 64 * K * 2, b_half = 32 * K * 2;   // second A / W piece of a K-tile: 64 / 32 rows further
    const int nk = K >> 6;
    char* const dma_dst = smem + wave * 2048;

    // ---- fragment read state (16 lanes x 4 k-chunks; row r keeps chunk c at slot c ^ (r & 7): conflict-free b128 reads)
    const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    const char* const pa0 = smem + (wm * 64 + fr) * 128 + sw0;
    const char* const pa1 = smem + (wm * 64 + fr) * 128 + sw1;
    const char* const pb0 = smem + (wn * 32 + fr) * 128 + sw0;
    const char* const pb1 = smem + (wn * 32 + fr) * 128 + sw1;

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 fa[4][2], fb0[2][2], fb1[2][2];

    // stage piece q = 4 t + o (o: 0 A rows 0-63, 1 W rows 0-31, 2 W rows 32-63, 3 A rows 64-127) into ring slot q & 7
#define STAGE(T, O, SLOT, CHK)                                                                                         \
    {                                                                                                                  \
        const bool ok_ = !(CHK) || (T) < nk;                                                                           \
        const int so_ = (T) * 128 + ((O) == 3 ? a_half : (O) == 2 ? b_half : 0);                                       \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                \
            if ((O) == 0 || (O) == 3)                                                                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(dma_dst + (SLOT) * PIECE + u * 1024), 16,      \
                                                         ok_ ? a_voff[u] : 0x80000000u, so_, 0, 0);                    \
            else                                                                                                       \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(dma_dst + (SLOT) * PIECE + u * 1024), 16,      \
                                                         ok_ ? b_voff[u] : 0x80000000u, so_, 0, 0);                    \
        }                                                                                                              \
    }
#define READ_A(SLOT)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                    \
        fa[i][0] = *reinterpret_cast<const half8*>(pa0 + (SLOT) * PIECE + i * 2048);                                   \
        fa[i][1] = *reinterpret_cast<const half8*>(pa1 + (SLOT) * PIECE + i * 2048);                                   \
    }
#define READ_B(FB, SLOT)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                    \
        FB[j][0] = *reinterpret_cast<const half8*>(pb0 + (SLOT) * PIECE + j * 2048);                                   \
        FB[j][1] = *reinterpret_cast<const half8*>(pb1 + (SLOT) * PIECE + j * 2048);                                   \
    }
#define MFMA_Q(MI, NI, FB)                                                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                              \
                acc[(MI) * 4 + i][(NI) * 2 + j] =                                                                      \
                    __builtin_amdgcn_mfma_f32_16x16x32_f16(FB[j][ks], fa[i][ks], acc[(MI) * 4 + i][(NI) * 2 + j], 0, 0, 0);
#define BAR()                                   \
    __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_barrier();               \
    __builtin_amdgcn_sched_barrier(0);
#define WAIT_STAGED()                                                        \
    if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              \
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // phase P (0..7 inside a pair of K-tiles kt, kt + 1): parity b = P >> 2 reads ring slots 4 b .. 4 b + 3, stages piece P + 6
#define PHASE(P, CHK)                                                                                                  \
    {                                                                                                                  \
        constexpr int b_ = (P) >> 2, ph_ = (P) & 3;                                                                    \
        if (TIMING && !(CHK)) ts_a[ph_] = __builtin_amdgcn_s_memtime();                                                          \
        if (ph_ == 0) { READ_B(fb0, 4 * b_ + 1); __builtin_amdgcn_sched_barrier(0); READ_A(4 * b_ + 0); }              \
        if (ph_ == 1) { READ_B(fb1, 4 * b_ + 2); }                                                                     \
        if (ph_ == 2) { READ_A(4 * b_ + 3); }                                                                          \
        STAGE(kt + (((P) + 6) >> 2), ((P) + 6) & 3, ((P) + 6) & 7, CHK);                                               \
        WAIT_STAGED(); /* every phase: phase 3 retires the W0 piece that phase 0 of the next K-tile reads first */        \
        if (TIMING && !(CHK)) ts_b[ph_] = __builtin_amdgcn_s_memtime();                                                          \
        BAR();                                                                                                         \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                                       \
        if (ph_ == 0) { MFMA_Q(0, 0, fb0); }                                                                           \
        if (ph_ == 1) { MFMA_Q(0, 1, fb1); }                                                                           \
        if (ph_ == 2) { MFMA_Q(1, 1, fb1); }                                                                           \
        if (ph_ == 3) { MFMA_Q(1, 0, fb0); }                                                                           \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                                       \
        if (TIMING && !(CHK)) ts_d[ph_] = __builtin_amdgcn_s_memtime();                                                          \
        BAR();                                                                                                         \
    }

    // ---- prologue: pieces 0..5 (K-tile 0 and the first half of K-tile 1); pieces 0, 1 must have landed
    {
        const int kt = 0;
        (void)kt;
        STAGE(0, 0, 0, true); STAGE(0, 1, 1, true); STAGE(0, 2, 2, true); STAGE(0, 3, 3, true);
        STAGE(1, 0, 4, true); STAGE(1, 1, 5, true);
        WAIT_STAGED();
        BAR();
    }
    if (STAG && wm == 1) { BAR(); }
    int kt = 0;
    for (; kt + 3 < nk; kt += 2) {
        PHASE(0, false) PHASE(1, false) PHASE(2, false) PHASE(3, false)
        PHASE(4, false) PHASE(5, false) PHASE(6, false) PHASE(7, false)
    }
    for (; kt + 1 < nk; kt += 2) {
        PHASE(0, true) PHASE(1, true) PHASE(2, true) PHASE(3, true)
        PHASE(4, true) PHASE(5, true) PHASE(6, true) PHASE(7, true)
    }
    if (kt < nk) {
        PHASE(0, true) PHASE(1, true) PHASE(2, true) PHASE(3, true)
    }
    if (STAG && wm == 0) { BAR(); }
    if (TIMING && lane == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            g_stamps[(blockIdx.x * 8 + wave) * 12 + p * 3 + 0] = ts_a[p];
            g_stamps[(blockIdx.x * 8 + wave) * 12 + p * 3 + 1] = ts_b[p];
            g_stamps[(blockIdx.x * 8 + wave) * 12 + p * 3 + 2] = ts_d[p];
        }
    }

    // ---- epilogue: lane (fr, fq) of fragment (i, j) holds row i*16 + fr, columns j*16 + fq*4 .. +3
    // (fragment index i = mi*4 + i', j = ni*2 + j' with rows wm*128 + mi*64 + i'*16, columns wn*64 + ni*32 + j'*16)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = m0 + wm * 128 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fq * 4;
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[i][j][r];
            *reinterpret_cast<half4*>(C + (size_t)row * N + col) = o;
        }
    }
#undef STAGE
#undef READ_A
#undef READ_B
#undef MFMA_Q
#undef PHASE
#endif
}

// The product's loop (gemm.hip k_gemm_f16_dma, tile 15): 256x256x64, 16 waves (4 x 4), 64x64 wave tiles, two LDS stages,
// the next K-tile's DMA issued before this K-tile's MFMAs, s_waitcnt vmcnt(0) + __syncthreads() per K-tile.
__global__ __launch_bounds__(1024) void k_gemm_2barrier(const half_t* __restrict__ A, const half_t* __restrict__ W, half_t* __restrict__ C,
                                                         int M, int N, int K, int tiles_m, int tiles_n, int remap) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int STAGE_B = 512 * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    int tile_m, tile_n;
    tile_of_block(blockIdx.x, tiles_m, tiles_n, remap, tile_m, tile_n);
    const int m0 = tile_m * 256, n0 = tile_n * 256;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((size_t)N * K * 2), 0x00020000);
    const int rsub = lane >> 3, ck = (lane & 7) ^ rsub;
    unsigned a_voff[2], b_voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        a_voff[u] = (unsigned)((m0 + (u * 16 + wave) * 8 + rsub) * K + ck * 8) * 2u;
        b_voff[u] = (unsigned)((n0 + (u * 16 + wave) * 8 + rsub) * K + ck * 8) * 2u;
    }
    const int nk = K >> 6;
#define DMA_TILE(KT, BUF)                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + (BUF) * STAGE_B + (u * 16 + wave) * 1024), 16, a_voff[u], (KT) * 128, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + (BUF) * STAGE_B + 256 * 128 + (u * 16 + wave) * 1024), 16, b_voff[u], (KT) * 128, 0, 0); \
    }
    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    DMA_TILE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frag_a = (wm * 64 + fr) * 128, frag_b = 256 * 128 + (wn * 64 + fr) * 128;
    const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) DMA_TILE(kt + 1, cur ^ 1);
        const char* st = smem + cur * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int sw = ks ? sw1 : sw0;
            half8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const half8*>(st + frag_a + i * 2048 + sw);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const half8*>(st + frag_b + j * 2048 + sw);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
#undef DMA_TILE
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fq * 4;
            half4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[i][j][r];
            *reinterpret_cast<half4*>(C + (size_t)row * N + col) = o;
        }
    }
#endif
}

__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned h = ((unsigned)i + seed) * 2654435761u;   // fp16 in [-2, 2): gemm_loop.hip's operand distribution
        p[i] = (0x3800u | ((h >> 3) & 0x87ffu)) | ((0x3800u | ((h >> 17) & 0x87ffu)) << 16);
    }
}

__global__ void k_ref(const half_t* A, const half_t* W, float* R, int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf((float)A[(size_t)m * K + k], (float)W[(size_t)n * K + k], s);
    R[(size_t)m * N + n] = s;
}

__global__ void k_cmp(const half_t* C, const float* R, size_t n, unsigned* bad, float* maxerr) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float c = (float)C[i], r = R[i], d = fabsf(c - r);
        if (!(d <= 2e-3f * fabsf(r) + 0.02f)) atomicAdd(bad, 1u);
        atomicMax(reinterpret_cast<unsigned*>(maxerr), __float_as_uint(d));
    }
}

__global__ void k_hash(const unsigned* p, size_t n, unsigned long long* out) {
    unsigned long long h = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        h += (unsigned long long)p[i] * (2 * i + 1);
    atomicAdd(out, h);
}

typedef void (*kern_t)(const half_t*, const half_t*, half_t*, int, int, int, int, int, int);
struct Variant { const char* name; kern_t fn; int threads; int lds; };

static void launch(const Variant& v, const half_t* A, const half_t* W, half_t* C, int M, int N, int K, int remap) {
    hipLaunchKernelGGL(v.fn, dim3((M / 256) * (N / 256)), dim3(v.threads), v.lds, 0, A, W, C, M, N, K, M / 256, N / 256, remap);
}

int main(int argc, char** argv) {
    const int big = argc > 1 ? atoi(argv[1]) : 4096;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    std::vector<Variant> vs = {
        {"8phase prio+stagger", k_gemm_8phase<3>, 512, 8 * PIECE},
        {"8phase stagger     ", k_gemm_8phase<2>, 512, 8 * PIECE},
        {"8phase prio        ", k_gemm_8phase<1>, 512, 8 * PIECE},
        {"8phase plain       ", k_gemm_8phase<0>, 512, 8 * PIECE},
        {"8phase p+s drain0  ", k_gemm_8phase<7>, 512, 8 * PIECE},
        {"2barrier 16 waves  ", k_gemm_2barrier, 1024, 2 * 512 * 128},
    };
    for (auto& v : vs) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, v.lds));
    const size_t maxe = (size_t)8192 * 8192;
    half_t *A, *W, *C;
    float* R;
    unsigned* bad;
    float* maxerr;
    unsigned long long* hash;
    CHECK(hipMalloc(&A, maxe * 2)); CHECK(hipMalloc(&W, maxe * 2)); CHECK(hipMalloc(&C, maxe * 2)); CHECK(hipMalloc(&R, maxe * 4));
    CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&maxerr, 4)); CHECK(hipMalloc(&hash, 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)A, maxe / 2, 1u);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)W, maxe / 2, 0x9e3779b9u);
    CHECK(hipDeviceSynchronize());

    // ---- refcheck: 256 / 512 (K = 192: odd K-tile count, tail paths) vs fp64 on the host; big vs the naive device GEMM
    int fails = 0;
    const int shapes[][3] = {{256, 256, 128}, {256, 256, 192}, {256, 256, 256}, {512, 512, 512}, {512, 256, 320}, {256, 512, 64}, {big, big, big}};
    for (const auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<half_t> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
        const bool host = (size_t)M * N * K <= (size_t)512 * 512 * 512;
        if (host) {
            // the kernels read [M][K] / [N][K] with row stride K: refill for this K so that rows differ between shapes
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)A, (size_t)M * K / 2, 7u + K);
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)W, (size_t)N * K / 2, 0x51ed27u + K);
            CHECK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
        } else {
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)A, maxe / 2, 1u);
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)W, maxe / 2, 0x9e3779b9u);
            hipLaunchKernelGGL(k_ref, dim3(N / 64, M / 4), dim3(256), 0, 0, A, W, R, M, N, K);
            CHECK(hipDeviceSynchronize());
        }
        for (const auto& v : vs) {
            CHECK(hipMemset(C, 0xff, (size_t)M * N * 2));
            launch(v, A, W, C, M, N, K, 1);
            CHECK(hipDeviceSynchronize());
            unsigned nbad = 0;
            float me = 0.f;
            if (host) {
                CHECK(hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost));
                for (int m = 0; m < M; ++m)
                    for (int n = 0; n < N; ++n) {
                        double r = 0;
                        for (int k = 0; k < K; ++k) r += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
                        const double d = fabs((double)(float)hC[(size_t)m * N + n] - r);
                        if (!(d <= 2e-3 * fabs(r) + 0.02)) ++nbad;
                        me = std::max(me, (float)d);
                    }
            } else {
                CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(maxerr, 0, 4));
                hipLaunchKernelGGL(k_cmp, dim3(2048), dim3(256), 0, 0, C, R, (size_t)M * N, bad, maxerr);
                CHECK(hipMemcpy(&nbad, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&me, maxerr, 4, hipMemcpyDeviceToHost));
            }
            printf("refcheck %5d x %5d x %5d  %s  %s  bad %u  max|d| %.4f\n", M, N, K, v.name, nbad ? "FAIL" : "ok", nbad, me);
            fails += nbad != 0;
        }
    }
    // ---- race screen: 20 runs of each variant at 256 / 512 / big, output hash must not move
    const int rs[][3] = {{256, 256, 256}, {512, 512, 512}, {big, big, big}};
    for (const auto& s : rs) {
        const int M = s[0], N = s[1], K = s[2];
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)A, maxe / 2, 1u);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned*)W, maxe / 2, 0x9e3779b9u);
        for (const auto& v : vs) {
            unsigned long long h0 = 0;
            int moved = 0;
            for (int run = 0; run < 20; ++run) {
                CHECK(hipMemset(hash, 0, 8));
                launch(v, A, W, C, M, N, K, 1);
                hipLaunchKernelGGL(k_hash, dim3(512), dim3(256), 0, 0, (const unsigned*)C, (size_t)M * N / 2, hash);
                unsigned long long h;
                CHECK(hipMemcpy(&h, hash, 8, hipMemcpyDeviceToHost));
                if (run == 0) h0 = h;
                else moved += h != h0;
            }
            printf("race screen %5d^3  %s  %s (%d of 19 reruns differ)\n", M, v.name, moved ? "UNSTABLE" : "bit-stable", moved);
            fails += moved != 0;
        }
    }
    // ---- timing: interleaved rounds, 40 launches each, random operands in [-2, 2); argv[3] = "zero": all-zero operands (the same
    // instruction stream at a lower power draw: what the clock gives back when the multipliers do not toggle)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const bool zero_fill = argc > 3 && !strcmp(argv[3], "zero");
    if (zero_fill) {
        CHECK(hipMemset(A, 0, maxe * 2));
        CHECK(hipMemset(W, 0, maxe * 2));
        printf("timing on ALL-ZERO operands\n");
    }
    for (int size : {big, 8192}) {
        const int M = size, N = size, K = size;
        const double fl = 2.0 * M * N * (double)K;
        for (int remap = 1; remap <= 2; ++remap) {
            std::vector<std::vector<double>> tf(vs.size());
            for (int r = 0; r < rounds; ++r)
                for (size_t vi = 0; vi < vs.size(); ++vi) {
                    const int reps = size > 4096 ? 10 : 40;
                    for (int w = 0; w < 3; ++w) launch(vs[vi], A, W, C, M, N, K, remap);
                    CHECK(hipEventRecord(e0));
                    for (int i = 0; i < reps; ++i) launch(vs[vi], A, W, C, M, N, K, remap);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    tf[vi].push_back(fl * reps / ms / 1e9);
                }
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                std::sort(tf[vi].begin(), tf[vi].end());
                printf("time %5d^3 remap %d  %s  median %7.1f  min %7.1f  max %7.1f TFLOP/s  (%.1f us per launch)\n", size, remap, vs[vi].name,
                       tf[vi][tf[vi].size() / 2], tf[vi].front(), tf[vi].back(), fl / tf[vi][tf[vi].size() / 2] / 1e6);
            }
        }
    }
    {
        // section timing of one workgroup's last K-tile pair (cycles, relative to wave 0's first stamp)
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_8phase<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * PIECE));
        Variant tv = {"8phase stagger timing", k_gemm_8phase<10>, 512, 8 * PIECE};
        for (int r = 0; r < 3; ++r) launch(tv, A, W, C, big, big, big, 2);   // stamps: phases 4..7 of the last main-loop iteration (K-tile nk - 3)
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(256 * 8 * 12);
        CHECK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8));
        for (int blk : {0, 131}) {
            const unsigned long long t0 = st[(blk * 8 + 0) * 12];
            printf("stamps of workgroup %d (cycles after wave 0's first stamp): per phase [LOAD start, LOAD end, MFMA issue end]\n", blk);
            for (int w = 0; w < 8; ++w) {
                printf("  wave %d:", w);
                for (int p = 0; p < 4; ++p)
                    printf("  [%5lld %5lld %5lld]", (long long)(st[(blk * 8 + w) * 12 + p * 3] - t0), (long long)(st[(blk * 8 + w) * 12 + p * 3 + 1] - t0),
                           (long long)(st[(blk * 8 + w) * 12 + p * 3 + 2] - t0));
                printf("\n");
            }
        }
    }
    printf(fails ? "FAILED (%d)\n" : "all checks passed\n", fails);
    return fails != 0;
}
