// Reproducer for VERDICT r3 weak #4 / next #5: does a dependent accumulator chain that MIXES v_mfma_f32_16x16x32_f16 and
// v_mfma_f32_16x16x16_f16 give wrong / run-to-run different sums on gfx950?  (round 3 saw exactly that inside k_xattn40.)
//
// Every wave computes C = A32 B32^T + A16 B16^T (16 x 16, K = 32 + 16) `chain` times into ONE accumulator -- integer-valued
// fp16 operands, so every partial sum is exact in fp32 and the expected result is known bit for bit -- in five forms:
//   0  k32 then k16, back to back (what hipcc schedules by itself)
//   1  k16 then k32
//   2  k32, s_nop 7, k16, s_nop 7
//   3  the K = 16 product issued as the K = 32 form on zero-extended operands (the product's workaround)
//   4  as 0 but through inline asm with NO compiler-inserted nops between the two shapes (hazard probe)
//   100+N / 200+N / 300+N  wait-state scans: N x s_nop 0 between k32 -> k16, k16 -> k32, and (control) k32 -> k32
// Each form runs `launches` times on all 256 CUs x 16 waves; every lane's 4 results are compared with the expected value.
// hipcc --offload-arch=gfx950 -O3 mfma_mix_repro.hip -o mfma_mix_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// operand values: small integers, different per (row / column, k)
__host__ __device__ inline int va(int row, int k) { return ((row * 3 + k * 5) % 7) - 3; }
__host__ __device__ inline int vb(int col, int k) { return ((col * 5 + k * 3) % 5) - 2; }

template <int FORM>
__global__ __launch_bounds__(1024) void k(float* out, unsigned* bad, int chain) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    // 16x16x32: lane (r, q) holds k = 8 q .. 8 q + 7 of row r (A) / column r (B); 16x16x16: k = 4 q .. 4 q + 3 (offset 32)
    half8 a8, b8, a8z, b8z;
    half4 a4, b4;
    for (int i = 0; i < 8; ++i) {
        a8[i] = (half_t)va(r, q * 8 + i);
        b8[i] = (half_t)vb(r, q * 8 + i);
    }
    for (int i = 0; i < 4; ++i) {
        a4[i] = (half_t)va(r, 32 + q * 4 + i);
        b4[i] = (half_t)vb(r, 32 + q * 4 + i);
    }
    // zero-extended K = 32 operands carrying the 16 remainder columns at k = 0..15 (lanes q = 0, 1)
    for (int i = 0; i < 8; ++i) {
        const int kk = q * 8 + i;
        a8z[i] = kk < 16 ? (half_t)va(r, 32 + kk) : (half_t)0;
        b8z[i] = kk < 16 ? (half_t)vb(r, 32 + kk) : (half_t)0;
    }
    floatx4 acc = {0, 0, 0, 0};
    for (int it = 0; it < chain; ++it) {
        if (FORM == 0) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
        } else if (FORM == 1) {
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
        } else if (FORM == 2) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
            asm volatile("s_nop 7\n s_nop 7" : "+v"(acc));
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
            asm volatile("s_nop 7\n s_nop 7" : "+v"(acc));
        } else if (FORM == 3) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8z, b8z, acc, 0, 0, 0);
        } else if (FORM == 4) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n"
                         : "+v"(acc) : "v"(a8), "v"(b8), "v"(a4), "v"(b4));
        } else if (FORM >= 100 && FORM < 200) {   // k32, (FORM - 100) wait states, k16, 16 wait states
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         ".rept %5\n s_nop 0\n .endr\n"
                         "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n"
                         "s_nop 7\n s_nop 7\n"
                         : "+v"(acc) : "v"(a8), "v"(b8), "v"(a4), "v"(b4), "n"(FORM - 100));
        } else if (FORM >= 200 && FORM < 300) {   // k16, (FORM - 200) wait states, k32, 16 wait states
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n"
                         ".rept %5\n s_nop 0\n .endr\n"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         "s_nop 7\n s_nop 7\n"
                         : "+v"(acc) : "v"(a8), "v"(b8), "v"(a4), "v"(b4), "n"(FORM - 200));
        } else if (FORM >= 300 && FORM < 400) {   // control: k32, (FORM - 300) wait states, k32 (same shape), 16 wait states
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n"
                         ".rept %5\n s_nop 0\n .endr\n"
                         "v_mfma_f32_16x16x32_f16 %0, %3, %4, %0\n"
                         "s_nop 7\n s_nop 7\n"
                         : "+v"(acc) : "v"(a8), "v"(b8), "v"(a8z), "v"(b8z), "n"(FORM - 300));
        }
    }
    asm volatile("s_nop 7\n s_nop 7" : "+v"(acc));
    // expected: D[row = 4 q + e][col = r] = chain * sum_{k < 48} A[row][k] B[col][k]   (the builtin computes A B^T with
    // the FIRST operand's rows as D rows: lane (r, q) holds D[4 q + e][r])
    unsigned nbad = 0;
    for (int e = 0; e < 4; ++e) {
        int s = 0;
        for (int kk = 0; kk < 48; ++kk) s += va(4 * q + e, kk) * vb(r, kk);
        const float want = (float)s * (float)chain;
        if (acc[e] != want) ++nbad;
        out[(blockIdx.x * 1024 + threadIdx.x) * 4 + e] = acc[e] - want;
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int FORM>
void run(float* out, unsigned* bad, int chain, int launches, const char* what) {
    unsigned total = 0, bad_launches = 0;
    for (int l = 0; l < launches; ++l) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL((k<FORM>), dim3(256), dim3(1024), 0, 0, out, bad, chain);
        unsigned h = 0;
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        total += h;
        bad_launches += h != 0;
    }
    printf("form %d (%s): chain %d, %d launches x 262144 lanes x 4 values: %u wrong values in %u launches\n", FORM, what, chain, launches, total,
           bad_launches);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20;
    float* out; unsigned* bad;
    hipMalloc(&out, (size_t)256 * 1024 * 4 * 4); hipMalloc(&bad, 4);
    for (int chain : {1, 7, 64}) {
        run<0>(out, bad, chain, launches, "k32 then k16, compiler-scheduled");
        run<1>(out, bad, chain, launches, "k16 then k32, compiler-scheduled");
        run<2>(out, bad, chain, launches, "k32, 16 nops, k16, 16 nops");
        run<3>(out, bad, chain, launches, "k16 as zero-extended k32");
        run<4>(out, bad, chain, launches, "inline asm, back to back, no nops");
    }
    // how many wait states does the dependent pair need?  (chain 7; the second gap is always 16 states)
    printf("# wait-state scan, chain 7: k32 -> N x s_nop 0 -> k16\n");
    run<100>(out, bad, 7, launches, "k32, 0, k16"); run<101>(out, bad, 7, launches, "k32, 1, k16"); run<102>(out, bad, 7, launches, "k32, 2, k16");
    run<103>(out, bad, 7, launches, "k32, 3, k16"); run<104>(out, bad, 7, launches, "k32, 4, k16"); run<105>(out, bad, 7, launches, "k32, 5, k16");
    run<106>(out, bad, 7, launches, "k32, 6, k16"); run<107>(out, bad, 7, launches, "k32, 7, k16"); run<108>(out, bad, 7, launches, "k32, 8, k16");
    run<110>(out, bad, 7, launches, "k32, 10, k16"); run<112>(out, bad, 7, launches, "k32, 12, k16");
    printf("# wait-state scan, chain 7: k16 -> N x s_nop 0 -> k32\n");
    run<200>(out, bad, 7, launches, "k16, 0, k32"); run<201>(out, bad, 7, launches, "k16, 1, k32"); run<202>(out, bad, 7, launches, "k16, 2, k32");
    run<203>(out, bad, 7, launches, "k16, 3, k32"); run<204>(out, bad, 7, launches, "k16, 4, k32"); run<205>(out, bad, 7, launches, "k16, 5, k32");
    run<206>(out, bad, 7, launches, "k16, 6, k32"); run<207>(out, bad, 7, launches, "k16, 7, k32"); run<208>(out, bad, 7, launches, "k16, 8, k32");
    run<210>(out, bad, 7, launches, "k16, 10, k32"); run<212>(out, bad, 7, launches, "k16, 12, k32");
    printf("# control, same shape: k32 -> N x s_nop 0 -> k32 (zero-extended operands)\n");
    run<300>(out, bad, 7, launches, "k32, 0, k32"); run<301>(out, bad, 7, launches, "k32, 1, k32"); run<302>(out, bad, 7, launches, "k32, 2, k32");
    return 0;
}
