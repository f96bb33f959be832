// Deep-pipelined fp16 MFMA GEMM / implicit-GEMM convolution main loop for gfx950 ("ping-pong", tile ids 30..).
//
// Serves the same contraction as gemm.hip (C = epilogue(sum_k A(m,k) W[n][k]), same GemmArgs, same fused epilogues) for
// the shapes that dominate the UNet forward (3x3 convolutions, GEGLU, FF-out): reference call site pipeline/guide.py:56-58
// (the single unet(...) call).  Structure (measured first in tools/micro/gemm_8phase.hip, profiles/r05_micro_8phase.txt:
// 1459 TFLOP/s at 4096^3 / 1603 at 8192^3 on random fp16 against 1353 / 1287 for the 2-barrier loop of gemm.hip):
//
//   * 8 wavefronts as WM_ x WN_, big wave tiles (MI x NI fragments of 16x16, e.g. 128x80), one workgroup per CU.
//   * LDS = a ring of 8 pieces: 2 K-tiles x { A0: the first MI/2 row fragments of every wave row, B0: the first ceil(NI/2)
//     column fragments of every wave column, B1: the remaining column fragments, A1: the remaining rows }.  Every piece is
//     filled by LDS-DMA (buffer_load_dwordx4 ... lds, 8 rows x 128 B per wave instruction, source-side XOR swizzle) and stays
//     in flight ACROSS barriers: the only VM wait in the loop is a counted s_waitcnt vmcnt(one K-tile's worth), never 0.
//   * A K-tile is four phases = the four (row half, column half) quadrants of the wave tile; a phase is
//       [ds_read_b128 of the new operand half + DMA of the piece six pieces ahead + vmcnt]  s_barrier  [MFMAs]  s_barrier
//     phase 0 reads A0 + B0, phase 1 B1, phase 2 A1, phase 3 nothing (B0 stays in registers).
//   * The two halves of the workgroup (waves 0-3 / 4-7: one wave of each per SIMD) run ONE BARRIER APART, so on every SIMD
//     one wave is inside its MFMA section while its partner issues the reads and DMAs of its next one.
//   Ordering (MI355X_MICROARCH.md, two waves per SIMD, item 7): a piece is read one phase after the phase whose counted
//   vmcnt retired it in every wave (RAW), and overwritten no sooner than two phases after its last ds_read (WAR).
//   EVERY phase waits: phase P stages piece P + 6 and its vmcnt(one K-tile's worth) retires pieces <= P + 2, i.e. phase 3 --
//   which reads nothing itself -- retires the B0 piece that phase 0 of the next K-tile reads first.  (Until late in round 5
//   phase 3 did not wait: B0 was then read on the strength of having been issued five phases earlier, and about one launch in
//   25,000 read a row group that had not landed -- found as a sporadic graph-vs-plan mismatch of the 50-step loop.)
//
// DMA addressing is scalar: an 8-row DMA group is 8 consecutive rows of A (or 8 consecutive pixels of one image row for a
// convolution: Wo % 8 == 0), so the per-lane part of the source offset is a constant and everything else (row base, filter
// tap, K offset) rides in the scalar soffset; zero padding is a per-lane select on the column only.  Requirements (checked by
// fd_gemm_pp_ok, everything else stays on gemm.hip's kernels): full tiles, K % 64 == 0, convolutions with Wo % 8 == 0 and no
// fused upsample.
#include "gemm_epilogue.h"

typedef __attribute__((address_space(3))) void* lds_ptr;

template <int WM_, int WN_, int MI, int NI, bool CONV, int EPI, bool HAS_K2>
__global__ __launch_bounds__(512) void k_gemm_f16_pp(GemmArgs g, unsigned a_bytes, unsigned w_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WM_ * WN_ == 8 && MI % 2 == 0, "8 waves; the row halves must be equal");
    constexpr int WTM = MI * 16, WTN = NI * 16, BM = WM_ * WTM, BN = WN_ * WTN;
    constexpr int MH = MI / 2, NI0 = (NI + 1) / 2, NI1 = NI / 2;
    constexpr int PA = WM_ * MH * 16, PB0 = WN_ * NI0 * 16, PB1 = WN_ * NI1 * 16;   // piece rows
    constexpr int OA0 = 0, OB0 = PA * 128, OB1 = OB0 + PB0 * 128, OA1 = OB1 + PB1 * 128, STAGE = OA1 + PA * 128;
    constexpr int GA = PA / 8, GB0 = PB0 / 8, GB1 = PB1 / 8;                        // 8-row DMA groups per piece
    constexpr int CA = (GA + 7) / 8, CB0 = (GB0 + 7) / 8, CB1 = (GB1 + 7) / 8;      // DMA instructions per wave and piece
    static_assert(GA % 8 == 0 && GB1 % 8 == 0, "only B0 may leave some waves one DMA short");
    // one K-tile's worth of this wave's DMAs (waves with an extra B0 group wait a little more than they must: safe)
    constexpr int INFLIGHT = 2 * (GA / 8) + GB0 / 8 + GB1 / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef FD_PP_STAMPS   // diagnostic build (tools/pp_stamps.py): s_memtime stamps of every wave into the split-K workspace
    const unsigned long long ts_start = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz
    unsigned long long ts_a[4] = {0, 0, 0, 0}, ts_b[4] = {0, 0, 0, 0}, ts_d[4] = {0, 0, 0, 0};
#define PP_STAMP(X) X = __builtin_amdgcn_s_memtime();
#else
#define PP_STAMP(X)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN_, wn = wave % WN_;
    const int fr = lane & 15, fq = lane >> 4;
    const int nb = g.tiles_m * g.tiles_n;
    int id = blockIdx.x;
    int kslice = blockIdx.y;
    if (g.sk_flat) {
        // split-K on a flat grid, K slices pinned to XCDs (workgroup b runs on XCD b % 8): the 8 / split_k XCDs of a slice share that
        // slice's weights and input channels, so every weight byte is fetched into 8 / split_k L2s instead of all eight
        // (PMC, 16x16-level conv, 256x320 x split 4 on the 2-D grid: 248 MB read for 50 MB of operands)
        const int per = 8 / g.split_k, xcd = id & 7, slot = id >> 3;     // split_k in {2, 4, 8}, nb % per == 0
        kslice = xcd / per;
        id = (xcd % per) * (nb / per) + slot;                            // each XCD of the slice owns a contiguous range of tiles
    } else {
        const int q = nb >> 3, r = nb & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_n = id % g.tiles_n, tile_m = id / g.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;
    // conv: the descriptor starts pad_l pixels BEFORE the tensor, so that the scalar offset of a row group whose leftmost tap
    // column is -pad_l stays >= 0 (the lanes of such columns never load: their voffset is out of range)
    const int pad_t = CONV ? (g.phase ? 1 - (z >> 1) : g.pad_t) : 0, pad_l = CONV ? (g.phase ? 1 - (z & 1) : g.pad_l) : 0;
    const int a_shift = pad_l * g.Cpix;   // halfs
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.A + (size_t)z * g.strideA - a_shift), 0, a_bytes + 2u * (unsigned)a_shift, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(g.W + (size_t)z * g.strideW), 0, w_bytes, 0x00020000);
    // (the appended phase over A2 -- HAS_K2 -- is a compile-time variant: its descriptor and branches cost ~10 SGPRs, and a
    // convolution kernel short of SGPRs keeps wave-uniform DMA offsets in VGPRs and waterfalls every LDS-DMA)
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(HAS_K2 ? g.A2 : g.A), 0, HAS_K2 ? g.a2_bytes : 0u, 0x00020000);

    // ---- this tile's bias / per-sample bias -> LDS, ahead of the first pieces (see k_gemm_f16_dma) -------------------
    float* bias_s = reinterpret_cast<float*>(smem + 2 * STAGE);
    if ((g.bias && g.bias_lds) || (EPI != 0 && EPI != 7)) {
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(g.bias ? (const void*)(g.bias + (size_t)z * g.strideBias) : (const void*)g.W), 0, g.bias ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
        if (wave * 64 + lane < BN)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(bias_s + wave * 64), 4,
                                                     (unsigned)(n0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }
    const int b_first = m0 / g.rows_per_batch;
    // (LayerNorm fold: bias2 = the weights' column sums, ONE row for every sample -- ldb2 == 0 --, staged whatever the sample boundary)
    const bool b2_staged = g.bias2 && g.bias_lds && (g.ln_stats != nullptr || (min(m0 + BM, g.M) - 1) / g.rows_per_batch == b_first);
    if (b2_staged || (EPI != 0 && EPI != 7)) {
        const __amdgpu_buffer_rsrc_t rsB2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(b2_staged ? (const void*)(g.bias2 + (size_t)b_first * g.ldb2) : (const void*)g.W), 0,
            b2_staged ? (unsigned)((g.N + 3) & ~3) * 4u : 0u, 0x00020000);
        if (wave * 64 + lane < BN)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(bias_s + BN + wave * 64), 4,
                                                     (unsigned)(n0 + wave * 64 + lane) * 4u, 0, 0, 0);
    }

    // ---- DMA source addressing: lane constants + per-group scalars -----------------------------------------------------
    // Scalar registers are the scarce resource of this kernel (a convolution short of SGPRs keeps wave-uniform DMA offsets in
    // VGPRs and then WATERFALLS every LDS-DMA): per DMA group only (pixel index, packed y|x) persist, the K-tile being staged
    // is four counters, everything else is derived where it is used.
    const int rsub = lane >> 3;                            // row inside the 8-row group
    const unsigned ck16 = (unsigned)(((lane & 7) ^ rsub) << 4);   // byte offset of the source chunk of this lane's LDS slot
    constexpr unsigned OOB = 0x80000000u;                 // voffset past every tensor: the load returns zeros
    const unsigned w_lane = (unsigned)(rsub * g.ldw) * 2u + ck16;
    // first row of this wave's DMA group i of A piece mi
#define PP_GROUP_ROW(MIH, I) (m0 + (((I) * 8 + wave) * 8 / (MH * 16)) * WTM + (MIH) * MH * 16 + ((I) * 8 + wave) * 8 % (MH * 16))
    // conv: pixel index of the group's first tap column relative to the (shifted) descriptor, (y + 0x4000) << 16 | (x + 0x4000)
    // of its first tap; linear: the row index
    int sA_px[2][CA], sA_yx[2][CA];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int m = PP_GROUP_ROW(mi, i);
            if (CONV) {
                const int hw = g.Ho * g.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int oy = rem / g.Wo, ox0 = rem - oy * g.Wo;
                const int y = oy * g.stride - pad_t, x = ox0 * g.stride - pad_l;
                sA_yx[mi][i] = __builtin_amdgcn_readfirstlane(((y + 0x4000) << 16) | (x + 0x4000));
                sA_px[mi][i] = __builtin_amdgcn_readfirstlane((b * g.Hi + y) * g.Wi + x + pad_l);
            } else {
                sA_yx[mi][i] = 0;
                sA_px[mi][i] = m;
            }
        }
    // first W row of this wave's DMA group i of piece B0 / B1
#define PP_B_ROW(NIH, I) (n0 + ((((I) * 8 + wave) * 8) / (((NIH) ? NI1 : NI0) * 16)) * WTN + ((NIH) ? NI0 * 16 : 0) + (((I) * 8 + wave) * 8) % (((NIH) ? NI1 : NI0) * 16))

    // ---- K range of this workgroup; the K-tile whose pieces are being staged ------------------------------------------
    const int nkc = g.K / BK;                            // K-tiles of the convolution / GEMM proper
    const int nk_all = nkc + (HAS_K2 ? g.K2 / BK : 0);   // + the appended phase over A2 (a compile-time variant)
    const int kt_per = __builtin_amdgcn_readfirstlane((nk_all + g.split_k - 1) / g.split_k);
    const int kt0 = kslice * kt_per;
    const int nk = min(nk_all, kt0 + kt_per);
    const int nkl = nk - kt0;
    const int ntaps = CONV ? __builtin_amdgcn_readfirstlane(g.K / g.Cin) : 1;
    // convolution: K-tiles visit all filter taps of one 64-channel slice before the next slice (the taps re-read the same input
    // rows, so the re-use distance in the XCD's L2 is one K-tile; 5-9 % faster than (tap, slice) order on this loop).
    // (Interleaving the appended K-tiles with the others -- one every nk_all / nk2 positions, so that their HBM rows do not arrive
    // as one burst at the end -- measured equal on FF-out and 5-15 % SLOWER on the ResBlock convolutions, profiles/r05_ab_pp.txt:
    // the appended form stays.)
    // (integer divisions run on the VALU: pin their wave-uniform results back into SGPRs, or the whole offset chain that hangs
    // off them is moved to VGPRs and every LDS-DMA gets a waterfall loop around its scalar offset)
#define PP_UNIFORM(X) __builtin_amdgcn_readfirstlane(X)
    int t_cur = kt0;                              // position of the K-tile being staged
    int a_idx = HAS_K2 ? max(kt0 - nkc, 0) : 0;   // appended K-tiles before t_cur
    int m_idx = HAS_K2 ? min(kt0, nkc) : kt0;     // main (convolution / GEMM proper) K-tiles before t_cur
    int c_kh = 0, c_kw = 0, c_ci0 = 0;     // conv: filter tap / channel slice of main K-tile m_idx
    if (CONV && m_idx > 0) {
        const int tap = PP_UNIFORM(m_idx % ntaps);
        c_ci0 = PP_UNIFORM(m_idx / ntaps) * BK;
        c_kh = PP_UNIFORM(tap / g.KW);
        c_kw = tap - c_kh * g.KW;
    }
    // (wave-uniform flags are kept as INTEGERS pinned to SGPRs: a uniform i1 combined with && lives in a lane mask, and its
    // zero-extension -- `if (flag) ++a; else ++b;` -- is emitted as VALU arithmetic that drags the offsets into VGPRs)
#define PP_FLAG(COND) __builtin_amdgcn_readfirstlane((COND) ? 1 : 0)
    int cur_a2 = HAS_K2 ? PP_FLAG(t_cur >= nkc) : 0;   // the K-tile at t_cur is an appended one
#define PP_IS_A2() (HAS_K2 && cur_a2 != 0)
#define PP_ADVANCE()                                                                                        \
    {                                                                                                       \
        a_idx += cur_a2;                                                                                    \
        m_idx += 1 - cur_a2;                                                                                \
        if (CONV) {                                                                                         \
            const int adv_ = 1 - cur_a2;                                                                    \
            const int wrap_w_ = PP_FLAG(c_kw + adv_ == g.KW);                                               \
            const int wrap_h_ = PP_FLAG(wrap_w_ != 0 && (c_kh + 1) * g.KW == ntaps);                        \
            c_kw = wrap_w_ ? 0 : c_kw + adv_;                                                               \
            c_kh = wrap_h_ ? 0 : c_kh + wrap_w_;                                                            \
            c_ci0 += wrap_h_ * BK;                                                                          \
        }                                                                                                   \
        ++t_cur;                                                                                            \
        if (HAS_K2) cur_a2 = PP_FLAG(t_cur >= nkc);                                                         \
    }
    // The appended phase reads plain rows A2[m][lda2].  For a convolution it is staged as the CENTRE tap of a second NHWC
    // tensor with lda2 channels (same spatial size, stride 1: row m is pixel (b, oy, ox)), so that one address form serves both:
    //   soffset = pixel index * bytes per pixel + K-tile offset,   lane part = rsub * pixel stride + chunk
    // The scalar operands of a piece's DMAs are PREPARED one phase ahead, behind that phase's MFMAs (the wave has issued
    // them and would otherwise sit at the barrier while they execute): a LOAD section is then only ds_reads + DMA issue.
    // Measured with the s_memtime stamps (tools/pp_stamps.py): with the address arithmetic inside the LOAD sections the
    // A0 section took ~780 cycles against 256-384 of MFMAs in the partner wave.
    int pA_soff[CA], pA_lim[CA], pB_soff[CB0];
#define PP_PREP_A(MIH)                                                                                      \
    {                                                                                                       \
        const bool a2_ = PP_IS_A2();                                                                        \
        const int ok_ = t_cur < nk ? 1 : 0;                                                                 \
        int pixb_, asoff_, kh_;                                                                             \
        if (CONV) {                                                                                         \
            pixb_ = a2_ ? g.lda2 * 2 : g.Cpix * 2;                                                          \
            asoff_ = a2_ ? (pad_t * g.Wi * g.lda2 + a_idx * BK) * 2 : ((c_kh * g.Wi + c_kw) * g.Cpix + c_ci0) * 2; \
            kh_ = a2_ ? pad_t : c_kh;                                                                       \
        } else {                                                                                            \
            pixb_ = a2_ ? g.lda2 * 2 : g.lda * 2;                                                           \
            asoff_ = (a2_ ? a_idx : m_idx) * BK * 2;                                                        \
            kh_ = 0;                                                                                        \
        }                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < CA; ++i) {                                                    \
            pA_soff[i] = sA_px[MIH][i] * pixb_ + asoff_;                                                    \
            if (CONV) {                                                                                     \
                const int y_ = (sA_yx[MIH][i] >> 16) - 0x4000 + kh_;                                        \
                pA_lim[i] = (ok_ != 0 && (unsigned)y_ < (unsigned)g.Hi) ? g.Wi : 0;                         \
            } else {                                                                                        \
                pA_lim[i] = (-ok_) & 0x7fffffff;                                                            \
            }                                                                                               \
        }                                                                                                   \
    }
#define PP_PREP_B(NIH)                                                                                      \
    {                                                                                                       \
        const bool a2_ = PP_IS_A2();                                                                        \
        const int wsoff_ = a2_ ? (g.K + a_idx * BK) * 2                                                     \
                               : (CONV ? ((c_kh * g.KW + c_kw) * g.Cin + c_ci0) * 2 : m_idx * BK * 2);      \
        _Pragma("unroll") for (int i = 0; i < ((NIH) ? CB1 : CB0); ++i)                                     \
            pB_soff[i] = PP_B_ROW(NIH, i) * g.ldw * 2 + wsoff_;                                             \
    }
#define PP_PREP(O)                                      \
    {                                                   \
        if ((O) == 0) { PP_ADVANCE(); PP_PREP_A(0); }   \
        if ((O) == 1) { PP_PREP_B(0); }                 \
        if ((O) == 2) { PP_PREP_B(1); }                 \
        if ((O) == 3) { PP_PREP_A(1); }                 \
    }
    // column test of a (conv) row group: lane x = x0 + rsub * stride must lie in [0, pA_lim); pA_lim is 0 when the tap's row,
    // or the whole K-tile, is out of range.  Linear rows: x == 0 < pA_lim.
#define PP_ISSUE_A(MIH, PAR)                                                                                \
    {                                                                                                       \
        const bool a2_ = PP_IS_A2();                                                                        \
        const int pixs_ = a2_ ? g.lda2 * 2 : (CONV ? g.stride * g.Cpix * 2 : g.lda * 2);                    \
        const int kw_ = a2_ ? pad_l : c_kw;                                                                 \
        const unsigned lanev_ = (unsigned)(rsub * pixs_) + ck16;                                            \
        _Pragma("unroll") for (int i = 0; i < CA; ++i) {                                                    \
            char* dst_ = smem + (PAR) * STAGE + ((MIH) ? OA1 : OA0) + (i * 8 + wave) * 1024;                \
            const int x0_ = CONV ? (sA_yx[MIH][i] & 0xffff) - 0x4000 + kw_ : 0;                             \
            const bool in_ = (unsigned)(x0_ + (CONV ? rsub * g.stride : 0)) < (unsigned)pA_lim[i];          \
            if (a2_)                                                                                        \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)dst_, 16, in_ ? lanev_ : OOB, pA_soff[i], 0, 0); \
            else                                                                                            \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)dst_, 16, in_ ? lanev_ : OOB, pA_soff[i], 0, 0); \
        }                                                                                                   \
    }
#define PP_ISSUE_B(NIH, PAR)                                                                                \
    {                                                                                                       \
        const unsigned wv_ = t_cur < nk ? w_lane : OOB;                                                     \
        _Pragma("unroll") for (int i = 0; i < ((NIH) ? CB1 : CB0); ++i)                                     \
            if ((NIH) || GB0 % 8 == 0 || i * 8 + wave < GB0)                                                \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + (PAR) * STAGE + ((NIH) ? OB1 : OB0) + (i * 8 + wave) * 1024), 16, \
                                                         wv_, pB_soff[i], 0, 0);                            \
    }
    // piece order of a K-tile: 0 A0, 1 B0, 2 B1, 3 A1
#define PP_ISSUE(O, PAR)                                \
    {                                                   \
        if ((O) == 0) { PP_ISSUE_A(0, PAR); }           \
        if ((O) == 1) { PP_ISSUE_B(0, PAR); }           \
        if ((O) == 2) { PP_ISSUE_B(1, PAR); }           \
        if ((O) == 3) { PP_ISSUE_A(1, PAR); }           \
    }

    // ---- fragment reads: row r keeps k-chunk c at 16-byte slot c ^ (r & 7) (conflict-free ds_read_b128) ----------------
    const int sw0 = ((0 + fq) ^ (fr & 7)) << 4, sw1 = ((4 + fq) ^ (fr & 7)) << 4;
    const char* const pa0 = smem + (wm * MH * 16 + fr) * 128 + sw0;
    const char* const pa1 = smem + (wm * MH * 16 + fr) * 128 + sw1;
    const char* const pb00 = smem + OB0 + (wn * NI0 * 16 + fr) * 128 + sw0;
    const char* const pb01 = smem + OB0 + (wn * NI0 * 16 + fr) * 128 + sw1;
    const char* const pb10 = smem + OB1 + (wn * NI1 * 16 + fr) * 128 + sw0;
    const char* const pb11 = smem + OB1 + (wn * NI1 * 16 + fr) * 128 + sw1;
    half8 fa[MH][2], fb0[NI0][2], fb1[NI1 > 0 ? NI1 : 1][2];
#define PP_READ_A(MIH, PAR)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < MH; ++i) {                                                        \
        fa[i][0] = *reinterpret_cast<const half8*>(pa0 + (PAR) * STAGE + ((MIH) ? OA1 : OA0) + i * 2048);   \
        fa[i][1] = *reinterpret_cast<const half8*>(pa1 + (PAR) * STAGE + ((MIH) ? OA1 : OA0) + i * 2048);   \
    }
#define PP_READ_B0(PAR)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NI0; ++j) {                                                       \
        fb0[j][0] = *reinterpret_cast<const half8*>(pb00 + (PAR) * STAGE + j * 2048);                       \
        fb0[j][1] = *reinterpret_cast<const half8*>(pb01 + (PAR) * STAGE + j * 2048);                       \
    }
#define PP_READ_B1(PAR)                                                                                     \
    _Pragma("unroll") for (int j = 0; j < NI1; ++j) {                                                       \
        fb1[j][0] = *reinterpret_cast<const half8*>(pb10 + (PAR) * STAGE + j * 2048);                       \
        fb1[j][1] = *reinterpret_cast<const half8*>(pb11 + (PAR) * STAGE + j * 2048);                       \
    }
    floatx4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#define PP_MFMA(MIH, FB, NJ, J0)                                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                        \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                      \
            _Pragma("unroll") for (int j = 0; j < (NJ); ++j)                                                \
                acc[(MIH) * MH + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                     \
                    FB[j][ks], fa[i][ks], acc[(MIH) * MH + i][(J0) + j], 0, 0, 0);
#define PP_BAR()                          \
    __builtin_amdgcn_sched_barrier(0);    \
    __builtin_amdgcn_s_barrier();         \
    __builtin_amdgcn_sched_barrier(0);
#define PP_WAIT() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
    // phase P (0..7) of a pair of K-tiles: parity P >> 2 is read, the piece six pieces ahead is staged
#define PP_PHASE(P)                                                                                         \
    {                                                                                                       \
        constexpr int par_ = (P) >> 2, ph_ = (P) & 3;                                                       \
        PP_STAMP(ts_a[ph_])                                                                                 \
        if (ph_ == 0) { PP_READ_B0(par_); __builtin_amdgcn_sched_barrier(0); PP_READ_A(0, par_); }          \
        if (ph_ == 1) { PP_READ_B1(par_); }                                                                 \
        if (ph_ == 2) { PP_READ_A(1, par_); }                                                               \
        PP_ISSUE(((P) + 6) & 3, (((P) + 6) >> 2) & 1);                                                      \
        PP_WAIT();                                                                                          \
        PP_STAMP(ts_b[ph_])                                                                                 \
        PP_BAR();                                                                                           \
        if (ph_ == 0) { PP_MFMA(0, fb0, NI0, 0); }                                                          \
        if (ph_ == 1) { PP_MFMA(0, fb1, NI1, NI0); }                                                        \
        if (ph_ == 2) { PP_MFMA(1, fb1, NI1, NI0); }                                                        \
        if (ph_ == 3) { PP_MFMA(1, fb0, NI0, 0); }                                                          \
        PP_PREP(((P) + 7) & 3);                                                                             \
        PP_STAMP(ts_d[ph_])                                                                                 \
        PP_BAR();                                                                                           \
    }

    // ---- prologue: K-tile 0 and the first half of K-tile 1 (six pieces); the first two must have landed ----------------
    PP_PREP_A(0); PP_ISSUE(0, 0); PP_PREP(1); PP_ISSUE(1, 0); PP_PREP(2); PP_ISSUE(2, 0); PP_PREP(3); PP_ISSUE(3, 0);
    PP_PREP(0); PP_ISSUE(0, 1); PP_PREP(1); PP_ISSUE(1, 1);
    PP_PREP(2);   // phase 0 stages B1 of K-tile 1
    // LayerNorm fold fed with the producer's partial sums (fd_gemm_desc.ln_stats_parts): the tile's rows finalised into LDS behind the six
    // prologue pieces; the barriers of the K loop separate this write from the epilogue's reads
    float* const stats_s = bias_s + 4 * BN;
    const bool ln_lds = (EPI == 5 || EPI == 6 || EPI == 7) && __builtin_amdgcn_readfirstlane(g.ln_parts > 1 ? 1 : 0) != 0;
    if constexpr (EPI == 5 || EPI == 6 || EPI == 7) {
        if (ln_lds) ln_tile_stats_to_lds<BM>(g, m0, tid, stats_s);
    }
    const lds_cfloat stats_tile = ln_lds ? (lds_cfloat)stats_s : (lds_cfloat) nullptr;
    PP_WAIT();
    PP_BAR();
    const bool late = wave >= 4;   // the half that runs one barrier behind
    if (late) { PP_BAR(); }
#ifdef FD_PP_STAMPS
    const unsigned long long ts_loop0 = __builtin_amdgcn_s_memtime();
#endif
    int l = 0;
#pragma clang loop unroll(disable)
    for (; l + 1 < nkl; l += 2) {
        PP_PHASE(0) PP_PHASE(1) PP_PHASE(2) PP_PHASE(3)
        PP_PHASE(4) PP_PHASE(5) PP_PHASE(6) PP_PHASE(7)
    }
#ifdef FD_PP_STAMPS
    const unsigned long long ts_sa[4] = {ts_a[0], ts_a[1], ts_a[2], ts_a[3]}, ts_sb[4] = {ts_b[0], ts_b[1], ts_b[2], ts_b[3]},
                             ts_sd[4] = {ts_d[0], ts_d[1], ts_d[2], ts_d[3]};   // phases 4..7 of the last full pair
#endif
    if (l < nkl) {
        PP_PHASE(0) PP_PHASE(1) PP_PHASE(2) PP_PHASE(3)
    }
    if (!late) { PP_BAR(); }
#ifdef FD_PP_STAMPS
    const unsigned long long ts_loop1 = __builtin_amdgcn_s_memtime();
#endif
#undef PP_PHASE
#undef PP_WAIT
#undef PP_BAR
#undef PP_MFMA
#undef PP_READ_A
#undef PP_READ_B0
#undef PP_READ_B1
#undef PP_ISSUE
#undef PP_ISSUE_A
#undef PP_ISSUE_B
#undef PP_PREP
#undef PP_PREP_A
#undef PP_PREP_B
#undef PP_ADVANCE
#undef PP_GROUP_ROW
#undef PP_B_ROW
#undef PP_IS_A2
#undef PP_UNIFORM
#undef PP_FLAG

    // ---- epilogue (the staged pieces past the K range were zero-fill DMAs into ring slots nobody reads; the bias tiles
    // landed with the first counted wait) --------------------------------------------------------------------------------
    if constexpr (EPI == 10) {
        // split-K partial: the raw fp32 tile goes to slab `kslice` of the workspace (k_splitk_finish sums the slabs in a
        // fixed order and applies the epilogue); every tile is full and N % 4 == 0
        float* __restrict__ P = g.ws + (size_t)kslice * g.M * g.N + (size_t)(m0 + wm * WTM + fr) * g.N + n0 + wn * WTN + fq * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
#ifdef FD_SPLITK_NO_STORE   // timing-only variant (tools/seam_probe.py)
                asm volatile("" ::"v"(acc[i][j]));
                if (g.M < 0)
#endif
                *reinterpret_cast<floatx4*>(P + (size_t)i * 16 * g.N + j * 16) = acc[i][j];
            }
    } else if constexpr (EPI == 14) {
        // EXPERIMENTAL in-launch split-K reduction (fd_gemm_desc.sk_sync; measured, not used by the product: DESIGN.md sec. 9 item 1b).
        // (1) publish this slice's fp32 tile exactly as EPI 10 does
        {
            float* __restrict__ P = g.ws + (size_t)kslice * g.M * g.N + (size_t)(m0 + wm * WTM + fr) * g.N + n0 + wn * WTN + fq * 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) *reinterpret_cast<floatx4*>(P + (size_t)i * 16 * g.N + j * 16) = acc[i][j];
        }
        // (2) arrive: the barrier orders every wave's stores before thread 0's agent-scope release (cumulativity), which writes the XCD's
        //     dirty lines back so that the tile's other slices -- on other XCDs -- can read them; (3) wait, bounded: a launch that is not
        //     fully resident would otherwise spin forever
        const int tile_id = tile_m * g.tiles_n + tile_n;
        unsigned* arrive = g.sk_sync + tile_id;
        unsigned* depart = g.sk_sync + g.tiles_m * g.tiles_n + tile_id;
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)g.split_k && ++spins < (1 << 22))
                __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        // (4) finish rows [kslice * BM / S, (kslice + 1) * BM / S) of the tile: the slabs in slice order, then the finish kernel's arithmetic
        {
            const int S = g.split_k, rows = BM / S, r0 = m0 + kslice * rows;
            constexpr int C4 = BN / 4;
            const size_t slab = (size_t)g.M * g.N;
            for (int e = tid; e < rows * C4; e += 512) {
                const int r = e / C4, c4 = e - r * C4;
                const int m = r0 + r, n = n0 + c4 * 4;
                const float* src = g.ws + (size_t)m * g.N + n;
                float4 a = *reinterpret_cast<const float4*>(src);
                for (int sidx = 1; sidx < S; ++sidx) {
                    const float4 q = *reinterpret_cast<const float4*>(src + (size_t)sidx * slab);
                    a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
                }
                float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), b2 = bb;
                if (g.bias) bb = *reinterpret_cast<const float4*>(g.bias + n);
                if (g.bias2) b2 = *reinterpret_cast<const float4*>(g.bias2 + (size_t)(m / g.rows_per_batch) * g.ldb2 + n);
                float v[4] = {fmaf(a.x, g.alpha, bb.x + b2.x), fmaf(a.y, g.alpha, bb.y + b2.y), fmaf(a.z, g.alpha, bb.z + b2.z), fmaf(a.w, g.alpha, bb.w + b2.w)};
                if (g.res) {
                    const half4 rr = *reinterpret_cast<const half4*>(g.res + (size_t)(g.res_rows ? m % g.res_rows : m) * g.ldr + n);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += (float)rr[k];
                }
                half4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (half_t)v[k];
                *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(g.C) + (size_t)m * g.ldc + n) = o;
            }
        }
        // (5) depart: the last slice to leave re-arms the tile's counters for the next launch
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned)g.split_k - 1u) {
                __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if constexpr (EPI == 0 || EPI == 7)
        gemm_epilogue<BM, BN, false, WM_, WN_, EPI == 7>(g, acc, m0, n0, wm, wn, fr, fq, z, (g.bias && g.bias_lds) ? (lds_cfloat)bias_s : (lds_cfloat) nullptr,
                                                         b2_staged ? (lds_cfloat)(bias_s + BN) : (lds_cfloat) nullptr, kslice, stats_tile);
    else if constexpr (EPI == 8 || EPI == 9) {
        // the row-statistics exchange buffer reuses the ring: outstanding zero-fill DMAs must have landed before it is written
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gemm_epilogue_fast<MI, NI, FD_ACT_NONE, EPI == 9, true, false, true, WN_>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            reinterpret_cast<float*>(smem), wm * WTM + fr, wn, m0);
    } else if constexpr (EPI == 11 || EPI == 12) {
        // lean (11) / lean + residual (12) with the GroupNorm partial sums of the tile's output (fd_gemm_desc.gn_part_out): the exchange
        // buffer reuses the ring, as for the row statistics above
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        gemm_epilogue_fast<MI, NI, FD_ACT_NONE, EPI == 12, true, false, false, WN_, WM_>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            reinterpret_cast<float*>(smem), wm * WTM + fr, wn, m0);
    } else
        gemm_epilogue_fast<MI, NI, (EPI == 3 || EPI == 6) ? FD_ACT_GEGLU : FD_ACT_NONE, EPI == 2, true, (EPI == 5 || EPI == 6)>(
            g, acc, m0 + wm * WTM + fr, n0 + wn * WTN, wn * WTN, fq, z, (lds_cfloat)bias_s, (lds_cfloat)(bias_s + BN),
            nullptr, wm * WTM + fr, wn, m0, stats_tile);
#ifdef FD_PP_STAMPS
    if (EPI != 10 && g.ws && lane == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(g.ws) + ((size_t)blockIdx.x * 8 + wave) * 16;
        o[0] = ts_start; o[1] = ts_loop0; o[2] = ts_loop1; o[3] = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int p = 0; p < 3; ++p) { o[4 + 3 * p] = ts_sa[p]; o[5 + 3 * p] = ts_sb[p]; o[6 + 3 * p] = ts_sd[p]; }
        o[13] = ts_sa[3]; o[14] = ts_sb[3];
        o[15] = __builtin_amdgcn_s_memrealtime() - rt_start;   // (replaces phase 3's MFMA-issue stamp)
    }
#endif
#undef PP_STAMP
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
template <int WM_, int WN_, int MI, int NI, bool CONV, int EPI, bool HAS_K2>
static int pp_launch_k2(GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = WM_ * MI * 16, BN = WN_ * NI * 16;
    g.tiles_m = g.M / BM;
    g.tiles_n = g.N / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * 128 + 4 * BN * sizeof(float) + ((EPI == 5 || EPI == 6 || EPI == 7) ? BM * 2 * sizeof(float) : 0);
    const unsigned long long a_bytes = CONV ? 2ull * (((unsigned long long)(g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi - 1) * g.Cpix + g.Cin)
                                            : 2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K);
    const unsigned long long w_bytes = 2ull * ((unsigned long long)(g.N - 1) * g.ldw + g.K + g.K2);
    static std::atomic<unsigned long long> configured{0};   // per device: every ping-pong kernel needs ~150 KB of LDS
    if (fd_first_on_device(&configured))
        FD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_f16_pp<WM_, WN_, MI, NI, CONV, EPI, HAS_K2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // split-K slices pinned to XCDs on a flat grid (see the kernel): -3.5..4 % on the 16x16-level convolutions (128-130 / 218-221 / 174-179 us
    // against 132-136 / 227-232 / 182-186, four interleaved samples each, profiles/r05_ab_pp.txt), neutral at split 2; FD_PP_SK_XCD=0: the 2-D grid (A/B)
    static const int sk_xcd = getenv("FD_PP_SK_XCD") ? atoi(getenv("FD_PP_SK_XCD")) : 1;
    const int nb = g.tiles_m * g.tiles_n;
    g.sk_flat = sk_xcd && (g.split_k == 2 || g.split_k == 4 || g.split_k == 8) && batch == 1 && nb % (8 / g.split_k) == 0 &&
                (nb * g.split_k) % 8 == 0;
    const dim3 grid = g.sk_flat ? dim3(nb * g.split_k, 1, batch) : dim3(nb, g.split_k, batch);
    hipLaunchKernelGGL((k_gemm_f16_pp<WM_, WN_, MI, NI, CONV, EPI, HAS_K2>), grid, dim3(512), lds, st, g,
                       (unsigned)a_bytes, (unsigned)w_bytes);
    FD_CHECK_LAUNCH("k_gemm_f16_pp");
    return FD_OK;
}

template <int WM_, int WN_, int MI, int NI, bool CONV, int EPI>
static int pp_launch_k(GemmArgs& g, int batch, hipStream_t st) {
    // the appended phase exists for the plain / residual / split-K epilogues of convolutions (ResBlock shortcut) and linears
    // (proj_out folded through FF-out)
    if constexpr (EPI == 1 || EPI == 2 || EPI == 10 || EPI == 0 || EPI == 11 || EPI == 12 || EPI == 14) {
        if (g.K2) return pp_launch_k2<WM_, WN_, MI, NI, CONV, EPI, true>(g, batch, st);
    } else {
        if (g.K2) {
            fd_set_error("fd_gemm_f16: the appended A2 phase has no ping-pong kernel with this epilogue");
            return FD_ESHAPE;
        }
    }
    return pp_launch_k2<WM_, WN_, MI, NI, CONV, EPI, false>(g, batch, st);
}

// the lean epilogue when its preconditions hold (launch_epi of gemm.hip), the generic one otherwise
template <int WM_, int WN_, int MI, int NI>
static int pp_launch(GemmArgs& g, int batch, hipStream_t st) {
    constexpr int BM = WM_ * MI * 16, BN = WN_ * NI * 16;
    const bool conv = g.mode == MODE_CONV;
    if (g.split_k > 1 && (g.N & 3) == 0 && g.sk_sync) {
        // experimental in-launch reduction: 320-wide tiles only (the split-K launches of the forward), every workgroup resident at once
        if constexpr (BN == 320) {
            if ((g.M / BM) * (g.N / BN) * g.split_k <= 256 && BM % g.split_k == 0 && batch == 1 && !g.out_f32 && g.act == FD_ACT_NONE && (g.ldc & 3) == 0 &&
                (!g.res || (g.ldr & 3) == 0) && (g.ldb2 & 3) == 0)
                return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 14>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 14>(g, batch, st);
        }
        fd_set_error("fd_gemm_f16: sk_sync (in-launch split-K reduction) needs a 320-wide ping-pong tile, tiles x split_k <= 256 workgroups, a plain fp16 output");
        return FD_ESHAPE;
    }
    if (g.split_k > 1 && (g.N & 3) == 0)
        return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 10>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 10>(g, batch, st);
    const bool lean = g.split_k == 1 && !g.out_f32 && !g.trans_out && g.bias_lds && (g.ldc & 7) == 0 &&
                      (!g.bias2 || g.ln_stats || g.rows_per_batch % BM == 0);
    if (g.gn_part_out) {
        // GroupNorm partial sums of the output: a tile that spans the row (N == BN == 320) and lies in one sample
        if constexpr (BN == 320) {
            if (lean && g.act == FD_ACT_NONE && !g.ln_stats && !g.ln_stats_out && g.N == BN && g.rows_per_batch % BM == 0 && !g.phase && batch == 1 &&
                (!g.res || (g.ldr & 3) == 0)) {
                if (g.res) return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 12>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 12>(g, batch, st);
                return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 11>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 11>(g, batch, st);
            }
        }
        fd_set_error("fd_gemm_f16: gn_part_out on a ping-pong tile needs the lean epilogue of a 320-wide tile inside one sample");
        return FD_ESHAPE;
    }
    if (g.ln_stats_out) {
        if (lean && !conv && (WN_ == 2 || g.N == BN) && g.act == FD_ACT_NONE && !g.ln_stats && !g.bias2) {
            if (g.res && (g.ldr & 3) == 0) return pp_launch_k<WM_, WN_, MI, NI, false, 9>(g, batch, st);
            if (!g.res) return pp_launch_k<WM_, WN_, MI, NI, false, 8>(g, batch, st);
        }
        fd_set_error("fd_gemm_f16: ln_stats_out on a ping-pong tile needs the lean epilogue of a plain or residual linear GEMM");
        return FD_ESHAPE;
    }
    if (g.ln_stats) {
        if (lean && !conv && !g.res) {
            if constexpr (NI % 2 == 0) {
                if (g.act == FD_ACT_GEGLU) return pp_launch_k<WM_, WN_, MI, NI, false, 6>(g, batch, st);
            }
            if (g.act == FD_ACT_NONE) return pp_launch_k<WM_, WN_, MI, NI, false, 5>(g, batch, st);
        }
        if (!conv) return pp_launch_k<WM_, WN_, MI, NI, false, 7>(g, batch, st);
        fd_set_error("fd_gemm_f16: LayerNorm fold on a convolution");
        return FD_ESHAPE;
    }
    if (lean) {
        if constexpr (NI % 2 == 0) {
            if (g.act == FD_ACT_GEGLU && !conv) return pp_launch_k<WM_, WN_, MI, NI, false, 3>(g, batch, st);
        }
        if (g.act == FD_ACT_NONE && g.res && (g.ldr & 3) == 0)
            return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 2>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 2>(g, batch, st);
        if (g.act == FD_ACT_NONE && !g.res)
            return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 1>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 1>(g, batch, st);
    }
    return conv ? pp_launch_k<WM_, WN_, MI, NI, true, 0>(g, batch, st) : pp_launch_k<WM_, WN_, MI, NI, false, 0>(g, batch, st);
}

void fd_gemm_pp_tile_shape(int tile, int* bm, int* bn) {
    switch (tile) {
        case 30: *bm = 256; *bn = 320; break;   // 2 x 4 waves of 128 x 80
        case 31: *bm = 256; *bn = 256; break;   // 2 x 4 waves of 128 x 64
        case 32: *bm = 128; *bn = 320; break;   // 2 x 4 waves of 64 x 80
        case 33: *bm = 256; *bn = 160; break;   // 4 x 2 waves of 64 x 80
        default: *bm = *bn = 0; break;
    }
}

bool fd_gemm_pp_ok(const GemmArgs& g, int batch, int tile) {
    int bm, bn;
    fd_gemm_pp_tile_shape(tile, &bm, &bn);
    if (!bm) return false;
    if (g.M % bm != 0 || g.N % bn != 0 || g.K % BK != 0 || g.K2 % BK != 0 || g.trans_out) return false;
    if (g.act == FD_ACT_GEGLU && (bn / (tile == 33 ? 2 : 4)) % 32 != 0) return false;   // value / gate fragment pairs per wave
    const unsigned long long a_bytes = g.mode == MODE_CONV ? 2ull * (((unsigned long long)(g.M / (g.Ho * g.Wo)) * g.Hi * g.Wi - 1) * g.Cpix + g.Cin)
                                                           : 2ull * ((unsigned long long)(g.M - 1) * g.lda + g.K);
    const unsigned long long w_bytes = 2ull * ((unsigned long long)(g.N - 1) * g.ldw + g.K + g.K2);
    if (a_bytes >= 0x7ffffff0ull || w_bytes >= 0x7ffffff0ull) return false;
    if (g.mode == MODE_CONV) {
        // an 8-row DMA group must be 8 consecutive pixels of one image row; no fused nearest upsample (per-lane x >> 1)
        if (g.Wo % 8 != 0 || g.up || g.Cin % BK != 0) return false;
        // the appended phase is staged as the centre tap of a second tensor of the same spatial size
        if (g.K2 && (g.stride != 1 || g.Ho != g.Hi || g.Wo != g.Wi || g.phase)) return false;
    }
    if (g.split_k > 1 && (g.K / BK + g.K2 / BK) / g.split_k < 2) return false;
    (void)batch;
    return true;
}

int fd_gemm_pp_launch(GemmArgs& g, int batch, hipStream_t st, int tile) {
    switch (tile) {
        case 30: return pp_launch<2, 4, 8, 5>(g, batch, st);
        case 31: return pp_launch<2, 4, 8, 4>(g, batch, st);
        case 32: return pp_launch<2, 4, 4, 5>(g, batch, st);
        case 33: return pp_launch<4, 2, 4, 5>(g, batch, st);
        default: fd_set_error("fd_gemm_f16: unknown ping-pong tile %d", tile); return FD_EINVAL;
    }
}
