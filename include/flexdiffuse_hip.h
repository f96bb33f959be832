/*
 * libflexdiffuse_hip.so -- C ABI of the MI355X (gfx950) hot path of flexdiffuse.
 *
 * The reference (tim-speed/flexdiffuse) has no FFI of its own: its boundary is three
 * duck-typed Python protocols (guidance.py:315-474 `Guide`, pipeline/guide.py:8-72
 * `GuideBase`, pipeline/flex.py:46-310 `FlexPipeline`).  The host side of this
 * project keeps those protocols in Python (package `flexdiffuse_amd`) and binds the
 * entry points below with ctypes (flexdiffuse_amd/hip.py).  INTEGRATION.md shows the
 * stub a maintainer of the reference would add.
 *
 * Conventions
 *  - every function returns FD_OK (0) or a negative FD_E* code; the message for the
 *    calling thread is available from fd_last_error();
 *  - all pointers are BORROWED DEVICE pointers (caller owns all memory; the library
 *    never allocates device memory) unless a parameter says "host";
 *  - `stream` is a hipStream_t passed as void*; every launch is asynchronous;
 *  - activations are NHWC / row-major fp16 ("h"), accumulation and statistics fp32;
 *  - no global state except the thread-local error string and the optional
 *    kernel-timing recorder (fd_prof_*).
 */
#ifndef FLEXDIFFUSE_HIP_H
#define FLEXDIFFUSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FD_OK 0
#define FD_EINVAL (-1) /* bad argument value */
#define FD_ESHAPE (-2) /* unsupported shape / alignment */
#define FD_EHIP (-3)   /* HIP runtime error */

#define FD_ABI_VERSION 12

int fd_abi_version(void);
const char* fd_last_error(void);
/* CU count, max engine clock (kHz), total HBM bytes and gcnArchName of `device`. */
int fd_device_info(int device, int* cu_count, int* clock_khz, int64_t* hbm_bytes,
                   char* arch, int arch_len);

/* Optional per-launch timing of a kernel family with HIP events recorded on the launch
 * stream (bench.py roofline leg).  Families: */
#define FD_FAMILY_GEMM 0      /* implicit-GEMM conv / GEMM MFMA kernel */
#define FD_FAMILY_ATTENTION 1 /* flash attention MFMA kernel */
#define FD_FAMILY_GROUPNORM 2 /* GroupNorm(+SiLU) statistics + apply */
#define FD_FAMILY_OTHER 3     /* (ABI 11) every other launch of the library: layout / elementwise / LayerNorm statistics / guidance */
int fd_prof_enable(int on);
/* Record events around every `stride`-th launch of each family only (default 1 = all): a pair of
 * event records costs ~6.6 us of stream time, which at one pair per launch is ~10 % of the pass
 * being measured.  fd_prof_collect then sums over the sampled launches. */
int fd_prof_set_stride(int stride);
/* Synchronises the recorded events of `family`, returns summed elapsed ms, declared work
 * (FLOPs for MFMA families, algorithmic bytes for HBM families) and launch count, and
 * forgets those records. Host pointers. */
int fd_prof_collect(int family, double* total_ms, double* total_work, int64_t* launches);
/* (ABI 9) the same plus `total_executed`: the work the kernels really issued -- equal to the declared work except
 * where a launch implements an op with fewer MACs than its definition (parity-decomposed upsample convolution: 4/9).
 * The hardware roofline is `executed` / time; `work` / time is the algorithmic-equivalent rate. */
int fd_prof_collect2(int family, double* total_ms, double* total_work, double* total_executed, int64_t* launches);
/* (ABI 11) Every recorded bracket one by one, in launch order: family, tag (an identity of the launch's shape and kernel choice; 0 =
 * none), elapsed ms, declared and executed work; then forgets all records.  Host arrays of `cap` entries, *n = entries written.
 * bench.py's roofline leg (the measurement of the `unet(...)` call of reference pipeline/guide.py:56-58) takes the MEDIAN per
 * (family, tag, work) group x the group's launch count, so one host stall inside one bracket cannot move a family's total. */
int fd_prof_drain(int32_t* family, uint32_t* tag, float* ms, double* work, double* executed, int64_t cap, int64_t* n);
/* Mean elapsed ms of an EMPTY event bracket (`pairs` back-to-back record pairs on `stream`);
 * subtract it per sampled launch to turn bracket time into kernel time. Host pointer. */
int fd_prof_calibrate(int pairs, double* ms_per_empty_pair, void* stream);

/* ------------------------------------------------------------------------------------
 * Launch plan.  The denoising loop (reference pipeline/flex.py:262-287) calls the UNet with the
 * same shapes at every step: the ~430 launches of a forward differ only in the CONTENTS of the
 * latent and timestep buffers.  While a thread records (fd_plan_record_begin .. _end) every launch
 * entry point of this library, besides launching, appends a by-value copy of its own call to the
 * plan; fd_plan_replay issues the recorded calls again on `stream` as ordinary eager launches
 * (same kernels, order and tile choices; none of the host front's per-op work).  The caller keeps
 * every recorded device address alive and unchanged and refreshes the input buffers between
 * replays.  Recording is per thread; a plan may be replayed from any ONE thread at a time.
 * ---------------------------------------------------------------------------------- */
typedef struct fd_plan fd_plan;
int fd_plan_create(fd_plan** out);
int fd_plan_destroy(fd_plan* plan);
int fd_plan_record_begin(fd_plan* plan); /* clears the plan; the calling thread records */
int fd_plan_record_end(fd_plan* plan);
int fd_plan_size(const fd_plan* plan, int* launches); /* recorded entry-point calls */
int fd_plan_replay(const fd_plan* plan, void* stream);

/* ------------------------------------------------------------------------------------
 * Guidance: CLIP image<->text token alignment and tween  (reference guidance.py)
 * ---------------------------------------------------------------------------------- */
#define FD_ORDER_TEXT 0   /* guidance.py:18 GUIDE_ORDER_TEXT   */
#define FD_ORDER_ALIGN 1  /* guidance.py:19 GUIDE_ORDER_ALIGN  */
#define FD_ORDER_DIRECT 2 /* guidance.py:20 GUIDE_ORDER_DIRECT */

/* Scratch floats needed by fd_guidance_map / fd_guidance_tween: B*N*L. */
int64_t fd_guidance_workspace_floats(int B, int N, int L);

/* Replaces guidance.py:23-85 `_map_emb` for B prompts at once.
 *   alt  [Balt][N][D] f32 guide tokens (alt_batched=0: one guide shared by all prompts)
 *   txt  [B][L][D]    f32 text tokens
 *   idx  [B][L] i32, s [B][L] f32 : row j = (guide index, similarity) of text column
 *        j+1, exactly as the reference's (L,2) array (last row stays (0,0)).
 * D % 32 == 0, L <= 96, N*(L|1)*4 <= 150 KiB. */
int fd_guidance_map(const float* alt, const float* txt, float* ws, int32_t* idx, float* s,
                    int B, int alt_batched, int N, int L, int D, int order, int reuse,
                    void* stream);

typedef struct fd_tween_params {
    double threshold_floor; /* guidance.py:205 */
    double threshold_mult;  /* guidance.py:206 */
    double clustered;       /* guidance.py:209 */
    double max_guidance;    /* guidance.py:210 */
    double header_max;      /* guidance.py:211 */
    int32_t order;          /* guidance.py:212 align_mode */
    int32_t reuse;          /* guidance.py:213 mapping_reuse */
} fd_tween_params;

/* Replaces guidance.py:215-272 `Tweener.tween` for B prompts at once (map + weights +
 * blend in two launches, no host round trip).
 *   base   [B][L][D] f32 text embeddings; alt as in fd_guidance_map
 *   lin_w  [L] f32 = torch.linspace(linear_start, linear_end, L) (guidance.py:231-233;
 *          computed by the host exactly as the reference does)
 *   out    [B][L][D] f32 tweened embeddings
 *   weights[B][L] f32 blend weights after the header cap (the vector the reference
 *          prints at guidance.py:255); idx/s as in fd_guidance_map
 *   status [B] i32: 0 ok, 1 = adjacent equal similarity peaks (the reference raises
 *          ZeroDivisionError at guidance.py:112; out is then the un-tweened base). */
int fd_guidance_tween(const float* base, const float* alt, const float* lin_w, float* ws,
                      float* out, float* weights, int32_t* idx, float* s, int32_t* status,
                      int B, int alt_batched, int N, int L, int D,
                      const fd_tween_params* p, void* stream);

/* Replaces guidance.py:288-312 `ConceptMapper.map`: for text row j, c = ct_idx[j];
 * if c >= 1 and ct_s[j] > 0.9: out[j+1] = guide[cm_idx[c-1]].  guide [N][D], out [L][D]. */
int fd_guidance_concept_override(const float* guide, const int32_t* cm_idx,
                                 const int32_t* ct_idx, const float* ct_s, float* out,
                                 int N, int L, int D, void* stream);

/* guidance.py:467-472 pure-image path: out[b][0][:] += (hdr[:] - out[b][0][:]) * 0.85 */
int fd_guidance_header_pull(float* out, const float* hdr, int B, int L, int D, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense contractions: fp16 MFMA GEMM / implicit-GEMM convolution with fused epilogue.
 * Replaces the cuDNN/cuBLAS kernels diffusers/transformers launch inside
 * `unet(...)` (reference pipeline/guide.py:56-58), `vae.decode/encode`
 * (pipeline/flex.py:118,189) and the CLIP towers (encode/clip.py:64,86-100).
 * ---------------------------------------------------------------------------------- */
#define FD_ACT_NONE 0
#define FD_ACT_SILU 1
#define FD_ACT_QUICK_GELU 2 /* x * sigmoid(1.702 x)  (CLIP) */
#define FD_ACT_GELU 3       /* exact erf GELU */
#define FD_ACT_GEGLU 4      /* value * gelu(gate); weight rows interleaved 16 value / 16 gate */

typedef struct fd_gemm_desc {
    const void* A;        /* fp16: linear [M][lda]; conv: NHWC input [B][in_h][in_w][in_c], pixel stride lda halfs (0 = in_c; >= in_c, % 8) */
    const void* W;        /* fp16 [N][ldw], K contiguous (conv: [Cout][kh][kw][Cin]) */
    void* C;              /* fp16 (or fp32 if out_f32) [M][ldc]; GEGLU: [M][N/2] */
    const float* bias;    /* [N] or NULL */
    const float* bias2;   /* per-sample bias [M/rows_per_sample][ld_bias2] or NULL */
    const void* residual; /* fp16 [M][ldr] added after the activation, or NULL */
    int32_t M, N, K;
    int32_t lda, ldw, ldc, ldr, ld_bias2;
    int32_t rows_per_sample; /* rows of A per sample (H*W or tokens); 0 = M */
    int32_t act;             /* FD_ACT_* */
    int32_t out_f32;
    float alpha;             /* scales the accumulator before bias; 0 means 1 */
    /* implicit-GEMM convolution (conv != 0): in_c % 64 == 0, K = kh*kw*in_c.
     * upsample2x: 1 = the input is read through a fused nearest-2x upsample (in_h / in_w are the LOW-resolution size).
     * 2 = PHASE-DECOMPOSED upsample convolution: diffusers' Upsample2D (nearest 2x, then conv3x3; inside `unet(...)`,
     * reference pipeline/guide.py:56-58, and `vae.decode`, pipeline/flex.py:118) computed as four 2x2 convolutions of the
     * low-resolution input, one per output-pixel parity (py, px) = (z >> 1, z & 1) with batch = 4: W is [4][N][2*2*in_c]
     * (batch_stride_w = N * ldw), the 3x3 taps that fall on the same source pixel pre-summed per parity (4/9 of the MACs,
     * same sums); M = B * in_h * in_w rows of the low-resolution grid, out_h = in_h, out_w = in_w, kh = kw = 2, and C is the
     * FULL-resolution [B][2 in_h][2 in_w][ldc] map into which slice z writes the pixels (2y + py, 2x + px). */
    int32_t conv, in_h, in_w, in_c, out_h, out_w, kh, kw, stride, pad_t, pad_l, upsample2x;
    /* transposed store: C is [sample][N][trans_ld] (row n, column m within the sample) */
    int32_t trans_out, trans_ld;
    int64_t trans_sample_stride;
    /* batched GEMM over blockIdx.z: element strides per batch */
    int32_t batch;
    int64_t batch_stride_a, batch_stride_w, batch_stride_c, batch_stride_res;
    /* scheduling: tile 0 = auto, 1 = 128x128, 2 = 128x160, 3 = 128x64, 4 = 64x64, 5 = 256x160 and
     * 6 = 256x128 (8 waves), 7/8 = the same with 3 LDS stages, 9 = 128x160 and 10 = 128x128 with
     * 8 waves, 11 = 128x64 with 8 waves, 12 = 128x160, 13 = 256x160 and 14 = 256x128 with 16 waves; 30 = 256x320, 31 = 256x256, 32 = 128x320, 33 = 256x160 on
     * the deep-pipelined ping-pong loop (csrc/gemm_pp.hip: 8 waves, full tiles and K % 64 == 0 only, FD_ESHAPE otherwise);
     * split_k 0 =
     * auto (needs `workspace`, fp32 [split_k][M][N]), 1 = off. Results are deterministic for
     * a given (shape, tile, split_k). */
    int32_t tile, split_k;
    void* workspace;
    int64_t workspace_bytes;
    /* LayerNorm folded into the GEMM (replaces the separate LayerNorm launch diffusers' BasicTransformerBlock
     * runs before attn1 / attn2 / ff inside `unet(...)`, reference pipeline/guide.py:56-58).  A holds the
     * UN-normalised rows x, W the weights pre-multiplied by the LayerNorm gain (W' = W diag(gamma), fp16),
     * bias = b + W beta, ln_colsum[n] = sum_k W'[n][k] (of the fp16 values), ln_stats[m] = (rstd_m,
     * -mean_m rstd_m) from fd_ln_row_stats_f16:
     *   C[m][n] = act(rstd_m (x W'^T)[m][n] - rstd_m mean_m ln_colsum[n] + bias[n])  ==  act(LN(x) W^T + b).
     * Linear GEMMs only; act NONE or GEGLU; works with trans_out.  NULL = off. */
    const float* ln_stats;  /* fp32 [M][2] */
    const float* ln_colsum; /* fp32 [N] */
    /* Producer side of the fold: also write ln_stats_out[m] = (rstd_m, -mean_m rstd_m) of the OUTPUT rows
     * (LayerNorm over the N columns, eps ln_eps, statistics of the fp16-rounded values), so that the next
     * GEMM can take them as its ln_stats without a separate statistics pass.  N == 320, M % 256 == 0: one workgroup
     * tile spans the row and writes the finished pair.  N a larger multiple of 160, M % 128 == 0: ln_stats_out is
     * [N / 160][M][2] RAW partial sums (sum, sum of squares per 160-column tile), to be combined by
     * fd_ln_finalize_stats_f32.  fd_gemm_can_emit_row_stats returns the slab count (0: not possible, 1: finished).
     * Linear, plain or residual epilogue. NULL = off. */
    float* ln_stats_out;
    float ln_eps;
    /* APPENDED phase: after the K columns of the GEMM / convolution the same K loop runs K2 more columns over the plain
     * rows of a second operand A2 [M][lda2] against W[:, K .. K+K2):
     *   C = epilogue( A-part(A, W[:, :K]) + A2 W[:, K:]^T ).
     * Two uses inside `unet(...)` (reference pipeline/guide.py:56-58), both weight folds done once at model construction:
     * a ResBlock's 1x1 shortcut convolution rides in conv2's accumulation (diffusers ResnetBlock2D.conv_shortcut: no
     * shortcut launch, no shortcut tensor written and re-read as the residual); and a transformer block's proj_out is
     * folded THROUGH the feed-forward output layer, proj_out(ff2(f) + h) = f (Wp W2)^T + h Wp^T + (bp + Wp b2): one GEMM
     * with A = f, A2 = h instead of two.  W is [N][ldw] with ldw >= K + K2; K % 64 == 0, K2 % 64 == 0; LDS-DMA path only. */
    const void* A2;
    int32_t lda2, K2;
    /* (ABI 8) batch > 1: `bias` advances by this many floats per batch (0: one bias for every batch).  With per-batch
     * weights (batch_stride_w) this is the GroupNorm fold of fd_groupnorm_fold_linear_f16: sample b's rows are
     * multiplied by ITS weights and get ITS bias.  LDS-DMA path with LDS-staged biases only.  With batch > 1,
     * ln_stats_out needs batch_stride_c == M * ldc (row index = batch * M + m). */
    int64_t batch_stride_bias;
    /* (ABI 11) GroupNorm(+SiLU) of the OUTPUT fused into the split-K finish pass: gn_out [M][N] (contiguous fp16) =
     * act(GroupNorm_{gn_groups}(C; gn_gamma, gn_beta, gn_eps)) over the rows_per_sample rows x N / gn_groups channels of each
     * (sample, group), statistics of the fp16-rounded values of C -- bit for bit what fd_groupnorm_nhwc_f16 gives on C.  Inside
     * `unet(...)` (reference pipeline/guide.py:56-58) a ResBlock's conv1 feeds nothing but norm2 + SiLU (diffusers ResnetBlock2D):
     * at the 16x16 / 8x8 levels, where the convolution is split over K, the pass that sums the fp32 partial slabs normalises the
     * tile it has just summed -- one launch less, and with gn_skip_c the convolution's own output never goes to HBM (C may then be
     * NULL).  Only honoured by split-K launches (ask fd_gemm_plan for the split, fd_gemm_can_fuse_groupnorm for the shape);
     * FD_ESHAPE otherwise.  act NONE, fp16 output, batch 1; residual row stride % 8 == 0.  NULL = off. */
    void* gn_out;
    const float* gn_gamma; /* [N] */
    const float* gn_beta;  /* [N] */
    int32_t gn_groups, gn_silu;
    float gn_eps;
    int32_t gn_skip_c;
    /* (ABI 11) GroupNorm PARTIAL SUMS of the output from the epilogue of the producing launch: gn_part_out
     * [M / rows_per_sample][chunks][gn_groups][2] fp32 = (sum, sum of squares) of the fp16-rounded outputs of each (sample, row
     * chunk, group) -- the layout fd_groupnorm_apply_parts_f16 / fd_groupnorm_fold_linear_parts_f16 consume, `chunks` =
     * fd_gemm_gn_parts_chunks(desc).  At the 64x64 level of `unet(...)` (reference pipeline/guide.py:56-58) the GroupNorms behind a
     * convolution (norm2 behind conv1, the transformer's input norm behind conv2) then need no statistics pass over the tensor the
     * convolution has just written.  Honoured where one tile spans the row and lies in one sample (N == 320, the 256x320 / 128x320
     * tiles, lean epilogue, act NONE, batch 1, no split-K); FD_ESHAPE otherwise -- ask fd_gemm_gn_parts_chunks first.  Uses gn_groups.
     * gn_part_chunks: the chunk count the buffer was sized for (what fd_gemm_gn_parts_chunks answered); a launch whose tile would write
     * another count -- a forced `tile` -- is refused instead of writing past the buffer.  NULL = off. */
    float* gn_part_out;
    int32_t gn_part_chunks;
    /* (ABI 11) TRANSPOSED TAIL: columns n >= trans_n0 of the output are stored transposed into C2 -- [sample][N - trans_n0][trans_ld]
     * (row n - trans_n0, column m within the sample; trans_ld / trans_sample_stride as for trans_out) -- while columns n < trans_n0 go to C
     * [M][ldc] as usual.  The self-attention of diffusers' BasicTransformerBlock.attn1 inside `unet(...)` (reference pipeline/guide.py:56-58)
     * projects q, k and v from the same LayerNorm'd rows: with the weights stacked [q | k | v] this is ONE launch that reads the hidden
     * states once and writes q|k row-major and V^T in the layout fd_attention_f16 takes, instead of two launches over the same rows.
     * Linear GEMM with the LayerNorm fold (ln_stats), act NONE, no residual / bias2 / batch; M %% 128 == 0, N and trans_n0 multiples of
     * 160, rows_per_sample %% 32 == 0, trans_ld %% 8 == 0.  0 = off. */
    int32_t trans_n0;
    void* C2;
    /* (ABI 11, EXPERIMENTAL: measured and left off) in-launch split-K reduction on the ping-pong tiles: sk_sync =
     * 2 x (output tiles) zero-initialised uint32 (arrival / departure counters, left zero by every launch).  Each K-slice workgroup publishes its
     * fp32 slab, arrives (agent-scope release), waits for the tile's other slices (acquire, bounded spin) and finishes 1/split_k of the tile's rows
     * in the fixed slice order -- the bits of the finish launch, without it.  Needs every workgroup of the launch RESIDENT at once
     * (tiles x split_k <= CUs, one per CU: refused otherwise) and nothing else competing for the CUs: two such launches from two processes
     * sharing a GPU can starve each other, which is why the product does not use it.  NULL = the finish launch. */
    void* sk_sync;
    /* (ABI 11) LayerNorm fold fed with the producer's PARTIAL sums: ln_stats_parts = k in {2, 4, 8} > 0 makes ln_stats the raw slabs
     * [k][ln_stats_rows][2] (sum, sum of squares per 160-column tile) that a producer GEMM wrote through ln_stats_out, instead of the finished
     * (rstd, -mean rstd) pairs: every tile of this launch finalises its own rows into LDS (the arithmetic of fd_ln_finalize_stats_f32, bit for
     * bit, over K columns with ln_fold_eps) -- the finalise launch between a transformer block's producer and consumer GEMMs inside `unet(...)`
     * (reference pipeline/guide.py:56-58) disappears.  One-tile LDS-DMA and ping-pong kernels (the persistent form is not used then). 0 = off. */
    int32_t ln_stats_parts, ln_stats_rows;
    float ln_fold_eps; /* 0 = 1e-5 */
    /* (ABI 12) residual_rows > 0: output row m adds residual row m %% residual_rows -- the residual of a GEMM whose rows are `rep` replicas of
     * a shared prefix (the classifier-free-guidance fan-out inside `unet(...)`, reference pipeline/guide.py:56-58: both halves of the batch read
     * the same hidden states until the first cross-attention) without materialising the replicas.  Linear GEMM, batch 1,
     * M %% residual_rows == 0 and residual_rows a multiple of the tile's rows (256 covers every tile but the 288-row one).  0 = row m. */
    int32_t residual_rows;
} fd_gemm_desc;

int fd_gemm_f16(const fd_gemm_desc* desc, void* stream);
/* > 0 when fd_gemm_f16 will honour ln_stats_out for an [M][N] fp16 output (row stride ldc, residual row stride ldr
 * or 0) with the library's current settings: 1 = the finished (rstd, -mean rstd) pairs, k > 1 = k slabs of partial
 * sums for fd_ln_finalize_stats_f32; 0: run fd_ln_row_stats_f16 on the output instead. */
int fd_gemm_can_emit_row_stats(int M, int N, int K, int ldc, int ldr);
/* The tile id and split-K factor fd_gemm_f16 would launch `d` with -- the same argument checks and cost model (reference call
 * site: every linear / convolution behind pipeline/guide.py:56-58), nothing launched, no HIP call: usable without a device.
 * Tile ids as in fd_gemm_desc.tile (30..33: the ping-pong kernels of gemm_pp.hip; -7: the 128x128 generic kernel with the
 * LayerNorm fold compiled in).  Pointers are only tested for NULL / alignment. */
int fd_gemm_plan(const fd_gemm_desc* d, int* tile, int* split_k);
/* (ABI 11) 1 when a split-K launch of an [M][N] output with `rows_per_sample` rows per sample and split factor `split_k` can take
 * fd_gemm_desc.gn_out with `groups` groups (the slab of one sample x a few groups fits the finish kernel's registers); 0: run
 * fd_groupnorm_nhwc_f16 on the output instead.  Host logic only. */
int fd_gemm_can_fuse_groupnorm(int M, int N, int rows_per_sample, int groups, int split_k);
/* (ABI 11) Row chunks per sample of fd_gemm_desc.gn_part_out for `d` (with d->gn_groups set): > 0 when fd_gemm_f16 will honour it with the
 * tile the rule picks, 0: run the statistics pass on the output instead.  Host logic only. */
int fd_gemm_gn_parts_chunks(const fd_gemm_desc* d);

/* Flash attention forward (scores never leave registers).  Q [B][n_q][ldq], K [B][n_k][ldk]
 * with head h at column h*head_dim; Vt [B][heads*head_dim][ldvt] is V transposed (keys
 * contiguous; columns n_k..ldvt-1 must be finite, e.g. zero); O [B][n_q][ldo].
 * head_dim % 8 == 0, <= 160.  scale <= 0 means head_dim^-0.5.  causal: key <= query.
 * q_prescaled != 0: Q already carries scale*log2(e) (e.g. folded into the q projection's
 * weights), so K.Q^T is used directly as the base-2 logit and `scale` is ignored (head_dim <= 80
 * or 160 only). */
typedef struct fd_attention_desc {
    const void* Q;
    const void* K;
    const void* Vt;
    void* O;
    int64_t q_sample_stride, k_sample_stride, vt_sample_stride, o_sample_stride;
    int32_t ldq, ldk, ldvt, ldo;
    int32_t batch, heads, n_q, n_k, head_dim;
    int32_t causal;
    float scale;
    int32_t q_prescaled;
} fd_attention_desc;

int fd_attention_f16(const fd_attention_desc* desc, void* stream);

/* Fused front half of the UNet's cross-attention at the 64x64 level (8 heads x 40 channels): the
 * LayerNorm-fold q projection and softmax(Q K^T) V over the step-invariant text context (65..80
 * keys) in ONE launch -- replaces the q-projection fd_gemm_f16 launch and the fd_attention_f16
 * launch of diffusers' BasicTransformerBlock.attn2 inside `unet(...)` (reference
 * pipeline/guide.py:56-58); the query matrix never goes to HBM.
 * fd_xattn_pack_kv_f16 packs the context's K [samples][n_keys][ldk] / V^T [samples][heads*head_dim][ldvt]
 * (the outputs of the to_k / to_v projections, computed once per context) into per-sample "images"
 * in MFMA fragment order, fd_xattn_image_bytes(heads, head_dim) bytes each (0 = unsupported shape). */
int64_t fd_xattn_image_bytes(int heads, int head_dim);
int fd_xattn_pack_kv_f16(const void* K, const void* Vt, void* k_image, void* v_image, int samples, int n_keys,
                         int heads, int head_dim, int ldk, int ldvt, int64_t k_sample_stride,
                         int64_t vt_sample_stride, void* stream);
typedef struct fd_xattn_desc {
    const void* x;          /* fp16 [M][ldx] UN-normalised hidden states */
    const void* wq;         /* fp16 [C][ldw] q weights with the LayerNorm gain and head_dim^-0.5 log2(e) folded in */
    const float* bias;      /* [C] folded bias (W beta), see fd_gemm_desc.ln_stats */
    const float* ln_colsum; /* [C] */
    const float* ln_stats;  /* [M][2] (rstd, -mean rstd) of the rows of x */
    const void* k_image;    /* [n_rep * M / rows_per_sample] images: replica r of sample b at index r * (M / rows_per_sample) + b */
    const void* v_image;
    void* out;              /* fp16 [n_rep * M][ldo]: attention output (heads concatenated), replica-major */
    int32_t M, ldx, ldw, ldo;
    int32_t rows_per_sample; /* query rows per sample; a multiple of 256 */
    int32_t n_rep;           /* context replicas sharing the same queries (CFG fan-out of a shared prefix); >= 1 */
    int32_t n_keys, heads, head_dim;
    /* (ABI 12) ln_stats_parts = k in {2, 4, 8}: ln_stats holds the k partial slabs [k][M][2] (sum, sum of squares per 160-column tile) that the
     * producer GEMM wrote through fd_gemm_desc.ln_stats_out; every tile finalises its own rows (fd_ln_finalize_stats_f32's arithmetic, bit for
     * bit, over heads * head_dim columns with ln_fold_eps) -- as fd_gemm_desc.ln_stats_parts does for the GEMM consumers.  0 = finished pairs. */
    int32_t ln_stats_parts;
    float ln_fold_eps;       /* 0 = 1e-5 */
} fd_xattn_desc;
int fd_xattn_q_f16(const fd_xattn_desc* desc, void* stream);

/* ------------------------------------------------------------------------------------
 * Normalisation / softmax (HBM-bound)
 * ---------------------------------------------------------------------------------- */
int64_t fd_groupnorm_workspace_floats(int B, int G);
/* GroupNorm(G, eps) [+ SiLU] on NHWC fp16 x [B][HW][C] -> y (may alias x). ws: scratch. */
int fd_groupnorm_nhwc_f16(const void* x, void* y, const float* gamma, const float* beta,
                          float* ws, int B, int HW, int C, int G, float eps, int silu,
                          void* stream);
/* Same with a row stride ldx >= C (in elements) on the INPUT: x is a column slice of a wider
 * [B*HW][ldx] matrix (the UNet writes skip tensors straight into the decoder's concatenation
 * buffers, pipeline/flex.py's UNet call -> diffusers' torch.cat of the skip connections). y is
 * contiguous [B][HW][C] and must not alias x. */
int fd_groupnorm_nhwc_ld_f16(const void* x, int ldx, void* y, const float* gamma, const float* beta,
                             float* ws, int B, int HW, int C, int G, float eps, int silu,
                             void* stream);
/* (ABI 11) The apply pass alone, from partial sums parts [B][chunks][G][2] that the PRODUCER of x wrote (fd_gemm_desc.gn_part_out):
 * y = GroupNorm(x)(+SiLU) without the statistics pass over x. */
int fd_groupnorm_apply_parts_f16(const void* x, int ldx, void* y, const float* gamma, const float* beta, const float* parts,
                                 int chunks, int B, int HW, int C, int G, float eps, int silu, void* stream);
/* GroupNorm folded into the linear layer that consumes it (the transformer block's norm -> proj_in, diffusers
 * Transformer2DModel.norm / proj_in inside the `unet(...)` call of reference pipeline/guide.py:56-58):
 *   proj_in(GN(x))[m][n] = sum_c w_out[b][n][c] x[m][c] + (bias[n] + (W beta)[n] - sum_g mean_{b,g} sum_{c in g} w_out[b][n][c]),
 *   w_out[b][n][c] = fp16(W[n][c] gamma_c rstd_{b,g(c)})
 * One statistics pass over x [B][HW][ldx] (the first launch of fd_groupnorm_nhwc_ld_f16), then per sample b the scaled
 * weights w_out [B][N][C] fp16 and bias_out [B][N] fp32 from the model constants wg = W diag(gamma) (fp16 [N][C]) and
 * biasf = bias + W beta (fp32 [N]).  (ABI 9) the mean term is summed over the ROUNDED w_out, so a group mean cancels
 * exactly whatever its size relative to the group's spread.  The normalised activation is never written: the consumer is
 * fd_gemm_f16 with batch = B, batch_stride_w = N * C, batch_stride_bias = N on x itself. */
int fd_groupnorm_fold_linear_f16(const void* x, int ldx, float* ws, int B, int HW, int C, int G, float eps,
                                 const void* wg, const float* biasf, int N,
                                 void* w_out, float* bias_out, void* stream);
/* (ABI 11) The same from the producer's partial sums (fd_gemm_desc.gn_part_out), without reading x at all. */
int fd_groupnorm_fold_linear_parts_f16(const float* parts, int chunks, int B, int HW, int C, int G, float eps,
                                       const void* wg, const float* biasf, int N, void* w_out, float* bias_out, void* stream);
/* LayerNorm over the last dim of fp16 x [rows][ldx] -> fp16 (or fp32) y [rows][ldy]. */
int fd_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta, int rows,
                     int C, int ldx, int ldy, float eps, int out_f32, void* stream);
/* Per-row LayerNorm statistics of fp16 x [rows][ldx] (exact two-pass, fp32): stats[row] = (rstd,
 * -mean * rstd) with rstd = 1 / sqrt(var + eps); consumed by fd_gemm_desc.ln_stats. */
int fd_ln_row_stats_f16(const void* x, float* stats, int rows, int C, int ldx, float eps, void* stream);
/* partials [n_tiles][M][2] (fd_gemm_desc.ln_stats_out of an N > 320 producer) -> stats [M][2] = (rstd, -mean rstd)
 * over the N columns, eps as fd_ln_row_stats_f16 (one-pass E[x^2] - mean^2 in fp32, fixed summation order). */
int fd_ln_finalize_stats_f32(const float* partials, float* stats, int M, int N, int n_tiles, float eps, void* stream);
/* In-place softmax(scale * x) over the first N columns of fp16 x [rows][ld]. */
int fd_softmax_rows_f16(void* x, int rows, int N, int ld, float scale, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout / elementwise helpers
 * ---------------------------------------------------------------------------------- */
/* NCHW fp32 [B][C][HW] * scale -> NHWC fp16 [rep*B][HW][c_pad] (channels >= C zeroed). */
int fd_nchw_f32_to_nhwc_f16(const float* x, void* y, int B, int C, int HW, int rep, int c_pad,
                            float scale, void* stream);
/* NHWC fp32 [B][HW][ld] -> NCHW fp32 [B][C][HW]: y = x*a + b, optionally clamped to [0,1]. */
int fd_nhwc_f32_to_nchw_f32(const float* x, float* y, int B, int C, int HW, int ld, float a,
                            float b, int clamp01, void* stream);
/* im2col for convolutions with fewer than 64 input channels; out [B*Ho*Wo][k_pad]. */
/* (ABI 12) 3x3 / stride 1 / pad 1 convolution of a NARROW input (Cin <= 4) read straight from the fp32 NCHW tensor -- the UNet's conv_in
 * (reference pipeline/guide.py:56-58, the first layer of `unet(...)`; diffusers UNet2DConditionModel.conv_in):
 *   y[(b H + i) W + j][co] = bias[co] + sum_{ky, kx, ci} half(x[b][ci][i + ky - 1][j + kx - 1] * scale) * w[co][ky][kx][ci]
 * (fp16 x fp16 products, fp32 accumulation, fp16 NHWC rows of stride ldy).  w: [Cout][3][3][4] fp16, channels >= Cin zero.  rep2 > 0 also
 * writes `rep2` replicas of the output to y2 (row stride ldy2, replica r at rows r B H W ...): the classifier-free-guidance fan-out's copy
 * of the first skip tensor.  One launch instead of fd_nchw_f32_to_nhwc_f16 + fd_im2col_f16 + fd_gemm_f16 (+ fd_repeat_rows_f16). */
int fd_conv3x3_narrow_f16(const float* x, const void* w, const float* bias, void* y, int ldy, void* y2, int ldy2, int rep2,
                          int B, int Cin, int H, int W, int Cout, float scale, void* stream);
int fd_im2col_f16(const void* x, void* y, int B, int Hi, int Wi, int Cin, int Ho, int Wo, int KH,
                  int KW, int stride, int pad_t, int pad_l, int k_pad, void* stream);
int fd_concat_channels_f16(const void* a, const void* b, void* out, int64_t M, int Ca, int Cb,
                           void* stream);
/* Classifier-free guidance (pipeline/guide.py:59-63) fused with the DDIM update the
 * reference obtains from scheduler.step (pipeline/flex.py:280-285), eta = 0:
 *   eps = u + g (t - u); x0 = (x - c1 eps)/c2; x <- c3 x0 + c4 eps
 * with c1 = sqrt(1-a_t), c2 = sqrt(a_t), c3 = sqrt(a_prev), c4 = sqrt(1-a_prev).
 * x: NCHW fp32 [B][C][HW] updated in place when do_step; eps_nhwc: UNet output
 * [(cfg?2:1)*B][HW][ld] fp32 (unconditional half first); eps_out (optional) NCHW fp32. */
int fd_cfg_ddim_step_f32(float* x, const float* eps_nhwc, float* eps_out, int B, int C, int HW,
                         int ld, int cfg, float guidance, float c1, float c2, float c3, float c4,
                         int v_prediction, int do_step, void* stream);
/* out = a*x + b*y; with exp_half_x: out = exp(0.5 x) * y * b (VAE posterior sampling). */
int fd_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b,
                 int exp_half_x, void* stream);
int fd_embed_tokens_f16(const int64_t* ids, const void* tok_emb, const void* pos_emb, void* out,
                        int B, int L, int D, int vocab, void* stream);
int fd_vit_assemble_f16(const void* patches, const void* cls, const void* pos, void* out, int B,
                        int T, int D, void* stream);
/* diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0): t fp32 -> fp16 [B][dim]; sample b reads
 * t[b * t_stride], t_stride 1 (one timestep per sample) or 0 (one device scalar for the batch: the
 * denoising loop's timestep, refreshed between replays of a launch plan). */
int fd_timestep_embedding_f16(const float* t, int t_stride, void* out, int B, int dim, void* stream);
/* dst[r][0..cols) = src[r][0..cols) for fp16 matrices with row strides lds / ldd (elements);
 * cols, lds, ldd multiples of 8, pointers 16-byte aligned.  The CFG fan-out copies of the UNet. */
int fd_copy2d_f16(const void* src, int lds, void* dst, int ldd, int64_t rows, int cols, void* stream);
/* (ABI 11) dst[r * rows + i][0..cols) = src[i][0..cols) for r < rep, one launch: the CFG fan-out of the shared UNet prefix (B samples -> rep * B;
 * the reference feeds the replicated batch from the start, pipeline/guide.py:46-58).  Same alignment rules as fd_copy2d_f16. */
int fd_repeat_rows_f16(const void* src, int lds, void* dst, int ldd, int64_t rows, int cols, int rep, void* stream);
/* CompositeGuide region blend (reference composition/guide.py:86-98), NCHW fp32 [C][H][W]:
 * dst[:, oy:oy+sh, ox:ox+sw] += blend * (src - dst) on the same box.  oy, ox >= 0: the host
 * resolves Python's slice semantics (negative starts count from the end of the axis) first;
 * the box is clipped to the canvas. */
int fd_region_blend_f32(float* dst, const float* src, int C, int H, int W, int oy, int ox, int sh,
                        int sw, float blend, void* stream);
int fd_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream);
int fd_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLEXDIFFUSE_HIP_H */
