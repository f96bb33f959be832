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

#define FD_ABI_VERSION 1

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
int fd_prof_enable(int on);
/* Synchronises the recorded events of `family`, returns summed elapsed ms, declared work
 * (FLOPs for MFMA families, algorithmic bytes for HBM families) and launch count, and
 * forgets those records. Host pointers. */
int fd_prof_collect(int family, double* total_ms, double* total_work, int64_t* launches);

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

#ifdef __cplusplus
}
#endif
#endif /* FLEXDIFFUSE_HIP_H */
