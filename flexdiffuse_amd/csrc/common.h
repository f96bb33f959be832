// Shared host-side helpers for libflexdiffuse_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/flexdiffuse_hip.h"

void fd_set_error(const char* fmt, ...);

#define FD_CHECK_ARG(cond, code, ...)                  \
    do {                                               \
        if (!(cond)) {                                 \
            fd_set_error(__VA_ARGS__);                 \
            return (code);                             \
        }                                              \
    } while (0)

#define FD_CHECK_LAUNCH(name)                                                    \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            fd_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_)); \
            return FD_EHIP;                                                      \
        }                                                                        \
    } while (0)

#define FD_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fd_set_error("%s failed: %s", #call, hipGetErrorString(e_));             \
            return FD_EHIP;                                                          \
        }                                                                            \
    } while (0)

static inline int fd_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- launch plan (fd_plan_*, runtime.hip): while the calling thread records, every launch entry
// point appends a by-value copy of its own call (stream left open) to the plan.  FD_PLAN(call)
// sits at the top of an entry point; `fd_s_` names the stream the replay passes in.
#ifdef __cplusplus
#include <functional>
bool fd_plan_recording();
void fd_plan_push(std::function<int(void*)> op);
#define FD_PLAN(...)                                                            \
    do {                                                                        \
        if (fd_plan_recording())                                                \
            fd_plan_push([=](void* fd_s_) -> int { return __VA_ARGS__; });      \
    } while (0)
#endif

// optional per-launch HIP-event timing of a kernel family (bench.py roofline leg)
// `work`: the ALGORITHMIC FLOPs / bytes of the op the launch implements; `executed` (< 0: same as work): what the
// kernel really issues (the parity-decomposed upsample convolution runs 4/9 of its 3x3 MACs)
// `tag`: an identity of the launch's shape / kernel choice (0 = none): bench.py groups the sampled brackets by
// (family, tag, work) and takes each group's MEDIAN, so one host stall inside one bracket cannot move a family's sum
void fd_prof_begin(int family, hipStream_t s, double work, double executed = -1.0, unsigned tag = 0);
void fd_prof_end(int family, hipStream_t s);
#ifdef __cplusplus
// bracket of one entry point of the FD_FAMILY_OTHER kind (layout / elementwise / LayerNorm / guidance launches): closes on
// every return path
struct FdProfScope {
    int family;
    hipStream_t st;
    FdProfScope(int f, void* stream, double bytes, unsigned tag = 0) : family(f), st((hipStream_t)stream) { fd_prof_begin(f, st, bytes, -1.0, tag); }
    ~FdProfScope() { fd_prof_end(family, st); }
};
static inline unsigned fd_tag(unsigned a, unsigned b = 0, unsigned c = 0, unsigned d = 0, unsigned e = 0, unsigned f = 0) {
    unsigned h = 2166136261u;
    const unsigned v[6] = {a, b, c, d, e, f};
    for (int i = 0; i < 6; ++i) h = (h ^ v[i]) * 16777619u;
    return h | 1u;
}
#endif

#ifdef __cplusplus
#include <atomic>
// One-time per-(kernel instantiation, DEVICE) setup such as hipFuncSetAttribute(MaxDynamicSharedMemorySize): a function
// attribute belongs to the device's copy of the code object, so a process that drives a second GPU must set it there too.
// `mask` is a static of the instantiation; true exactly once per device (thread-safe).
static inline bool fd_first_on_device(std::atomic<unsigned long long>* mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    const unsigned long long bit = 1ull << (dev & 63);
    return (mask->fetch_or(bit) & bit) == 0;
}
#endif

typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef half_t half4 __attribute__((ext_vector_type(4)));
typedef half_t half2v __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fd_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float fd_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
