/* tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for <hip/hip_runtime.h> that lets the host compiler (clang++) compile the engine's .hip
 * sources for the host, so that the CPU test suite can drive the very kernel source that ships --
 * through the same C-ABI and Python mirror -- against the oracle when there is no GPU at hand.
 * A launch runs block after block; inside a block every lane is a fiber (tests/hostsim/hipsim_rt.cpp):
 *   - cross-lane operations (__shfl*, __ballot, __all, __any) park the lane until the lanes of its
 *     wavefront (64) that execute the operation with it have arrived; __syncthreads likewise per block;
 *   - every load / store outside the lane's own stack parks it too (the sources are compiled with
 *     -fsanitize=thread purely for the call that puts in front of each access; no sanitizer runtime is
 *     linked), and parked lanes are let go in program order (source position of the access, from the
 *     debug info), which gives a wavefront the lock-step memory order the kernels are written for
 *     ("every lane reads X, then lane 0 overwrites X");
 *   - atomics are uninterrupted; "device memory" is host memory; streams and events are no-ops.
 *
 * What it shows: the logic of a kernel -- indexing, protocol rules, cross-lane choreography, LDS hand-offs
 * -- is what the oracle says.  What it cannot show: anything about the device (memory model between
 * wavefronts, occupancy, speed, what the gfx950 compiler does); tests/test_*_gpu.py do that.  The package
 * itself never loads a library built with it.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)::hipsim::dyn_shared();

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipsim_idx { unsigned x, y, z; };
extern thread_local hipsim_idx blockIdx, threadIdx, blockDim, gridDim;
static const int warpSize = 64;

namespace hipsim {
enum { K_SHFL = 1, K_BALLOT, K_ALL, K_ANY, K_BARRIER };
/* called by a lane: park at a cross-lane operation, come back with its result */
uint64_t collective(int kind, const void *site, const void *frame, uint64_t val, int arg);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
void *dyn_shared();
/* memory no kernel may touch (the gaps -DSMR_ARENA_GUARD leaves between the arrays of an arena); a kernel
 * access into one aborts with the access site.  dev_free lifts the marks inside the freed block. */
void poison(const void *p, size_t n);
void *dev_malloc(size_t n);
void dev_free(void *p);
}  // namespace hipsim
extern "C" unsigned long hipsim_partial_wave_ops(void);   /* wave operations that met with live lanes elsewhere */

typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef enum { hipSuccess = 0, hipErrorOutOfMemory = 2 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "out of memory"; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = hipsim::dev_malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { hipsim::dev_free(p); return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemset2D(void *p, size_t pitch, int v, size_t w, size_t h) { for (size_t r = 0; r < h; r++) memset((char *)p + r * pitch, v, w); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError(void) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 0; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }   // (the low 24 bits of each: v_mul_u32_u24)
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline unsigned __lane_id(void) { return threadIdx.x & 63u; }
inline void __threadfence(void) {}
inline void __threadfence_block(void) {}
/* not instrumented: no parking point inside, so the read-modify-write is atomic among the fibers */
#define HIPSIM_ATOMIC inline __attribute__((no_sanitize("thread")))
template <typename T, typename U> HIPSIM_ATOMIC T atomicAdd(T *p, U v) { T o = *p; *p = (T)(o + v); return o; }
template <typename T, typename U> HIPSIM_ATOMIC T atomicOr(T *p, U v) { T o = *p; *p = (T)(o | v); return o; }
template <typename T, typename U> HIPSIM_ATOMIC T atomicMax(T *p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <typename T, typename U> HIPSIM_ATOMIC T atomicMin(T *p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <typename T, typename U> HIPSIM_ATOMIC T atomicExch(T *p, U v) { T o = *p; *p = (T)v; return o; }

/* noinline: the return address identifies the call site, which is how lanes that execute the same
 * operation find each other.  convergent: what the device compiler knows about these operations -- the
 * host compiler must not make a call depend on a condition it did not depend on (unswitching, jump
 * threading), or lanes on different copies would never meet.  (noduplicate would say the same for
 * clones, but clang then ignores noinline.) */
#define HIPSIM_XLANE __attribute__((noinline, convergent))
template <typename T> HIPSIM_XLANE T __shfl(T v, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffles of up to 8 bytes");
    uint64_t x = 0;
    memcpy(&x, &v, sizeof(T));
    // width < 64: the source lane is taken inside the caller's own width-aligned segment (HIP / CUDA semantics)
    const int lane = (int)(threadIdx.x & 63u), tgt = (lane & ~(width - 1)) | (src & (width - 1));
    const uint64_t r = hipsim::collective(hipsim::K_SHFL, __builtin_return_address(0), __builtin_frame_address(1), x, tgt & 63);
    T o;
    memcpy(&o, &r, sizeof(T));
    return o;
}
template <typename T> HIPSIM_XLANE T __shfl_xor(T v, int mask, int width = 64) {
    static_assert(sizeof(T) <= 8, "shuffles of up to 8 bytes");
    (void)width;
    uint64_t x = 0;
    memcpy(&x, &v, sizeof(T));
    const uint64_t r = hipsim::collective(hipsim::K_SHFL, __builtin_return_address(0), __builtin_frame_address(1), x, (int)((threadIdx.x ^ (unsigned)mask) & 63u));
    T o;
    memcpy(&o, &r, sizeof(T));
    return o;
}
HIPSIM_XLANE inline unsigned long long __ballot(int pred) {
    return hipsim::collective(hipsim::K_BALLOT, __builtin_return_address(0), __builtin_frame_address(1), pred ? 1 : 0, 0);
}
HIPSIM_XLANE inline int __all(int pred) {
    return (int)hipsim::collective(hipsim::K_ALL, __builtin_return_address(0), __builtin_frame_address(1), pred ? 1 : 0, 0);
}
HIPSIM_XLANE inline int __any(int pred) {
    return (int)hipsim::collective(hipsim::K_ANY, __builtin_return_address(0), __builtin_frame_address(1), pred ? 1 : 0, 0);
}
HIPSIM_XLANE inline void __syncthreads(void) {
    (void)hipsim::collective(hipsim::K_BARRIER, nullptr, nullptr, 0, 0);
}

#define hipLaunchKernelGGLInternal(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)        \
    do {                                                                                                 \
        (void)(streamId);                                                                                \
        ::hipsim::launch((numBlocks), (numThreads), (size_t)(memPerBlock), [&]() { kernelName(__VA_ARGS__); }); \
    } while (0)
#define hipLaunchKernelGGL(kernelName, ...) hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__)
