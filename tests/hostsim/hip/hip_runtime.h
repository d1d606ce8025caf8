/* tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for <hip/hip_runtime.h> that lets g++ compile the engine's .hip sources for the host
 * and run a launch as a plain loop over the grid, one "lane" at a time.  It exists so that the CPU
 * test suite can drive the very kernel source that ships (through the same C-ABI) against the
 * oracle when there is no GPU at hand.  It is only valid for kernels whose lanes do not talk to
 * each other: no LDS, no barriers, and shuffles only in the "sum a per-lane counter over the
 * wavefront" idiom (a lone lane sees 0 from every other lane).  That is the Raft and EPaxos
 * engines; the MultiPaxos engine and the RS kernels are wave-cooperative and are NOT covered.
 * It says nothing about how a kernel behaves on the GPU (memory model, occupancy, speed) --
 * tests/test_*_gpu.py do that -- and the package itself never loads a library built with it.
 */
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipsim_idx { unsigned x, y, z; };
inline thread_local hipsim_idx blockIdx, threadIdx, blockDim, gridDim;

typedef void *hipStream_t;
typedef enum { hipSuccess = 0, hipErrorOutOfMemory = 2 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "out of memory"; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
inline hipError_t hipGetLastError(void) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 0; return hipSuccess; }

inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline unsigned __lane_id(void) { return 0; }
template <typename T> inline T __shfl_xor(T, int) { return T(0); }      /* the other lanes are idle: they hold 0 */
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }

#define hipLaunchKernelGGLInternal(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)        \
    do {                                                                                                 \
        const dim3 hs_g = (numBlocks), hs_b = (numThreads);                                              \
        (void)(memPerBlock); (void)(streamId);                                                           \
        gridDim = {hs_g.x, hs_g.y, hs_g.z}; blockDim = {hs_b.x, hs_b.y, hs_b.z};                         \
        for (unsigned hs_i = 0; hs_i < hs_g.x; hs_i++)                                                   \
            for (unsigned hs_t = 0; hs_t < hs_b.x; hs_t++) {                                             \
                blockIdx = {hs_i, 0, 0}; threadIdx = {hs_t, 0, 0};                                       \
                kernelName(__VA_ARGS__);                                                                 \
            }                                                                                            \
    } while (0)
#define hipLaunchKernelGGL(kernelName, ...) hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__)
