// Internal helpers shared by the translation units of libsummerset_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/summerset_hip.h"

// a value every lane of the wavefront holds alike (e.g. the wavefront's index in its block), said to the compiler so that what
// is indexed by it is loaded through the scalar unit; the host pass (and tests/hostsim) see the plain value
#if defined(__HIP_DEVICE_COMPILE__)
#define SMR_WAVE_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#else
#define SMR_WAVE_UNIFORM(x) ((uint32_t)(x))
#endif

namespace smr {

void set_error(const std::string &msg);

inline int fail(int code, const std::string &msg) {
    set_error(msg);
    return code;
}

#define SMR_HIP_TRY(expr)                                                                 \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            return ::smr::fail(SMR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                 \
    } while (0)

// Event counters (commits, redirects, ...) are kept as SMR_CTR_SHARDS partial sums, one 64-byte line each: a kernel's
// wavefronts add to different lines, the host sums them when it reads.  One shared word per counter looked harmless and
// WAS the kernels' run time: ~15 ns per same-address atomic, 1024 wavefronts -> 15-30 us (profiles/round2/r2q: the Raft reply
// kernel went from 33 to 126 us when eight lanes per group made it 8192 wavefronts -- its atomics, not its loads).
constexpr uint32_t SMR_CTR_SHARDS = 256, SMR_CTR_STRIDE = 8;     // u64 words per shard
constexpr size_t SMR_CTR_WORDS = (size_t)SMR_CTR_SHARDS * SMR_CTR_STRIDE;
__device__ __forceinline__ void ctr_add(unsigned long long *base, int k, unsigned long long x) {
    const uint32_t shard = ((blockIdx.x + blockIdx.y * 37u) * 4u + (threadIdx.x >> 6)) & (SMR_CTR_SHARDS - 1u);
    atomicAdd(&base[(size_t)shard * SMR_CTR_STRIDE + (uint32_t)k], x);
}
// host: the n <= SMR_CTR_STRIDE counters behind `dev_base`, shards summed
inline hipError_t ctr_read(const unsigned long long *dev_base, int n, unsigned long long *out) {
    static thread_local unsigned long long buf[SMR_CTR_WORDS];
    hipError_t e = hipMemcpy(buf, dev_base, sizeof(buf), hipMemcpyDeviceToHost);
    for (int k = 0; k < n; k++) {
        out[k] = 0;
        for (uint32_t s2 = 0; s2 < SMR_CTR_SHARDS; s2++) out[k] += buf[(size_t)s2 * SMR_CTR_STRIDE + k];
    }
    return e;
}

// bump allocator over one hipMalloc'd arena: one allocation per engine object,
// every array 256-byte aligned so wave-wide accesses start on a cache line.
struct Arena {
    char *base = nullptr;
    size_t size = 0, used = 0;
    size_t reserve(size_t bytes) {
        size_t off = (used + 255) & ~size_t(255);
        used = off + bytes;
#ifdef SMR_ARENA_GUARD
        // kernel-source simulation only (tests/hostsim): an unowned gap behind every array, reported when a
        // kernel touches it, so an index that runs off the end of its array cannot quietly land in the neighbour
        if (base) hipsim::poison(base + used, SMR_ARENA_GUARD);
        used += SMR_ARENA_GUARD;
#endif
        return off;
    }
    template <typename T> T *at(size_t off) const { return (T *)(base + off); }  // C cast: T may carry an address space on the device pass
};

}  // namespace smr
