// Internal helpers shared by the translation units of libsummerset_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/summerset_hip.h"

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
