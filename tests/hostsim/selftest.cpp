// tests/hostsim/selftest.cpp -- what the emulation promises, checked on tiny kernels (built and run by
// tests/test_hostsim.py with the same flags as the library).  `selftest guard` must die on the guard gap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

// divergent callers of a wave-aggregated append (the shape of Lane::record_commit) + a counter flush
__device__ __forceinline__ void append(unsigned me, unsigned *cnt, unsigned *list, unsigned item) {
    const int lane = __lane_id();
    for (unsigned long long todo = __ballot(1); todo;) {
        const int first = __ffsll((long long)todo) - 1;
        const unsigned long long mask = __ballot(me == __shfl(me, first)) & todo;
        todo &= ~mask;
        if (!((mask >> lane) & 1ull)) continue;
        unsigned int base = 0;
        if (lane == first) base = atomicAdd(cnt + me, (unsigned int)__popcll(mask));
        base = __shfl(base, first);
        list[me * 256 + base + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull))] = item;
        break;
    }
}
__global__ void k_append(unsigned *cnt, unsigned *list, unsigned *ctr, unsigned G) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned n = 0;
    if (g < G && g % 3 != 0) { append(g & 1, cnt, list, g); n = 1; }
    unsigned x = n;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if (__lane_id() == 0 && x) atomicAdd(ctr, x);
}

// lock-step memory order: every lane reads the flag, THEN lane 0 sets it; on the device no lane sees it set
__global__ void k_read_then_write(unsigned *flag, unsigned *seen) {
    const unsigned t = threadIdx.x;
    const unsigned f = flag[blockIdx.x];
    if ((t & 63) == 0) flag[blockIdx.x] = 1;
    seen[blockIdx.x * blockDim.x + t] = f;
}

// LDS hand-off across the wavefronts of a block
__global__ void k_lds(unsigned *out) {
    __shared__ unsigned sh[256];
    sh[threadIdx.x] = threadIdx.x * 3;
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = sh[255 - threadIdx.x];
}

// a loop with a wave operation per iteration and lanes that leave early
__global__ void k_loop(unsigned *out) {
    const unsigned t = threadIdx.x;
    unsigned acc = 0;
    for (unsigned i = 0; i < (t & 7u) + 1; i++) acc += (unsigned)__popcll(__ballot(1));
    out[t] = acc;
}

__global__ void k_oob(unsigned *a, unsigned n) { a[n + threadIdx.x] = 1; }

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "guard")) {
        unsigned *a;
        hipMalloc((void **)&a, 64 * 4 + 4096);
        hipsim::poison(a + 64, 4096);
        hipLaunchKernelGGL(k_oob, dim3(1), dim3(64), 0, 0, a, 63u);      // thread 1 lands in the gap
        printf("guard not hit\n");
        return 0;
    }
    int bad = 0;
    for (unsigned G : {1u, 5u, 64u, 70u, 300u}) {
        unsigned cnt[2] = {0, 0}, ctr = 0;
        static unsigned list[512];
        memset(list, 0xFF, sizeof(list));
        hipLaunchKernelGGL(k_append, dim3((G + 127) / 128), dim3(128), 0, 0, cnt, list, &ctr, G);
        unsigned want[2] = {0, 0};
        for (unsigned g = 0; g < G; g++) if (g % 3) want[g & 1]++;
        bool ok = cnt[0] == want[0] && cnt[1] == want[1] && ctr == want[0] + want[1];
        for (unsigned me = 0; me < 2 && ok; me++)                        // per list: ascending within a wavefront's append
            for (unsigned i = 0; i < cnt[me]; i++) ok = ok && list[me * 256 + i] % 3 != 0 && (list[me * 256 + i] & 1) == me;
        if (!ok) { printf("append G=%u: cnt %u %u ctr %u want %u %u\n", G, cnt[0], cnt[1], ctr, want[0], want[1]); bad++; }
    }
    {
        unsigned flag[3] = {0, 0, 0};
        static unsigned seen[3 * 128];
        hipLaunchKernelGGL(k_read_then_write, dim3(3), dim3(128), 0, 0, flag, seen);
        // within a wavefront nobody sees the flag its lane 0 sets later; the block's other wavefront may
        for (unsigned b = 0; b < 3; b++)
            for (unsigned t = 0; t < 64; t++)
                if (seen[b * 128 + t] != 0) { printf("read-then-write: block %u lane %u saw %u\n", b, t, seen[b * 128 + t]); bad++; break; }
        if (!(flag[0] == 1 && flag[1] == 1 && flag[2] == 1)) { printf("read-then-write: flags not set\n"); bad++; }
    }
    {
        static unsigned out[2 * 256];
        hipLaunchKernelGGL(k_lds, dim3(2), dim3(256), 0, 0, out);
        for (unsigned i = 0; i < 512; i++)
            if (out[i] != (255 - (i & 255)) * 3) { printf("lds: out[%u] = %u\n", i, out[i]); bad++; break; }
    }
    {
        static unsigned out[64];
        hipLaunchKernelGGL(k_loop, dim3(1), dim3(64), 0, 0, out);
        // iteration i is executed by the lanes with (t & 7) >= i: 64 - 8 i of them
        for (unsigned t = 0; t < 64; t++) {
            unsigned want = 0;
            for (unsigned i = 0; i <= (t & 7u); i++) want += 64 - 8 * i;
            if (out[t] != want) { printf("loop: lane %u got %u want %u\n", t, out[t], want); bad++; break; }
        }
    }
    printf(bad ? "FAILED\n" : "ok\n");
    return bad ? 1 : 0;
}
