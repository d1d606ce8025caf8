// Microbenchmark (round 4), second part: what a dependent round costs when it also carries the engine's other access shapes.
// 512 blocks x 10 wavefronts as ep_cluster_tick_kernel; every round = one coalesced 16-byte load (the dependence) plus MODE:
//   0 nothing else                     1 + one coalesced 16 B store           2 + one scattered 4 B store (lanes 1280 B apart: hc)
//   3 + five scattered 4 B loads (hc)  4 + hc loads and the hc store          5 + scattered 8 B load and store (kv [key][G])
//   6 + three coalesced 16 B stores and the hc store (an acceptor's stores)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(640) void chase(u32x4 *__restrict__ p, uint32_t *__restrict__ hc, uint64_t *__restrict__ kv, uint32_t G, uint32_t RC, int K, uint32_t *out) {
    const uint32_t g = blockIdx.x * 128 + (threadIdx.x & 127u);
    const uint32_t rep = threadIdx.x >> 7;
    uint32_t x = 0, h = 12345u;
    uint32_t *myhc = hc + (size_t)rep * G * 320 + (size_t)g * 320;            // [g][64 keys][5]
    uint64_t *mykv = kv + (size_t)rep * 64 * G;                               // [key][G]
    for (int k = 0; k < K; k++) {
        h = h * 1664525u + 1013904223u;
        const uint32_t rc = rep * (RC / 5) + ((h >> 8) + x) % (RC / 5);
        const uint32_t key = ((h >> 4) ^ (g * 2654435761u >> 7)) & 63u;       // per lane
        uint32_t acc = p[rc * G + g].x;
        if (MODE == 3 || MODE == 4) for (int i = 0; i < 5; i++) acc += myhc[key * 5 + i];
        if (MODE == 5) acc += (uint32_t)mykv[key * G + g];
        x = acc;
        if (MODE == 1 || MODE == 6) p[((rc + 7) % RC) * G + g] = (u32x4){x, 1, 2, 3};
        if (MODE == 6) { p[((rc + 9) % RC) * G + g] = (u32x4){x, 1, 2, 3}; p[((rc + 11) % RC) * G + g] = (u32x4){x, 1, 2, 3}; }
        if (MODE == 2 || MODE == 4 || MODE == 6) myhc[key * 5 + (k % 5)] = x;
        if (MODE == 5) mykv[key * G + g] = x;
    }
    if (x == 77u) out[0] = x;
}

int main() {
    const uint32_t G = 65536, RC = 2400;
    u32x4 *p; uint32_t *out, *hc; uint64_t *kv;
    CK(hipMalloc(&p, (size_t)RC * G * 16)); CK(hipMemset(p, 0, (size_t)RC * G * 16)); CK(hipMalloc(&out, 4));
    CK(hipMalloc(&hc, (size_t)5 * G * 320 * 4)); CK(hipMemset(hc, 0, (size_t)5 * G * 320 * 4));
    CK(hipMalloc(&kv, (size_t)5 * 64 * G * 8)); CK(hipMemset(kv, 0, (size_t)5 * 64 * G * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 64;
#define RUN(M, name) for (int rep = 0; rep < 3; rep++) { \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((chase<M>), dim3(G / 128), dim3(640), 0, 0, p, hc, kv, G, RC, K, out); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        if (rep) printf("%-64s %8.1f us  %6.2f us per round\n", name, ms * 1e3, ms * 1e3 / K); }
    RUN(0, "0 coalesced 16 B load only");
    RUN(1, "1 + coalesced 16 B store");
    RUN(2, "2 + scattered 4 B store (hc)");
    RUN(3, "3 + five scattered 4 B loads (hc)");
    RUN(4, "4 + hc loads + hc store");
    RUN(5, "5 + scattered 8 B load + store (kv)");
    RUN(6, "6 + three coalesced stores + hc store");
    return 0;
}
