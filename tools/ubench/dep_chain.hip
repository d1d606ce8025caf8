// Microbenchmark (round 4): what one dependent round trip costs a wavefront of the EPaxos cluster tick.
// 512 blocks x 10 wavefronts (one block per CU and a half, as ep_cluster_tick_kernel runs), every lane a chain of K dependent
// 16-byte loads; the cell of step k is (rc_k, g):
//   rows : [RC][G] planes -- a wavefront's 1 KB of cell rc lies G*16 B = 1 MB from its 1 KB of cell rc+1 (the engine's layout)
//   tiled: [G/64][RC][64]  -- a wavefront's cells are contiguous (RC KB)
// and rc_k either the same for every wavefront at step k (lock-step: what the tick mostly does) or per-wavefront random.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int PERWAVE, int WIDE>
__global__ __launch_bounds__(640) void chase(const u32x4 *__restrict__ p, uint32_t G, uint32_t RC, int K, uint32_t *out) {
    const uint32_t g = blockIdx.x * 128 + (threadIdx.x & 127u);       // two sets of 64 groups, five wavefronts on each
    const uint32_t rep = threadIdx.x >> 7;                             // replica 0..4: its own fifth of the cells
    uint32_t x = 0, h = 12345u + (PERWAVE ? (blockIdx.x * 10 + (threadIdx.x >> 6)) * 2654435761u : 0u);
    for (int k = 0; k < K; k++) {
        h = h * 1664525u + 1013904223u;
        uint32_t acc = 0;
#pragma unroll
        for (int w = 0; w < WIDE; w++) {                               // WIDE independent loads per round
            const uint32_t rc = (rep * (RC / 5) + ((h >> 8) + w * 7919u + x) % (RC / 5));
            const uint32_t i = LAYOUT == 0 ? rc * G + g : ((g >> 6) * RC + rc) * 64 + (g & 63u);
            acc += p[i].x;
        }
        x = acc;                                                       // (zero: the buffer is; the dependence is what matters)
    }
    if (x == 77u) out[0] = x;
}

int main() {
    const uint32_t G = 65536, RC = 2400;                               // 5 replicas x 3 planes x 5 rows x 32 columns
    u32x4 *p; uint32_t *out;
    CK(hipMalloc(&p, (size_t)RC * G * 16)); CK(hipMemset(p, 0, (size_t)RC * G * 16)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 64;
#define RUN(L, PW, WD, name) for (int rep = 0; rep < 2; rep++) { \
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((chase<L, PW, WD>), dim3(G / 128), dim3(640), 0, 0, p, G, RC, K, out); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("%-44s %8.1f us  %6.2f us per round (2 blocks per CU in turn: /2 = %5.2f)  %7.0f GB/s\n", name, ms * 1e3, ms * 1e3 / K, ms * 1e3 / K / 2, \
               (double)G * 5 * K * WD * 16 / (ms * 1e-3) / 1e9); }
    RUN(0, 0, 1, "rows  lock-step, 1 load per round");
    RUN(1, 0, 1, "tiled lock-step, 1 load per round");
    RUN(0, 1, 1, "rows  per-wavefront cells, 1 load per round");
    RUN(1, 1, 1, "tiled per-wavefront cells, 1 load per round");
    RUN(0, 0, 6, "rows  lock-step, 6 loads per round");
    RUN(1, 0, 6, "tiled lock-step, 6 loads per round");
    RUN(0, 1, 6, "rows  per-wavefront cells, 6 loads per round");
    RUN(1, 1, 6, "tiled per-wavefront cells, 6 loads per round");
    return 0;
}
