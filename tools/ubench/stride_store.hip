// Microbenchmark: 1024 waves, each storing 32 "slots" x 3 arrays.
//   layout A: slot stride = G elements (ring rows [W][G])      -> 256-512 KB apart
//   layout B: wave-tiled [G/64][W][64]                         -> 256-512 B apart
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_rows(uint64_t *a, uint32_t *b, uint32_t *c, uint32_t G, uint32_t W, uint32_t base, int S) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < S; k++) {
        size_t i = (size_t)((base + k) & (W - 1)) * G + g;
        a[i] = g + k; b[i] = k; c[i] = g;
    }
}
__global__ __launch_bounds__(256) void k_tiled(uint64_t *a, uint32_t *b, uint32_t *c, uint32_t G, uint32_t W, uint32_t base, int S) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    for (int k = 0; k < S; k++) {
        size_t i = (((size_t)(g >> 6) * W + ((base + k) & (W - 1))) << 6) | (g & 63);
        a[i] = g + k; b[i] = k; c[i] = g;
    }
}
int main() {
    const uint32_t G = 65536, W = 512; const int S = 32;
    uint64_t *a; uint32_t *b, *c;
    CK(hipMalloc(&a, (size_t)W * G * 8)); CK(hipMalloc(&b, (size_t)W * G * 4)); CK(hipMalloc(&c, (size_t)W * G * 4));
    CK(hipMemset(a, 0, (size_t)W * G * 8)); CK(hipMemset(b, 0, (size_t)W * G * 4)); CK(hipMemset(c, 0, (size_t)W * G * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            for (int t = 0; t < 16; t++) {
                if (mode == 0) hipLaunchKernelGGL(k_rows, dim3(G / 256), dim3(256), 0, 0, a, b, c, G, W, t * S, S);
                else hipLaunchKernelGGL(k_tiled, dim3(G / 256), dim3(256), 0, 0, a, b, c, G, W, t * S, S);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = 16.0 * G * S * 16;
            printf("%s: %.1f us/launch, %.0f GB/s\n", mode ? "tiled [G/64][W][64]" : "rows  [W][G]      ", ms * 1e3 / 16, bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
