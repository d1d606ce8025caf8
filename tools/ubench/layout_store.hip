// Microbenchmark of the R1 append pattern (per wave: S iterations x {1 load, 2 u64 + 4 u32 stores})
// under three layouts of the row-dimensioned arrays:
//   rows    X[row][g]
//   tiled   X[g/64][row][g%64]
//   blockB  X[row/B][g/64][row%B][g%64]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int B>
__device__ __forceinline__ size_t idx(uint32_t rows, uint32_t row, uint32_t g, uint32_t G) {
    if (MODE == 0) return (size_t)row * G + g;
    if (MODE == 1) return (((size_t)(g >> 6) * rows + row) << 6) | (g & 63);
    return ((((size_t)(row / B) * (G >> 6) + (g >> 6)) * B + (row % B)) << 6) | (g & 63);
}
template <int MODE, int B>
__global__ __launch_bounds__(256) void k(uint64_t *a0, uint64_t *a1, uint32_t *b0, uint32_t *b1, uint32_t *b2, uint32_t *b3,
                                         const uint32_t *tok, uint32_t G, uint32_t W, uint32_t cap, uint32_t base, int S) {
    uint32_t g = blockIdx.x * 256 + threadIdx.x;
    for (int k0 = 0; k0 < S; k0 += 8) {
        uint32_t t[8];
#pragma unroll
        for (int q = 0; q < 8; q++) t[q] = tok[(size_t)(k0 + q) * G + g];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            uint32_t slot = base + k0 + q;
            size_t i = idx<MODE, B>(W, slot & (W - 1), g, G), o = idx<MODE, B>(cap, k0 + q, g, G);
            a0[i] = 0x101; b0[i] = t[q]; b1[i] = 0x12345 | t[q];
            b2[o] = slot; a1[o] = 0x101; b3[o] = t[q];
        }
    }
}
int main() {
    const uint32_t G = 65536, W = 512, cap = 512; const int S = 32;
    uint64_t *a0, *a1; uint32_t *b0, *b1, *b2, *b3, *tok;
    size_t n = (size_t)W * G;
    CK(hipMalloc(&a0, n * 8)); CK(hipMalloc(&a1, n * 8)); CK(hipMalloc(&b0, n * 4)); CK(hipMalloc(&b1, n * 4));
    CK(hipMalloc(&b2, n * 4)); CK(hipMalloc(&b3, n * 4)); CK(hipMalloc(&tok, (size_t)S * G * 4));
    CK(hipMemset(tok, 1, (size_t)S * G * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"rows [row][g]", "tiled [g/64][row][64]", "block16", "block4", "block64"};
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            for (int t = 0; t < 16; t++) {
                dim3 gr(G / 256), bl(256);
                uint32_t base = t * S;
                if (mode == 0) hipLaunchKernelGGL((k<0, 1>), gr, bl, 0, 0, a0, a1, b0, b1, b2, b3, tok, G, W, cap, base, S);
                if (mode == 1) hipLaunchKernelGGL((k<1, 1>), gr, bl, 0, 0, a0, a1, b0, b1, b2, b3, tok, G, W, cap, base, S);
                if (mode == 2) hipLaunchKernelGGL((k<2, 16>), gr, bl, 0, 0, a0, a1, b0, b1, b2, b3, tok, G, W, cap, base, S);
                if (mode == 3) hipLaunchKernelGGL((k<2, 4>), gr, bl, 0, 0, a0, a1, b0, b1, b2, b3, tok, G, W, cap, base, S);
                if (mode == 4) hipLaunchKernelGGL((k<2, 64>), gr, bl, 0, 0, a0, a1, b0, b1, b2, b3, tok, G, W, cap, base, S);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double bytes = 16.0 * G * S * 36;
            printf("%-24s %.1f us/launch, %.0f GB/s\n", names[mode], ms * 1e3 / 16, bytes / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
