// scratch/mfma_probe.hip — dumps inputs and outputs of v_mfma_f32_16x16x32_bf16 so that the accumulation model of the matrix
// core (order / width of the 32-term sum) can be fitted offline (scratch/mfma_fit.py).  Diagnostic, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n) {
    const int t = blockIdx.x, l = threadIdx.x;
    if (t >= n) return;
    // A[t][16][32] row-major (row m, k), B[t][16][32] (col n, k), C/D[t][16][16] (m, n)
    bf16x8 a, b;
    const uint16_t* ap = A + (size_t)t * 512 + (l & 15) * 32 + 8 * (l >> 4);
    const uint16_t* bp = B + (size_t)t * 512 + (l & 15) * 32 + 8 * (l >> 4);
    uint16_t ta[8], tb[8];
    for (int j = 0; j < 8; ++j) { ta[j] = ap[j]; tb[j] = bp[j]; }
    __builtin_memcpy(&a, ta, 16); __builtin_memcpy(&b, tb, 16);
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(size_t)t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(size_t)t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
static uint16_t f2bf(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
int main(int argc, char** argv) {
    const int n = 4096;
    std::vector<uint16_t> A((size_t)n * 512), B((size_t)n * 512);
    std::vector<float> C((size_t)n * 256), D((size_t)n * 256);
    srand(12345);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (int t = 0; t < n; ++t) {
        const int mode = t % 8;     // exponent spreads: 0 uniform, 1..: wider dynamic range, sparse, one huge term, ...
        for (int i = 0; i < 512; ++i) {
            float x = rnd(), y = rnd();
            if (mode == 1) { x *= ldexpf(1.0f, rand() % 16 - 8); y *= ldexpf(1.0f, rand() % 16 - 8); }
            if (mode == 2) { x *= ldexpf(1.0f, rand() % 40 - 20); y *= ldexpf(1.0f, rand() % 40 - 20); }
            if (mode == 3) { if (rand() % 4) x = 0; }
            if (mode == 4) { x = (rand() & 1) ? 1.0f : 0.0f; }                      // plane-like inputs
            if (mode == 5) { x = fabsf(x); y = fabsf(y); }                           // no cancellation
            if (mode == 6) { x *= ldexpf(1.0f, (i % 32) - 16); }                     // exponent ramp along k
            if (mode == 7) { x = ldexpf(1.0f, -(i % 32)); y = 1.0f + ldexpf(1.0f, -7); }   // sticky-bit probe
            A[(size_t)t * 512 + i] = f2bf(x); B[(size_t)t * 512 + i] = f2bf(y);
        }
        for (int i = 0; i < 256; ++i) { float c = rnd(); if (mode == 2) c *= ldexpf(1.0f, rand() % 40 - 20); if (mode == 3) c = 0; C[(size_t)t * 256 + i] = c; }
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    FILE* f = fopen(argc > 1 ? argv[1] : "gpurun_out/mfma_probe.bin", "wb");
    fwrite(&n, 4, 1, f); fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, C.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("mfma probe: %d tiles written\n", n);
    return 0;
}
