// scratch/mfma_probe2.hip (usage: mfma_probe2 tests.bin out.bin) — dumps inputs and outputs of v_mfma_f32_16x16x32_bf16 so that the accumulation model of the matrix
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
    int n = 0; FILE* fi = fopen(argv[1], "rb"); if (!fi || fread(&n, 4, 1, fi) != 1) { printf("no input\n"); return 2; }
    std::vector<uint16_t> A((size_t)n * 512), B((size_t)n * 512);
    std::vector<float> C((size_t)n * 256), D((size_t)n * 256);
    if (fread(A.data(), 2, A.size(), fi) != A.size() || fread(B.data(), 2, B.size(), fi) != B.size() || fread(C.data(), 4, C.size(), fi) != C.size()) { printf("short input\n"); return 2; }
    fclose(fi);
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, n);
    if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    FILE* f = fopen(argv[2], "wb");
    fwrite(&n, 4, 1, f); fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, C.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("mfma probe: %d tiles written\n", n);
    return 0;
}
