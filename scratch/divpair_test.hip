// bit-exactness check of div_pair (two IEEE fp32 divisions sharing packed FMA steps) against the compiler's '/'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
#include "../alphagpu_amd/csrc/agz_divpair.hpp"
__global__ void k(const float* n, const float* d, float* a, float* b, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < N) {
        float q0, q1; agz::div_pair(n[2*i], d[2*i], n[2*i+1], d[2*i+1], q0, q1);
        a[2*i] = q0; a[2*i+1] = q1;
        b[2*i] = n[2*i] / d[2*i]; b[2*i+1] = n[2*i+1] / d[2*i+1];
    }
}
int main() {
    const int N = 1 << 24;
    std::vector<float> n(N), d(N);
    std::mt19937_64 g(1);
    const float sp[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-38f, 1e-45f, 3e38f, INFINITY, -INFINITY, NAN, 1e-4f, 0.75f, 1.17549435e-38f, 5e-39f, 2.0f, 3.0f};
    for (int i = 0; i < N; ++i) {
        uint32_t a = (uint32_t)g(), b = (uint32_t)g();
        int mode = i & 7;
        if (mode == 0) { memcpy(&n[i], &a, 4); memcpy(&d[i], &b, 4); }                    // any bit patterns
        else if (mode == 1) { n[i] = sp[a % 16]; d[i] = sp[b % 16]; }
        else if (mode == 2) { n[i] = sp[a % 16]; memcpy(&d[i], &b, 4); }
        else { n[i] = (float)(a >> 8) / 16777216.0f * (mode == 3 ? 1e-30f : 2.0f); d[i] = 1e-4f + (float)(b >> 8) / 16777216.0f * 3.0f; }   // the search's range
    }
    float *dn, *dd, *da, *db; hipMalloc(&dn, N * 4); hipMalloc(&dd, N * 4); hipMalloc(&da, N * 4); hipMalloc(&db, N * 4);
    hipMemcpy(dn, n.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 2 / 256), dim3(256), 0, 0, dn, dd, da, db, N);
    std::vector<uint32_t> a(N), b(N);
    hipMemcpy(a.data(), da, N * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, N * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < N; ++i) {
        bool nan_a = (a[i] & 0x7fffffffu) > 0x7f800000u, nan_b = (b[i] & 0x7fffffffu) > 0x7f800000u;
        if (a[i] != b[i] && !(nan_a && nan_b)) { if (bad < 5) printf("mismatch %d: %a / %a -> %08x vs %08x\n", i, n[i], d[i], a[i], b[i]); ++bad; }
    }
    printf("div_pair vs '/': %ld mismatches of %d\n", bad, N);
    return bad != 0;
}
