// bit-exactness check of agz_fastdiv.hpp against the compiler's '/' over the range its callers guarantee
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include "../alphagpu_amd/csrc/agz_fastdiv.hpp"
__global__ void k(const float* n, const float* d, float* a, float* b, float* c, int N) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < N) {
        float q0, q1; agz::fd_div_pair(n[2*i], d[2*i], n[2*i+1], d[2*i+1], q0, q1);
        a[2*i] = q0; a[2*i+1] = q1;
        b[2*i] = n[2*i] / d[2*i]; b[2*i+1] = n[2*i+1] / d[2*i+1];
        const float r = agz::fd_rcp(d[2*i]);                       // shared denominator form
        c[2*i] = agz::fd_div(n[2*i], d[2*i], r);
        float s0, s1; agz::fd_div2(n[2*i], d[2*i], r, n[2*i+1], d[2*i], r, s0, s1);
        c[2*i+1] = (s0 == c[2*i] || (s0 != s0)) ? s1 : __builtin_nanf("");     // second quotient by the FIRST denominator
    }
}
int main() {
    const int N = 1 << 26;
    std::vector<float> n(N), d(N);
    std::mt19937_64 g(7);
    auto rnd_in = [&](int lo, int hi) {                            // random sign, exponent in [lo, hi], random mantissa
        uint32_t m = (uint32_t)g() & 0x7fffffu, s = (uint32_t)(g() & 1) << 31;
        int e = lo + (int)(g() % (uint64_t)(hi - lo + 1));
        uint32_t u = s | ((uint32_t)(e + 127) << 23) | m; float x; memcpy(&x, &u, 4); return x;
    };
    for (int i = 0; i < N; ++i) {
        int mode = i & 7;
        if (mode == 0) { n[i] = rnd_in(-100, 100); d[i] = rnd_in(-100, 100); }
        else if (mode == 1) { n[i] = 0.0f; d[i] = rnd_in(-100, 100); }
        else if (mode == 2) { n[i] = fabsf(rnd_in(-86, 11)); d[i] = fabsf(rnd_in(-14, 4)); }            // policy rows
        else if (mode == 3) { n[i] = fabsf(rnd_in(-62, 0)); d[i] = fabsf(rnd_in(0, 7)); }               // softmax
        else if (mode == 4) { n[i] = fabsf(rnd_in(-70, 0)); d[i] = fabsf(rnd_in(-70, 0)); }             // normalize
        else if (mode == 5) { n[i] = rnd_in(-10, 25); d[i] = -fabsf(rnd_in(-10, 39)); }                 // Newton step
        else if (mode == 6) { n[i] = (float)((uint32_t)g() >> 8) / 16777216.0f * 2.0f; d[i] = 1e-4f + (float)((uint32_t)g() >> 8) / 16777216.0f * 3.0f; }
        else { n[i] = rnd_in(-3, 3); d[i] = rnd_in(-3, 3); }
        // keep the quotient inside the stated range
        float qq = fabsf(n[i] / d[i]);
        if (n[i] != 0.0f && !(qq >= ldexpf(1.0f, -120) && qq <= ldexpf(1.0f, 95))) { n[i] = 1.0f; }
    }
    for (int i = 0; i + 1 < N; i += 2) {                               // the shared-denominator form divides n[i+1] by d[i] as well
        float qq = fabsf(n[i + 1] / d[i]);
        if (n[i + 1] != 0.0f && !(qq >= ldexpf(1.0f, -120) && qq <= ldexpf(1.0f, 95))) n[i + 1] = d[i];
    }
    float *dn, *dd, *da, *db, *dc; hipMalloc(&dn, (size_t)N * 4); hipMalloc(&dd, (size_t)N * 4); hipMalloc(&da, (size_t)N * 4); hipMalloc(&db, (size_t)N * 4); hipMalloc(&dc, (size_t)N * 4);
    hipMemcpy(dn, n.data(), (size_t)N * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), (size_t)N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(N / 2 / 256), dim3(256), 0, 0, dn, dd, da, db, dc, N);
    std::vector<uint32_t> a(N), b(N), c(N);
    hipMemcpy(a.data(), da, (size_t)N * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, (size_t)N * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, (size_t)N * 4, hipMemcpyDeviceToHost);
    long bad = 0, bad2 = 0;
    for (int i = 0; i < N; ++i) {
        if (a[i] != b[i]) { if (bad < 5) printf("pair mismatch %d: %a / %a -> %08x vs %08x\n", i, n[i], d[i], a[i], b[i]); ++bad; }
        float ref = (i & 1) ? n[i] / d[i - 1] : n[i] / d[i]; uint32_t ru; memcpy(&ru, &ref, 4);
        if (c[i] != ru) { if (bad2 < 5) printf("shared mismatch %d: %a / %a -> %08x vs %08x\n", i, n[i], (i & 1) ? d[i - 1] : d[i], c[i], ru); ++bad2; }
    }
    printf("fd_div_pair vs '/': %ld mismatches of %d; shared-denominator fd_div / fd_div2 vs '/': %ld mismatches\n", bad, N, bad2);
    return (bad != 0 || bad2 != 0);
}
