// issue cost of a few vector instructions on gfx950: N waves per SIMD, independent streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define REP8(X) X X X X X X X X
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x), "v"(c.x));) }
        if (OP == 1) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m), "v"(c));) }
        if (OP == 2) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 3) { REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %2, %2, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %4, %4, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %6, %6, %6 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 4) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m));) }
        if (OP == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x) : "vcc");) }
        if (OP == 6) { REP8(asm volatile("v_div_scale_f32 %0, vcc, %0, %8, %0\n v_div_scale_f32 %1, vcc, %1, %8, %1\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x), "v"(c.x) : "vcc");) }
        if (OP == 7) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x));) }
        if (OP == 8) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n v_and_or_b32 %1, %1, %8, %9\n v_bfe_u32 %2, %2, 3, 8\n v_add3_u32 %3, %3, %8, %9\n v_lshl_or_b32 %4, %4, 3, %8\n v_max3_f32 %5, %5, %8, %9\n v_alignbit_b32 %6, %6, %8, 8\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x), "v"(c.x));) }
        if (OP == 9) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x) : "vcc");) }
        if (OP == 10) { REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32 %3, %3, %8, s[22:23]\n v_cmp_lt_f32 s[24:25], %4, %8\n v_cndmask_b32 %5, %5, %8, s[24:25]\n v_cmp_lt_f32 s[26:27], %6, %8\n v_cndmask_b32 %7, %7, %8, s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m.x) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
    }
    if (OP == 1 || OP == 4) { a0 = p0.x + p0.y + p1.x + p1.y; a1 = p2.x + p2.y + p3.x + p3.y; a2 = p4.x + p4.y + p5.x + p5.y; a3 = p6.x + p6.y + p7.x + p7.y; }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = 2000, nb = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<nb, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<nb, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * wgs_per_cu;      // one wave of each workgroup per SIMD
    printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f ns per instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wgs_per_cu, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, out); run<1>("v_pk_fma_f32", 1, out); run<2>("v_rcp_f32", 1, out); run<3>("v_add_f32_dpp row_shl", 1, out); run<4>("v_pk_mul_f32", 1, out); run<5>("v_cndmask_b32 vcc", 1, out); run<6>("div_scale/fmas/fixup mix", 1, out); run<7>("v_mul_lo/hi_u32", 1, out); run<8>("3-operand int mix", 1, out); run<9>("v_cmp_lt_f32 vcc", 1, out); run<10>("v_cmp sgpr + cndmask sgpr", 1, out); }
        if (w == 2) { run<0>("v_fma_f32", 2, out); run<1>("v_pk_fma_f32", 2, out); run<2>("v_rcp_f32", 2, out); run<3>("v_add_f32_dpp row_shl", 2, out); run<4>("v_pk_mul_f32", 2, out); run<5>("v_cndmask_b32 vcc", 2, out); run<6>("div_scale/fmas/fixup mix", 2, out); run<7>("v_mul_lo/hi_u32", 2, out); run<8>("3-operand int mix", 2, out); run<9>("v_cmp_lt_f32 vcc", 2, out); run<10>("v_cmp sgpr + cndmask sgpr", 2, out); }
        if (w == 4) { run<0>("v_fma_f32", 4, out); run<1>("v_pk_fma_f32", 4, out); run<2>("v_rcp_f32", 4, out); run<3>("v_add_f32_dpp row_shl", 4, out); run<4>("v_pk_mul_f32", 4, out); run<5>("v_cndmask_b32 vcc", 4, out); run<6>("div_scale/fmas/fixup mix", 4, out); run<7>("v_mul_lo/hi_u32", 4, out); run<8>("3-operand int mix", 4, out); run<9>("v_cmp_lt_f32 vcc", 4, out); run<10>("v_cmp sgpr + cndmask sgpr", 4, out); }
    }
    return 0;
}
