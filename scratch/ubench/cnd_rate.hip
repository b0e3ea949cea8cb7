// what does v_cndmask_b32 cost on gfx950, by form and neighbourhood?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define OPS8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
template <int OP> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.5f, a2 = 2.5f, a3 = 3.5f, a4 = 4.5f, a5 = 5.5f, a6 = 6.5f, a7 = 7.5f;
    const float m = 1.0001f, c = 0.5f;
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc", "s20", "s21", "s22", "s23"
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc" ARGS);) }
        if (OP == 1) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" ARGS);) }
        if (OP == 2) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" ARGS);) }
        // a different destination than the sources (no read-modify-write of the same register)
        if (OP == 3) { REP8(asm volatile("v_cndmask_b32_e32 %0, %1, %8, vcc\n v_cndmask_b32_e32 %1, %2, %8, vcc\n v_cndmask_b32_e32 %2, %3, %8, vcc\n v_cndmask_b32_e32 %3, %4, %8, vcc\n v_cndmask_b32_e32 %4, %5, %8, vcc\n v_cndmask_b32_e32 %5, %6, %8, vcc\n v_cndmask_b32_e32 %6, %7, %8, vcc\n v_cndmask_b32_e32 %7, %0, %8, vcc" ARGS);) }
        // one select among three fmas
        if (OP == 4) { REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_cndmask_b32_e32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" ARGS);) }
        // the compiler's usual pair: compare into vcc, select on vcc
        if (OP == 5) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32_e32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32_e32 %7, %7, %8, vcc" ARGS);) }
        if (OP == 6) { REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_f32 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_lt_f32 s[22:23], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]" ARGS);) }
        // values that are not denormal bit patterns / plain integers
        if (OP == 7) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" ARGS);) }
        if (OP == 8) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" ARGS);) }
        if (OP == 9) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" ARGS);) }
        if (OP == 10) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_lshlrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_and_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" ARGS);) }
        if (OP == 11) { REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc" ARGS);) }
        if (OP == 12) { REP8(asm volatile("v_mov_b32_dpp %0, %0 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %1, %1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n v_mov_b32_dpp %3, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" ARGS);) }
        if (OP == 13) { REP8(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_f32_ubyte0 %6, %6\n v_cvt_f32_i32 %7, %7" ARGS);) }
        if (OP == 14) { REP8(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s22, %6\n v_readfirstlane_b32 s23, %7" ARGS);) }
        if (OP == 15) { REP8(asm volatile("v_exp_f32 %0, %0\n v_log_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_rsq_f32 %3, %3\n v_exp_f32 %4, %4\n v_log_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_rsq_f32 %7, %7" ARGS);) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int OP> void run(const char* name, float* out) {
    const int iters = 4000, w = 4, nb = 256 * w;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<nb, 256>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<OP><<<nb, 256>>>(out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * w;
    printf("%-44s %.3f ms -> %.2f ns / instruction / SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<9>("v_add/mul/max/sub_f32 (warm-up)", out);
    run<0>("v_cndmask_b32_e32 vcc (dst = src0)", out);
    run<1>("v_cndmask_b32_e64 s[20:21]", out);
    run<2>("v_cndmask_b32_e64 vcc", out);
    run<3>("v_cndmask_b32_e32 vcc (dst != src)", out);
    run<4>("1 cndmask vcc + 3 fma", out);
    run<5>("v_cmp vcc ; cndmask vcc", out);
    run<6>("v_cmp sgpr ; cndmask sgpr", out);
    run<7>("v_mov_b32", out);
    run<8>("v_add_u32", out);
    run<9>("v_add/mul/max/sub_f32", out);
    run<10>("v_and/or/xor/shift", out);
    run<11>("v_add_co / v_addc_co vcc", out);
    run<12>("dpp mov/max/add", out);
    run<13>("v_cvt mix", out);
    run<14>("v_readfirstlane", out);
    run<15>("v_exp/log/sqrt/rsq", out);
    return 0;
}
