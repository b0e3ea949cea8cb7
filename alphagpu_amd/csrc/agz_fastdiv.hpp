// agz_fastdiv.hpp — correctly rounded fp32 quotients for operands in the search's range, at half the instructions.
//
// The compiler lowers n / d to v_div_scale x2, v_rcp, five FMA-class steps, v_div_fmas, v_div_fixup (agz_divpair.hpp).  The
// scale / fixup steps only act on operands near the ends of the exponent range (denormal operands or quotients, exponent
// differences >= 96, |n| < 2^-102): for
//        n == 0  or  2^-100 <= |n| <= 2^100,      2^-100 <= |d| <= 2^100,      2^-120 <= |n/d| <= 2^95
// v_div_scale returns its operand unchanged, v_div_fmas is a plain fma and v_div_fixup returns its first operand, so the
// quotient is EXACTLY what the middle steps compute — and those are kept, in the compiler's order.  The callers guarantee
// the range (agz_tree_eager.hpp states why for every call site); scratch/fastdiv_test.hip checks 2^26 operand pairs of that
// range against '/', bit for bit.  A denominator shared by many quotients (softmax, normalize) is refined once.
#pragma once
#include <hip/hip_runtime.h>
#include "agz_divpair.hpp"

namespace agz {

// refined reciprocal of d: rcp, fma0 = fma(-d, rcp, 1), fma1 = fma(fma0, rcp, rcp)
__device__ __forceinline__ float fd_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// n / d given r = fd_rcp(d): mul = n r; fma2 = fma(-d, mul, n); fma3 = fma(fma2, r, mul); fma4 = fma(-d, fma3, n); fma(fma4, r, fma3)
__device__ __forceinline__ float fd_div(float n, float d, float r) {
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}
// two quotients with one (refined) reciprocal each, the FMA steps packed
__device__ __forceinline__ void fd_div2(float n0, float d0, float r0, float n1, float d1, float r1, float& q0, float& q1) {
    const f32x2 n = {n0, n1}, d = {d0, d1}, r = {r0, r1};
    f32x2 q = n * r;
    f32x2 e = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(e, r, q);
    e = __builtin_elementwise_fma(-d, q, n);
    q = __builtin_elementwise_fma(e, r, q);
    q0 = q.x; q1 = q.y;
}
// two quotients, two denominators (reciprocals refined here, packed)
__device__ __forceinline__ void fd_div_pair(float n0, float d0, float n1, float d1, float& q0, float& q1) {
    const f32x2 d = {d0, d1}, one = {1.0f, 1.0f};
    f32x2 r = {__builtin_amdgcn_rcpf(d0), __builtin_amdgcn_rcpf(d1)};
    const f32x2 e0 = __builtin_elementwise_fma(-d, r, one);
    r = __builtin_elementwise_fma(e0, r, r);
    fd_div2(n0, d0, r.x, n1, d1, r.y, q0, q1);
}

}  // namespace agz
