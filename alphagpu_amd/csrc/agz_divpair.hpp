// agz_divpair.hpp — two IEEE-754 fp32 divisions at once.
//
// The tree arithmetic is strict IEEE: every n / d is the correctly rounded quotient, which the compiler lowers to
// v_div_scale x2, v_rcp, five FMA-class steps, v_div_fmas, v_div_fixup (11 VALU instructions; AMDGPU LowerFDIV32).  The
// five middle steps of TWO independent divisions are the same arithmetic on two values: gfx90a+ executes them as packed
// v_pk_fma_f32 / v_pk_mul_f32 (IEEE per component), 16 instructions for two quotients instead of 22.  The sequence and the
// operand order are exactly the compiler's, so the bits are the same as n0 / d0 and n1 / d1 (scratch/divpair_test.hip
// checks 16 M operand pairs including zeros, infinities, NaNs and denormals).
#pragma once
#include <hip/hip_runtime.h>

namespace agz {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void div_pair(float n0, float d0, float n1, float d1, float& q0, float& q1) {
    bool s0, s1, t0, t1;
    const float ds0 = __builtin_amdgcn_div_scalef(n0, d0, false, &t0), ns0 = __builtin_amdgcn_div_scalef(n0, d0, true, &s0);
    const float ds1 = __builtin_amdgcn_div_scalef(n1, d1, false, &t1), ns1 = __builtin_amdgcn_div_scalef(n1, d1, true, &s1);
    const f32x2 ds = {ds0, ds1}, ns = {ns0, ns1}, one = {1.0f, 1.0f};
    f32x2 r = {__builtin_amdgcn_rcpf(ds0), __builtin_amdgcn_rcpf(ds1)};
    const f32x2 e = __builtin_elementwise_fma(-ds, r, one);
    r = __builtin_elementwise_fma(e, r, r);
    f32x2 q = ns * r;
    const f32x2 e2 = __builtin_elementwise_fma(-ds, q, ns);
    q = __builtin_elementwise_fma(e2, r, q);
    const f32x2 e3 = __builtin_elementwise_fma(-ds, q, ns);
    q0 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e3.x, r.x, q.x, s0), d0, n0);
    q1 = __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e3.y, r.y, q.y, s1), d1, n1);
}

}  // namespace agz
