// agz_wave.hpp — lane-group primitives of the tree kernel: G consecutive lanes of a wavefront work on one game tree and talk
// through DPP (a few cycles) instead of LDS.
#pragma once
#include "agz_device.hpp"

namespace agz {

#ifdef AGZ_STAMPS
// diagnostic builds only: per-phase cycle sums kept in LDS (stamp_lds[0..15] sums, [16] last time stamp) and updated by the
// first ACTIVE lane, so that the attribution is right inside divergent code as well
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime();                                   \
                      if (lane_id() == (int)__builtin_ctzll(__ballot(1))) { stamp_lds[i] += t_ - stamp_lds[16]; stamp_lds[16] = t_; } } while (0)
#define STAMPW(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(i); } while (0)   // waits are charged to the phase that issued the loads
#else
#define STAMP(i) do { } while (0)
#define STAMPW(i) do { } while (0)
#endif

#define AGZ_WSYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// what changes from rollout to rollout (kept apart from TreePar so that a caller looping over rollouts — k_search_small — can
// leave the big parameter block in constant kernel-argument memory).  fin: the call only closes the search.
struct StepFlags { uint32_t rollout; int do_reset, do_expand, do_select, last, fin = 0; };

template <int NC, bool REV> __device__ __forceinline__ WPos<NC> grp_load_pos(const Pos* p) {
    WPos<NC> w;
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
#pragma unroll
    for (int i = 0; i < NC; ++i) { w.p.c[i] = q[i]; w.o.c[i] = q[3 + i]; w.lg.c[i] = REV ? q[6 + i] : 0ull; }
    const uint32_t tail = *reinterpret_cast<const uint32_t*>(q + 9);
    w.player = (int)(int8_t)(tail & 0xff);
    w.aux = (int)(int8_t)((tail >> 8) & 0xff);
    return w;
}

template <int CTRL, int BANK> __device__ __forceinline__ int dpp_mov(int old, int x) {
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false);
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_QUAD_B0 = 0x00, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118,
       DPP_SHR1 = 0x111, DPP_SHL4 = 0x104, DPP_SHL8 = 0x108, DPP_QUAD_B3 = 0xFF, DPP_QUAD_B13 = 0xF5 };

template <int G> __device__ __forceinline__ int grp_bcast(int x) {          // value of the group's lane 0
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group size");
    if (G == 2) return dpp_mov<0xA0, 0xF>(x, x);                            // quad_perm [0,0,2,2]
    if (G == 16) x = dpp_mov<DPP_SHR8, 0xC>(x, x);                          // lanes 8..15 <- lanes 0..7
    if (G >= 8) x = dpp_mov<DPP_SHR4, 0xA>(x, x);                           // lanes 4..7 (12..15) <- lanes 0..3 (8..11)
    if (G >= 4) x = dpp_mov<DPP_QUAD_B0, 0xF>(x, x);
    return x;
}
template <int G> __device__ __forceinline__ float grp_bcast(float x) { return __int_as_float(grp_bcast<G>(__float_as_int(x))); }
template <int G> __device__ __forceinline__ float grp_bcast_last(float xf) {   // value held by the LAST lane of the group
    int x = __float_as_int(xf);
    if (G == 2) x = dpp_mov<DPP_QUAD_B13, 0xF>(x, x);
    if (G == 16) x = dpp_mov<DPP_SHL8, 0x3>(x, x);
    if (G >= 8) x = dpp_mov<DPP_SHL4, 0x5>(x, x);
    if (G >= 4) x = dpp_mov<DPP_QUAD_B3, 0xF>(x, x);
    return __int_as_float(x);
}
// Group reductions with the lane exchange FUSED into the arithmetic instruction (v_add_u32_dpp / v_max_*_dpp: one instruction per
// step; a v_mov_b32_dpp + compare + select sequence is six).  A DPP operand written by the preceding VALU instruction needs two
// wait states: the s_nop 1 in front of every step.  (v_max_f32 and `y > x ? y : x` agree on every non-NaN pair.)
#define AGZ_DPP_STEP(op, ctrl, x) asm("s_nop 1\n\t" op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf" : "+v"(x))
template <int G> __device__ __forceinline__ int grp_sum(int x) {
    static_assert(G == 2 || G == 4 || G == 8, "lane-groups of 2, 4 or 8");
    AGZ_DPP_STEP("v_add_u32_dpp", "quad_perm:[1,0,3,2]", x);
    if constexpr (G >= 4) AGZ_DPP_STEP("v_add_u32_dpp", "quad_perm:[2,3,0,1]", x);
    if constexpr (G >= 8) AGZ_DPP_STEP("v_add_u32_dpp", "row_half_mirror", x);
    return x;
}
template <int G> __device__ __forceinline__ int grp_max_i(int x) {
    static_assert(G == 2 || G == 4 || G == 8, "lane-groups of 2, 4 or 8");
    AGZ_DPP_STEP("v_max_i32_dpp", "quad_perm:[1,0,3,2]", x);
    if constexpr (G >= 4) AGZ_DPP_STEP("v_max_i32_dpp", "quad_perm:[2,3,0,1]", x);
    if constexpr (G >= 8) AGZ_DPP_STEP("v_max_i32_dpp", "row_half_mirror", x);
    return x;
}
template <int G> __device__ __forceinline__ float grp_max(float x) {
    static_assert(G == 2 || G == 4 || G == 8, "lane-groups of 2, 4 or 8");
    AGZ_DPP_STEP("v_max_f32_dpp", "quad_perm:[1,0,3,2]", x);
    if constexpr (G >= 4) AGZ_DPP_STEP("v_max_f32_dpp", "quad_perm:[2,3,0,1]", x);
    if constexpr (G >= 8) AGZ_DPP_STEP("v_max_f32_dpp", "row_half_mirror", x);
    return x;
}
// a += t[lane + d], b += u[lane + d] for d = 1 .. G - 1, in that order (the source-order sum of G consecutive lanes' values ends up
// in the first of them; lanes whose source lies past the 16-lane row add 0 — only a group's first lane is read afterwards)
template <int G = 8> __device__ __forceinline__ void grp_pull_sums(float& a, const float t, float& b, const float u) {
#define AGZ_PULL2(d) "v_add_f32_dpp %0, %2, %0 row_shl:" #d " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
                     "v_add_f32_dpp %1, %3, %1 row_shl:" #d " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    // (early clobber: a starts as a copy of t and must not share its register)
    if constexpr (G == 8)
        asm("s_nop 1\n\t" AGZ_PULL2(1) AGZ_PULL2(2) AGZ_PULL2(3) AGZ_PULL2(4) AGZ_PULL2(5) AGZ_PULL2(6) AGZ_PULL2(7) : "+&v"(a), "+&v"(b) : "v"(t), "v"(u));
    else if constexpr (G == 4)
        asm("s_nop 1\n\t" AGZ_PULL2(1) AGZ_PULL2(2) AGZ_PULL2(3) : "+&v"(a), "+&v"(b) : "v"(t), "v"(u));
    else
        asm("s_nop 1\n\t" AGZ_PULL2(1) : "+&v"(a), "+&v"(b) : "v"(t), "v"(u));
#undef AGZ_PULL2
}
template <int D> __device__ __forceinline__ float lane_shl(float x) {      // value of lane + D (same 16-lane row), own value past the row's end
    return __int_as_float(dpp_mov<0x100 + D, 0xF>(__float_as_int(x), __float_as_int(x)));
}
// exclusive prefix sum over the G lanes of a group: lane sub gets the sum of the lanes 0 .. sub - 1 (log2 G row_shr steps; a lane
// whose source lies in the neighbouring group of the DPP row, or outside the row, adds nothing)
template <int G = 8> __device__ __forceinline__ int grp_excl_prefix8(const int v, const int sub) {
    int a = v;
    int t = dpp_mov<0x111, 0xF>(0, a); a += sub >= 1 ? t : 0;
    if constexpr (G >= 4) { t = dpp_mov<0x112, 0xF>(0, a); a += sub >= 2 ? t : 0; }
    if constexpr (G >= 8) { t = dpp_mov<0x114, 0xF>(0, a); a += sub >= 4 ? t : 0; }
    return a - v;
}
__device__ __forceinline__ float lane_shr1(float x) { return __int_as_float(dpp_mov<DPP_SHR1, 0xF>(__float_as_int(x), __float_as_int(x))); }

// Source-order running sums over the group's G*KPL values (lane sub holds block sub): returns the sum of everything BEFORE the
// lane's own block — the lanes take turns, lane t adds its KPL values to what lane t-1 ended with (one DPP row_shr:1 per turn),
// bit-identical to the source-order loop.  nl = lanes whose block holds real actions (ceil(A / KPL), wave-uniform): the blocks
// of the lanes behind them are all +0 padding, which a sum passes through unchanged, so their turns are not taken — the total is
// handed down the remaining lanes by one move per lane.  The start of lane nl - 1 needs no turn of its own; its end (the total)
// does.  (Connect4: 2 turns instead of 8, Gobang 9x9: 7 / 6 instead of 8 / 7.)
template <int KPL, bool WANT_TOTAL, int G = 8>
__device__ __forceinline__ float grp_ordered_start(const float (&x)[KPL], int sub, float& total, int nl = G) {
    float a = 0.0f, st = 0.0f;
    const int turns = WANT_TOTAL ? nl : nl - 1;
#pragma unroll 1
    for (int t = 0; t < turns; ++t) {
        const float carry = lane_shr1(a);                       // what the previous lane ended with
        const float s0 = sub == 0 ? 0.0f : carry;
        if (sub == t) st = s0;
        a = s0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) a += x[j];                // only lane t's result is final in turn t
    }
    if (WANT_TOTAL) {
#pragma unroll 1
        for (int t = nl; t < G; ++t) { const float carry = lane_shr1(a); a = sub == t ? carry : a; }   // padding lanes pass the total on
        total = grp_bcast_last<G>(a);
    } else { const float carry = lane_shr1(a); if (sub == nl - 1) st = carry; }
    return st;
}

}  // namespace agz
