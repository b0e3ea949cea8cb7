// agz_device.hpp — shared device helpers: Philox streams, the exact-mode exp, wave primitives, tree layout.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "agz_games.hpp"

namespace agz {

// ---------------------------------------------------------------------------------------------------
// Randomness.  The reference draws CUDA.rand(maxLengthGame, L) per rollout (mcts_gpu.jl:397) and uses the
// Julia global RNG for move sampling (:520); both unseeded.  We define counter-based streams keyed by the
// GLOBAL game id so results do not depend on slot compaction or on how games are sharded over GPUs.
// ---------------------------------------------------------------------------------------------------
AGZ_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// uniforms in (0,1] standing for prob[cpt,i] (mcts_gpu.jl:178): one Philox block serves four consecutive depths,
// counter = (game, step, rollout, depth >> 2), word = depth & 3
AGZ_HD void uniform_search4(uint64_t seed, uint32_t game, uint32_t step, uint32_t rollout, uint32_t dquad, float u[4]) {
    uint32_t o[4];
    philox4x32_10(game, step, rollout, dquad, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    for (int i = 0; i < 4; ++i) u[i] = (float)((o[i] >> 8) + 1u) * 5.9604644775390625e-8f;
}
AGZ_HD float uniform_search(uint64_t seed, uint32_t game, uint32_t step, uint32_t rollout, uint32_t depth) {
    float u[4];
    uniform_search4(seed, game, step, rollout, depth >> 2, u);
    const uint32_t w = depth & 3u;
    return w == 0 ? u[0] : (w == 1 ? u[1] : (w == 2 ? u[2] : u[3]));
}
// uniform in (0,1) standing for rand() inside StatsBase.sample (mcts_gpu.jl:520): (23 random bits + 1/2) 2^-23 — the odd multiples
// of 2^-24, every one exactly representable in fp32 (no rounding in the conversion, so host and device agree whatever the
// compiler's intermediate precision; never 1: rand() is in [0,1)).  Never 0: the
// reference's Float64 rand() is 0 with probability 2^-53 — never in practice — whereas a 24-bit draw that could be 0 would stop the
// duel's all-actions walk (:606) at action 1 whatever its weight once in 2^24 draws
AGZ_HD float uniform_move(uint64_t seed, uint32_t game, uint32_t step) {
    uint32_t o[4];
    philox4x32_10(game, step, 0u, 0x80000000u, (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return (float)(2u * (o[0] >> 9) + 1u) * 5.9604644775390625e-8f;
}

// ---------------------------------------------------------------------------------------------------
// exp used by softmax / sigmoid in EXACT mode: Cephes-style range reduction + degree-5 polynomial written
// as explicit fma steps so that host (C oracle) and device agree bit for bit.
// ---------------------------------------------------------------------------------------------------
AGZ_HD float fma_rn(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmaf_rn(a, b, c);
#else
    return __builtin_fmaf(a, b, c);
#endif
}
AGZ_HD float exp_spec(float x) {
    if (x < -104.0f) return 0.0f;
    if (x > 88.5f) return __builtin_inff();
    float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma_rn(kf, -0.693145751953125f, x);
    r = fma_rn(kf, -1.42860682030941723212e-6f, r);
    float z = r * r;
    float p = 1.9875691500e-4f;
    p = fma_rn(p, r, 1.3981999507e-3f);
    p = fma_rn(p, r, 8.3334519073e-3f);
    p = fma_rn(p, r, 4.1665795894e-2f);
    p = fma_rn(p, r, 1.6666665459e-1f);
    p = fma_rn(p, r, 5.0000001201e-1f);
    float y = fma_rn(p, z, r);
    y = y + 1.0f;
    int k = (int)kf;
    union { uint32_t u; float f; } s;
    if (k >= -126) { s.u = (uint32_t)(k + 127) << 23; return y * s.f; }
    s.u = (uint32_t)(k + 127 + 64) << 23;
    return (y * s.f) * 5.42101086242752217e-20f;
}
// exp of the bf16-mode softmax (argument <= 0): 2^(x log2 e) from a degree-6 polynomial on the fraction and an exact scaling
// (v_ldexp_f32).  13 instructions; unlike v_exp_f32 it is a DEFINITION that a CPU restatement can follow, so the bf16
// mode is reproducible bit for bit.  2^t with t < -125 is 0.
AGZ_HD float exp2_spec(float x) {
    const float t = x * 1.44269504088896341f;
    if (!(t >= -125.0f)) return 0.0f;
    const float n = __builtin_rintf(t), f = t - n;
    float p = 1.5403530393381609e-4f;
    p = fma_rn(p, f, 1.3333558146428443e-3f);
    p = fma_rn(p, f, 9.6181291076284772e-3f);
    p = fma_rn(p, f, 5.5504108664821580e-2f);
    p = fma_rn(p, f, 2.4022650695910071e-1f);
    p = fma_rn(p, f, 6.9314718055994531e-1f);
    p = fma_rn(p, f, 1.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexpf(p, (int)n);
#else
    return __builtin_ldexpf(p, (int)n);
#endif
}
AGZ_HD float sigmoid_spec(float x) {      // NNlib sigma (DenseNet.jl:197, :301)
    float t = exp_spec(-__builtin_fabsf(x));
    return x >= 0.0f ? 1.0f / (1.0f + t) : t / (1.0f + t);
}

// ---------------------------------------------------------------------------------------------------
// Tree layout in HBM (per slot = one game tree, one wavefront works on it):
//   meta  [L][V]   u32   : node word  (parent | action<<8 | flags<<16)            256 B/slot at V=64
//   recs  [L][V]   rec   : [prior f32 x A2][rank u8 x A2][cid u8 x A2][edge {q, prior} x VL][visits u8 x VL]   (agz_tree_eager.hpp)
//   states[L][V]   Pos   : 80 B positions
// A2 = 8 lanes x KPL actions >= A.  Node 0 is the root.  No array is ever re-zeroed: a record is fully written
// when its node is expanded and only read while the node's EXPANDED bit is set.
// ---------------------------------------------------------------------------------------------------
enum : uint32_t {
    M_EXPANDED = 1u << 16,   // vnodes.expanded == 1
    M_STALE = 1u << 17,      // vnodes.uptodate != 1
    M_TERM = 1u << 18,       // isOver flag of the node's position
    M_TV_SHIFT = 19,         // 2 bits: terminal value*2 (0, 1, 2  ->  0.0, 0.5, 1.0)
    M_EVAL = 1u << 21,       // isOver already evaluated
    M_EXISTS = 1u << 22
};

struct TreePar {
    GamePar G;
    int32_t L, V;            // L = END of the slot range [slot0, L) of this launch
    int32_t slot0;           // first slot (sub-batches on parallel streams)
    int32_t gpw;             // games per wave (<= 8; fewer = sparse waves for small batches: the
                             // time of a rollout is the deepest descent among the games that share a wave / workgroup)
    uint32_t rec_bytes, off_q, off_vis, A2;
    uint8_t* recs;
    Pos* states;
    uint32_t* meta;
    uint32_t *ncount, *leaf, *game_id, *cnt_p, *cnt_new;
    uint32_t *wl, *wl_n, *sp;   // work lists [blocks][wl_cap], their lengths [blocks], last path node per slot [L]
    uint32_t wl_cap;
    int32_t fastdiv;         // operands of the tree's divisions are inside agz_fastdiv.hpp's range (bf16 mode, cpuct in [2^-10, 2^10])
    int32_t final_;          // this launch only closes the search (expand + backup of the last rollout): nothing is recomputed
    // network i/o
    void* planes;            // [L][INP] bf16 (or f32 when planes_f32)
    int32_t INP, planes_f32;
    const float* logits;     // [L][LGS]
    int32_t LGS;
    float* prior_eval;       // [L][A] softmaxed priors (capture / inject boundary, mcts_gpu.jl:414-417)
    float* v_eval;           // [L]
    float* policy_final;     // [L][A]
    uint64_t seed;
    uint32_t step, rollout;
    const uint32_t* slot_ply;     // [L] the ply ("step" of the uniforms' key) of every slot's game: one value for all slots in a plain search and in
                                  // a lock-step generation, per game once finished games' slots are refilled with new games (agz_selfplay with
                                  // more games than slots)
    float cpuct;
    int32_t training;
    int32_t do_reset, do_expand, do_select, last, exact, inject, capture;
    unsigned long long* dbg;      // diagnostic builds (-DAGZ_STAMPS): per-phase cycle sums; unused otherwise
    unsigned long long* rank_fault;   // rows by the root's legal rank: counts the expansions whose ROOT had more legal actions than the rows hold
                                      // (the dispatch picks the row width from a bound on the legal actions; a root that breaks the bound must not
                                      // lose actions silently: the generation fails)
};

// ---------------------------------------------------------------------------------------------------
// wave primitives (64 lanes)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// the lanes of the wave whose predicate holds — on a bool (hip's __ballot(int) leaves a 0/1 select and a compare in front of a mask that is already one)
__device__ __forceinline__ uint64_t wballot(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int ufirst(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t ufirst(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ float ufirst(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ uint64_t ufirst(uint64_t x) {
    uint32_t lo = ufirst((uint32_t)x), hi = ufirst((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ float rdlane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ uint32_t rdlane(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }

__device__ __forceinline__ float wave_max(float x) {
    for (int o = 32; o > 0; o >>= 1) { float y = __shfl_xor(x, o, 64); x = x > y ? x : y; }
    return x;
}
__device__ __forceinline__ int wave_sum_i(int x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
__device__ __forceinline__ float wave_sum_f(float x) {   // fixed butterfly order (deterministic, NOT source order)
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// One strictly ordered step of a left-to-right running sum across a 16-lane row:
//   acc[l] = acc[l-1] + x[l]   for lanes whose left neighbour is inside the row; the row's first lane keeps acc.
// (DPP row_shr:1 without bound_ctrl leaves lanes with no source lane unwritten.)
__device__ __forceinline__ float dpp_step(float acc, float x) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x));
    return acc;
}
__device__ __forceinline__ void dpp_step2(float& a, float xa, float& b, float xb) {
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b) : "v"(xa), "v"(xb));
}

// Source-order running sum over the 64 lanes of one register:
//   pre[l] = fl(..fl(fl(carry + x[0]) + x[1]) .. + x[l]),  carry' = pre[last].
// `nz` is the mask of lanes whose x is not +0: adding +0 is the identity, so 16-lane rows without a set bit
// are skipped and a row is only walked up to its last set lane.  Lanes past that point return an undefined
// prefix (callers only look at lanes with nz set).  When stop_u > 0 the walk ends after the first row whose
// running sum reaches stop_u (the reference's `break`, mcts_gpu.jl:178-180).
__device__ __forceinline__ float chain64(float x, uint64_t nz, float& carry, bool use_stop, float stop_u, bool& stopped) {
    const int lane = lane_id();
    float pre = carry;
    asm volatile("s_nop 4");
    for (int row = 0; row < 4; ++row) {
        uint32_t bits = (uint32_t)(nz >> (16 * row)) & 0xffffu;
        if (bits == 0 || stopped) continue;
        int hb = 31 - __builtin_clz(bits);
        float acc = (lane == 16 * row) ? carry + x : x;
        for (int s = 0; s < hb; ++s) acc = dpp_step(acc, x);
        carry = rdlane(acc, 16 * row + hb);
        pre = ((lane >> 4) == row) ? acc : pre;
        if (use_stop && carry >= stop_u) stopped = true;
    }
    return pre;
}
// two independent running sums over the same lane mask (Newton's S and g, mcts_gpu.jl:142-151)
__device__ __forceinline__ void chain64x2(float xa, float xb, uint64_t nz, float& ca, float& cb) {
    const int lane = lane_id();
    asm volatile("s_nop 4");
    for (int row = 0; row < 4; ++row) {
        uint32_t bits = (uint32_t)(nz >> (16 * row)) & 0xffffu;
        if (bits == 0) continue;
        int hb = 31 - __builtin_clz(bits);
        float a = (lane == 16 * row) ? ca + xa : xa;
        float b = (lane == 16 * row) ? cb + xb : xb;
        for (int s = 0; s < hb; ++s) dpp_step2(a, xa, b, xb);
        ca = rdlane(a, 16 * row + hb);
        cb = rdlane(b, 16 * row + hb);
    }
}

// wave-uniform load of a Pos (all lanes read the same 80 bytes; values are then forced scalar)
template <int NC> __device__ __forceinline__ WPos<NC> load_pos(const Pos* p) {
    WPos<NC> w;
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    for (int i = 0; i < NC; ++i) { w.p.c[i] = ufirst(q[i]); w.o.c[i] = ufirst(q[3 + i]); w.lg.c[i] = ufirst(q[6 + i]); }
    uint32_t tail = ufirst(*reinterpret_cast<const uint32_t*>(q + 9));
    w.player = (int)(int8_t)(tail & 0xff);
    w.aux = (int)(int8_t)((tail >> 8) & 0xff);
    return w;
}
template <int NC> __device__ __forceinline__ void store_pos(Pos* p, const WPos<NC>& w) {
    if (lane_id() == 0) *p = pack(w);
}

}  // namespace agz
