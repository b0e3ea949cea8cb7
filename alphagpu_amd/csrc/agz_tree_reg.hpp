// agz_tree_reg.hpp — PUCT tree kernel, fourth generation: node rows live in REGISTERS.
//
// G lanes per game tree (64/G trees per wavefront); lane `sub` of a group owns the contiguous block of KPL actions
// k = sub*KPL .. sub*KPL+KPL-1 of the node row (prior, q, visits|child), loaded straight from HBM with 16-B loads.
// KPL is a compile-time constant, so every O(A) loop is fully unrolled register arithmetic:
//   * order-free work (lambda*P/(alpha-Q), alpha0 max, counts, exp, legality) runs on all lanes at once;
//   * an ordered fp32 sum (prior_rem, sampling prefix, softmax denominator, normalize: mcts_gpu.jl:120-131, 172-181,
//     260-268) is walked by the lanes TAKING TURNS: lane t adds its KPL values to the carry handed over by lane t-1
//     (one DPP row_shr:1 per turn) — G*(KPL+2) instructions for the whole group, bit-identical to the source-order
//     loop.  The sampling prefix is then re-derived by all lanes in parallel from their now-known starting sums.
//   * Newton's child terms (:142-151) go through a small LDS table indexed by child node id (written while the row is
//     scanned), are compacted in creation order and summed by the group's lane 0.
// LDS per game: meta (V words) + child table (2 V floats) = 768 B at V = 64, i.e. 6 KiB per wave at G = 8.
#pragma once
#include "agz_tree_grp.hpp"
#include "agz_divpair.hpp"

namespace agz {

#ifndef AGZ_REG_WAVES
#define AGZ_REG_WAVES 4
#endif
enum { DPP_SHR1 = 0x111, DPP_SHL4 = 0x104, DPP_SHL8 = 0x108, DPP_QUAD_B3 = 0xFF, DPP_QUAD_B13 = 0xF5 };

// value held by the LAST lane of the group, in every lane
template <int G> __device__ __forceinline__ float grp_bcast_last(float xf) {
    int x = __float_as_int(xf);
    if (G == 2) x = dpp_mov<DPP_QUAD_B13, 0xF>(x, x);
    if (G == 16) x = dpp_mov<DPP_SHL8, 0x3>(x, x);            // lanes 0..7 <- lanes 8..15
    if (G >= 8) x = dpp_mov<DPP_SHL4, 0x5>(x, x);             // lanes 0..3 (8..11) <- lanes 4..7 (12..15)
    if (G >= 4) x = dpp_mov<DPP_QUAD_B3, 0xF>(x, x);
    return __int_as_float(x);
}
template <int G> __device__ __forceinline__ uint64_t grp_or64(uint64_t v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    if (G >= 2) { lo |= dpp_mov<DPP_XOR1, 0xF>(0, lo); hi |= dpp_mov<DPP_XOR1, 0xF>(0, hi); }
    if (G >= 4) { lo |= dpp_mov<DPP_XOR2, 0xF>(0, lo); hi |= dpp_mov<DPP_XOR2, 0xF>(0, hi); }
    if (G >= 8) { lo |= dpp_mov<DPP_HALF_MIRROR, 0xF>(0, lo); hi |= dpp_mov<DPP_HALF_MIRROR, 0xF>(0, hi); }
    if (G >= 16) { lo |= dpp_mov<DPP_MIRROR, 0xF>(0, lo); hi |= dpp_mov<DPP_MIRROR, 0xF>(0, hi); }
    return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32);
}
template <int D> __device__ __forceinline__ float lane_shl(float x) {      // value of lane + D (same 16-lane row), own value past the row's end
    return __int_as_float(dpp_mov<0x100 + D, 0xF>(__float_as_int(x), __float_as_int(x)));
}
__device__ __forceinline__ float lane_shr1(float x) { return __int_as_float(dpp_mov<DPP_SHR1, 0xF>(__float_as_int(x), __float_as_int(x))); }

// Source-order sum of the group's G*KPL values (lane sub holds block sub).  Returns the total in every lane and
// leaves in `start` the running sum BEFORE the lane's own block (its correct starting value).
template <int G, int KPL>
__device__ __forceinline__ float grp_ordered_sum(const float (&x)[KPL], int sub, float& start) {
    float a = 0.0f, st = 0.0f;
#pragma unroll 1                                                // keeps the kernel inside the instruction cache
    for (int t = 0; t < G; ++t) {
        const float carry = lane_shr1(a);                       // what the previous lane ended with
        const float s0 = sub == 0 ? 0.0f : carry;
        if (sub == t) st = s0;
        a = s0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) a += x[j];                // only lane t's result is final in turn t
    }
    start = st;
    return grp_bcast_last<G>(a);
}

struct RegLds { int meta, tabp, tabq, stride; };
__host__ __device__ inline RegLds reg_lds_layout(int V) {
    RegLds o;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    o.meta = 0;
    o.tabp = up16(V * 4);                                       // child table: prior / q by child node id; compacted in place
    o.tabq = o.tabp + up16(V * 4);
    o.stride = o.tabq + up16(V * 4);
    return o;
}

// WV = waves per SIMD the register budget is cut for: 4 (128 VGPRs, a few spills) keeps every wave of a 32768-game launch
// resident; 3 (no spills) is faster as soon as the launch fits 3 waves per SIMD.
// what changes from rollout to rollout (kept apart from TreePar so that a caller looping over rollouts — k_search_small —
// can leave the big parameter block in constant kernel-argument memory)
struct StepFlags { uint32_t rollout; int do_reset, do_expand, do_select, last, fin = 0; };   // fin: the launch only closes the search (eager kernel)

// LEAN: the caller guarantees V <= 64, V % 4 == 0, bf16 network mode and no inject / capture (k_search_small): the code for
// larger trees, odd tree sizes, the exact mode and the teacher-forcing hooks is compiled out — 3-5 % faster (less code in the instruction cache).
template <int FAM, int NC, int G, int KPL, bool LEAN = false>
__device__ __forceinline__ void rollout_reg_body(const TreePar& T, const StepFlags SF, uint8_t* const lds, const int bidx);

template <int FAM, int NC, int G, int KPL, int WV = AGZ_REG_WAVES>
__global__ __launch_bounds__(64, G <= 4 ? 2 : WV) void k_rollout_reg(const TreePar T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_reg[];
    const StepFlags SF = {T.rollout, T.do_reset, T.do_expand, T.do_select, T.last};
    rollout_reg_body<FAM, NC, G, KPL>(T, SF, lds_reg, (int)blockIdx.x);
}

// One wave: expand + backup of the previous rollout and select + encode of this one for the 64/G games of wave-block `bidx`
// (also called from k_search_small, agz_search_small.hpp, with the wave's own LDS window).
template <int FAM, int NC, int G, int KPL, bool LEAN>
__device__ __forceinline__ void rollout_reg_body(const TreePar& T, const StepFlags SF, uint8_t* const lds, const int bidx) {
    using GM = Game<FAM, NC>;
    constexpr bool REV = FAM == F_REV;
    constexpr int NG = 64 / G;
    constexpr int AP = G * KPL;                                  // padded row length (== T.A2)
    static_assert(KPL % 4 == 0, "block of actions per lane must be a multiple of 4");
    const GamePar& P = T.G;
    int lane_ = lane_id();   // PHASE setup
    asm volatile("" : "+v"(lane_));                           // opaque per call: per-lane addresses are not hoisted out of a caller's rollout loop
    const int lane = lane_ & 63, g = lane / G, sub = lane % G;
    const int GPW = T.gpw;                                        // games of this wave (<= NG)
    const int slot = T.slot0 + bidx * GPW + g;
    const bool live = g < GPW && slot < T.L;
    const bool lead = sub == 0;
    const int A = P.A, V = T.V, ROWS = (int)T.rec_bytes;
    const RegLds LO = reg_lds_layout(V);
    uint8_t* const mine = lds + (size_t)g * LO.stride;
    uint32_t* const mymeta = reinterpret_cast<uint32_t*>(mine + LO.meta);
    float* const tabp = reinterpret_cast<float*>(mine + LO.tabp);
    float* const tabq = reinterpret_cast<float*>(mine + LO.tabq);
    const int sl = live ? slot : 0;
    uint8_t* const myrecs = T.recs + (size_t)sl * V * ROWS;
    Pos* const mystates = T.states + (size_t)sl * V;
    float2* const myaux = T.aux + (size_t)sl * V;
    const uint32_t gbits_shift = (uint32_t)(g * G);
    const uint64_t gmask = G == 64 ? ~0ull : (((1ull << G) - 1ull) << gbits_shift);
    const int k0 = sub * KPL;                                    // first action of this lane's block
    const bool small = LEAN || V <= 64;                          // node ids fit a 64-bit set
    const bool inject = !LEAN && T.inject, capture = !LEAN && T.capture;
    const bool exact = !LEAN && T.exact, planes_f32 = !LEAN && T.planes_f32;   // (the lean build is the bf16-network build)
#ifdef AGZ_STAMPS
    unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + (size_t)NG * LO.stride);
    if (lane < 17) stamp_lds[lane] = lane == 16 ? __builtin_amdgcn_s_memtime() : 0ull;
#endif

    // ---- stage the meta rows of the wave's games (coalesced) ------------------------------------------
    uint32_t ncount = 1, leafn = 0;   // PHASE stage meta rows
    uint32_t* const gmeta = T.meta + (size_t)sl * V;              // every change of a meta word is written through
    if (SF.do_reset) {
        if (lead) { mymeta[0] = M_EXISTS; if (live) gmeta[0] = M_EXISTS; }
    } else {
        if (live) { ncount = T.ncount[slot]; leafn = T.leaf[slot]; }
        if constexpr (LEAN) {                                     // V <= 64: one 16-B piece per lane and game, all games in flight together
            const int v4 = V >> 2;
            uint4 buf[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                const int sj = T.slot0 + bidx * GPW + j;
                buf[j] = (lane < v4 && j < GPW && sj < T.L) ? reinterpret_cast<const uint4*>(T.meta + (size_t)sj * V)[lane] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < NG; ++j)
                if (lane < v4) *reinterpret_cast<uint4*>(lds + (size_t)j * LO.stride + LO.meta + (size_t)lane * 16) = buf[j];
        } else if ((V & 3) == 0 && V <= 256) {                    // all rows of the wave in flight together: one memory latency
            const int v4 = V >> 2, n4 = NG * v4;                   // 16-B pieces per game / per wave (<= 8 per lane)
            uint4 buf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                if (c < n4) {
                    const int j = c / v4, sj = T.slot0 + bidx * GPW + j;
                    buf[i] = (j < GPW && sj < T.L) ? reinterpret_cast<const uint4*>(T.meta + (size_t)sj * V)[c - j * v4] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                if (c < n4) { const int j = c / v4; *reinterpret_cast<uint4*>(lds + (size_t)j * LO.stride + LO.meta + (size_t)(c - j * v4) * 16) = buf[i]; }
            }
        } else {
            for (int j = 0; j < NG; ++j) {
                const int sj = T.slot0 + bidx * GPW + j;
                if (j >= GPW || sj >= T.L) break;
                uint32_t* dm = reinterpret_cast<uint32_t*>(lds + (size_t)j * LO.stride + LO.meta);
                for (int c = lane; c < V; c += 64) dm[c] = T.meta[(size_t)sj * V + c];
            }
        }
    }
    AGZ_WSYNC();
    STAMP(0);
    uint32_t add_p = 0, add_new = 0;

    // =============================================================================================
    // expand (mcts_gpu.jl:250-302) + backUp (:306-328) of the previous rollout's leaf
    // =============================================================================================
    if (SF.do_expand) {   // PHASE expand: load logits
        const int lf = (int)leafn;
        uint32_t ml = live ? mymeta[lf] : (uint32_t)M_TERM;
        const bool term = (ml & M_TERM) != 0;
        const bool doexp = live && !term;
        float vleaf = 0.0f;
        if (doexp) {
            vleaf = T.v_eval[slot];
            const WPos<NC> st = grp_load_pos<NC, REV>(mystates + lf);
            float x[KPL];
            const float* src = inject ? T.prior_eval + (size_t)slot * A : T.logits + (size_t)slot * T.LGS;
            if constexpr (LEAN) {                                     // logits rows are padded to LGS >= G*KPL floats: 16-B loads, then mask
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(src + k0 + j);
                    x[j] = a.x; x[j + 1] = a.y; x[j + 2] = a.z; x[j + 3] = a.w;
                }
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (k0 + j < A) ? x[j] : -__builtin_inff();
            } else {
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (k0 + j < A) ? src[k0 + j] : (inject ? 0.0f : -__builtin_inff());
            }
            if (!inject) {                                            // softmax!(prior) (:417), source-order sum   // PHASE expand: softmax
                float mx = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < KPL; ++j) mx = x[j] > mx ? x[j] : mx;
                mx = grp_max<G>(mx);
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (k0 + j < A) ? (exact ? exp_spec(x[j] - mx) : exp2_spec(x[j] - mx)) : 0.0f;
                float st0;
                const float s = grp_ordered_sum<G, KPL>(x, sub, st0);
#pragma unroll
                for (int j = 0; j < KPL; j += 2) div_pair(x[j], s, x[j + 1], s, x[j], x[j + 1]);
                if (capture) {
#pragma unroll
                    for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.prior_eval[(size_t)slot * A + k0 + j] = x[j];
                }
            }
            bool lg[KPL]; int nl = 0;                                 // legal mask; masked priors (:260-268 / :284-290)   // PHASE expand: legal mask + normalize sum
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                lg[j] = (k0 + j < A) && GM::canPlay(P, st, k0 + j);
                x[j] = lg[j] ? x[j] : 0.0f;
                nl += lg[j] ? 1 : 0;
            }
            nl = grp_sum<G>(nl);
            float st1;
            const float normalize = grp_ordered_sum<G, KPL>(x, sub, st1);
            const bool rootmix = lf == 0 && T.training;               // :270-275 vs :277-279, :292-294   // PHASE expand: mix / divide + write row
            const float Af = (float)nl;
            uint8_t* rec = myrecs + (size_t)lf * ROWS;
            int npos = 0;
            float qn_[KPL];
#pragma unroll
            for (int j = 0; j < KPL; j += 2)                                // one division serves both forms, two per call
                div_pair(rootmix ? 0.75f * x[j] : x[j], normalize, rootmix ? 0.75f * x[j + 1] : x[j + 1], normalize, qn_[j], qn_[j + 1]);
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const float qn = qn_[j];
                float pr = rootmix ? (lg[j] ? qn + 0.25f / Af : 0.0f) : qn;
                if (k0 + j >= A) pr = 0.0f;
                x[j] = pr;
                npos += pr > 0.0f ? 1 : 0;
            }
            if (__builtin_expect(lf == 0, 0)) {                       // root expansion (second launch of a search): rare, kept out of line
#pragma unroll
                for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)slot * A + k0 + j] = x[j];
            }
            npos = grp_sum<G>(npos);                                  // "A" of :125-131 never changes after the expansion
            if (lead) myaux[lf] = make_float2(0.0f, (float)npos);
#pragma unroll
            for (int j = 0; j < KPL; j += 4) {
                *reinterpret_cast<float4*>(rec + (size_t)(k0 + j) * 4) = make_float4(x[j], x[j + 1], x[j + 2], x[j + 3]);
                *reinterpret_cast<float4*>(rec + T.off_q + (size_t)(k0 + j) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(rec + T.off_vc + (size_t)(k0 + j) * 2) = make_uint2(0u, 0u);
            }
            ml |= M_EXPANDED;                                         // :256
            if (lead) { mymeta[lf] = ml; gmeta[lf] = ml; }
        } else if (__builtin_expect(live && lf == 0, 0)) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)slot * A + k0 + j] = 0.0f;   // terminal root
        }
        STAMP(2);
        // ---- backUp (:306-328): the group walks the path together (meta words are in LDS) and lane (i mod G) takes
        // ancestor i; the read-modify-writes of all ancestors are then issued at once, one memory latency per G levels.
        if (live) {   // PHASE backup
            const int tv2 = (int)((ml >> M_TV_SHIFT) & 3u);
            float valf = vleaf; double vald = 0.5 * (double)tv2;
            int cur = lf; uint32_t mcur = ml;
            while (cur != 0) {
                int mypar = -1, mymv = 0; float myvf = 0.0f; double myvd = 0.0;
                for (int i = 0; i < G && cur != 0; ++i) {
                    const int par = (int)(mcur & 0xffu), mv = (int)((mcur >> 8) & 0xffu);
                    if (i == sub) { mypar = par; mymv = mv; myvf = valf; myvd = vald; }
                    valf = 1.0f - valf; vald = 1.0 - vald;                               // :324
                    mcur = mymeta[par];
                    if (lead) { mymeta[par] = mcur | M_STALE; if (!(mcur & M_STALE)) gmeta[par] = mcur | M_STALE; }   // :321 uptodate = 0
                    cur = par;
                }
                if (mypar >= 0) {
                    uint8_t* rec = myrecs + (size_t)mypar * ROWS;
                    float* qp = reinterpret_cast<float*>(rec + T.off_q) + mymv;
                    uint16_t* vp = reinterpret_cast<uint16_t*>(rec + T.off_vc) + mymv;
                    const float q = *qp; const uint32_t vc = *vp;
                    const float vis = (float)(vc & 0xffu);
                    float nq;
                    if (term) nq = (float)(((double)(vis * q) + (1.0 - myvd)) / (double)(vis + 1.0f));
                    else nq = (vis * q + (1.0f - myvf)) / (vis + 1.0f);                  // :319
                    *qp = nq;
                    *vp = (uint16_t)(vc + 1u);                                           // :320
                }
            }
        }
        STAMP(4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        AGZ_WSYNC();
        STAMP(5);
    }

    // =============================================================================================
    // kdescendTree! (mcts_gpu.jl:100-199) + decoder (:202-223)
    // =============================================================================================
    if (SF.do_select) {   // PHASE select: setup
        const uint32_t gid = live ? T.game_id[slot] : 0u;
        int node = 0, depth = 0;
        uint32_t mn = live ? mymeta[0] : 0u;
        WPos<NC> lst; bool have_state = false;
        for (int i = 0; i < NC; ++i) { lst.p.c[i] = 0; lst.o.c[i] = 0; lst.lg.c[i] = 0; }
        lst.player = 1; lst.aux = 0;
        bool descending = live && (mn & M_EXPANDED);
        int create_from = -1, create_move = 0; uint32_t create_vc = 0;
        float uq[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        while (__ballot(descending)) {
            if (descending) {   // PHASE descent: row loads + Philox
                if (lead) ++add_p;
                // ---- this lane's block of the node row, straight from HBM
                const uint8_t* rec = myrecs + (size_t)node * ROWS;
                float p[KPL], q[KPL]; uint32_t vw[KPL / 2];                // vw[j] = vc[2j] | vc[2j+1] << 16
                const float2 ax = myaux[node];
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(rec + (size_t)(k0 + j) * 4);
                    p[j] = a.x; p[j + 1] = a.y; p[j + 2] = a.z; p[j + 3] = a.w;
                    const uint2 c = *reinterpret_cast<const uint2*>(rec + T.off_vc + (size_t)(k0 + j) * 2);
                    vw[j / 2] = c.x; vw[j / 2 + 1] = c.y;
                }
                const bool stale = (mn & M_STALE) != 0;
                if (stale) {
#pragma unroll
                    for (int j = 0; j < KPL; j += 4) {
                        const float4 b = *reinterpret_cast<const float4*>(rec + T.off_q + (size_t)(k0 + j) * 4);
                        q[j] = b.x; q[j + 1] = b.y; q[j + 2] = b.z; q[j + 3] = b.w;
                    }
                }
                if ((depth & 3) == 0) uniform_search4(T.seed, gid, T.step, SF.rollout, (uint32_t)depth >> 2, uq);   // overlaps the loads
                const int uw = depth & 3;
                const float u = uw == 0 ? uq[0] : (uw == 1 ? uq[1] : (uw == 2 ? uq[2] : uq[3]));
                STAMP(7);
                float alpha = 0.0f, lambda = 0.0f;   // PHASE descent: visit count, child mask, rank scatter
                float pol[KPL];
                if (stale) {                                               // :114
                    // :120-131.  prior_rem (the source-order sum of the priors of childless actions) and the count of positive
                    // priors change only when a child is created under the node: both come from aux[] (written by the expansion
                    // and by the creation step at the end of this kernel), only the visit total is re-counted here.
                    int vs = 0;
                    uint64_t cm = 0;
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const uint32_t c = (j & 1) ? (vw[j / 2] >> 16) : (vw[j / 2] & 0xffffu);
                        vs += (int)(c & 0xffu);
                        const uint32_t ch = c >> 8;
                        if (small) cm |= ch != 0 ? 1ull << ch : 0ull;      // set of this node's child ids
                        else if (ch != 0) { tabp[ch] = p[j]; tabq[ch] = q[j]; }   // child table by child node id
                    }
                    vs = grp_sum<G>(vs);
                    int nch = 0;
                    if (small) {
                        // children in creation order = ascending node id (:144-146): the rank of child id c among the set
                        // bits of the group's id mask IS its creation index -> the table is written already compacted
                        const uint64_t M = grp_or64<G>(cm);
                        nch = __popcll(M);
#pragma unroll
                        for (int j = 0; j < KPL; ++j) {
                            const uint32_t c = (j & 1) ? (vw[j / 2] >> 16) : (vw[j / 2] & 0xffffu);
                            const uint32_t ch = c >> 8;
                            if (ch != 0) { const int r = __popcll(M & ((1ull << ch) - 1ull)); tabp[r] = p[j]; tabq[r] = q[j]; }
                        }
                    }
                    const float nf = 1.0f + (float)vs, Af = ax.y;   // PHASE descent: lambda, alpha0
                    float prior_rem = ax.x;                                 // ordered (:122-124)
                    lambda = T.cpuct * __builtin_sqrtf(nf) / (Af + nf);    // :132
                    prior_rem *= lambda;                                    // :134
                    float am = 0.0f;                                        // :133-138
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const float lp = lambda * p[j];
                        const float gap = lp > 1e-4f ? lp : 1e-4f;
                        const float c = q[j] + gap;
                        am = c > am ? c : am;
                    }
                    alpha = grp_max<G>(am);
                    STAMP(8);
                    // V > 64: children in creation order = nodes i with parent(i) == node, ascending i (:144-146); compact the   // PHASE descent: child compaction (V > 64)
                    // id-indexed table in place
                    AGZ_WSYNC();
                    for (int base = 1; !small && base < (int)ncount; base += 8 * G) {
                        uint32_t mi[8]; float tp[8], tq[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = base + G * j + sub;
                            mi[j] = i < (int)ncount ? mymeta[i] : 0xffffffffu;
                            tp[j] = i < (int)ncount ? tabp[i] : 0.0f; tq[j] = i < (int)ncount ? tabq[i] : 0.0f;
                        }
                        AGZ_WSYNC();                                       // all reads of this batch precede its writes
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = base + G * j + sub;
                            const bool isc = i < (int)ncount && (int)(mi[j] & 0xffu) == node;
                            const uint32_t bits = (uint32_t)((__ballot(isc) & gmask) >> gbits_shift);
                            if (isc) {
                                const int pos = nch + __popc(bits & ((1u << sub) - 1u));   // pos < i: never an unread entry
                                tabp[pos] = tp[j]; tabq[pos] = tq[j];
                            }
                            nch += __popc(bits);
                        }
                        AGZ_WSYNC();
                    }
                    STAMP(9);
                    float err = __builtin_inff();   // PHASE descent: Newton
                    // element c of the Newton sums: c == 0 is the prior_rem term (S = prior_rem/alpha, g = -prior_rem/alpha^2,
                    // :142-143), c = 1..nch the children in creation order (:144-151).  With nch < G there is one element per
                    // lane; with more, blocks of G elements at a time.  Either way the ordered sums run lane to lane through DPP.
                    const bool fast = nch < G;
                    float top_l = 0.0f, qv_l = 0.0f;
                    if (sub == 0) top_l = prior_rem;
                    else if (sub <= nch && fast) { top_l = lambda * tabp[sub - 1]; qv_l = tabq[sub - 1]; }   // :147-148
                    for (int it = 0; it < 100; ++it) {                     // :141-162
                        float S = 0.0f, gg = 0.0f;
                        STAMP(10);
                        if (fast) {
                            float t = 0.0f, uu = 0.0f;
                            if (sub <= nch) { const float bot = alpha - qv_l; div_pair(top_l, bot, -top_l, bot * bot, t, uu); }
                            // the group's lead lane pulls element d from lane d (DPP row_shl:d) and adds them in order;
                            // lanes beyond nch contribute +0 (exact), the other lanes' sums are discarded
                            float a = t, b = uu;
#define AGZ_PULL(d) if (G > d) { a += lane_shl<d>(t); b += lane_shl<d>(uu); }
                            AGZ_PULL(1) AGZ_PULL(2) AGZ_PULL(3) AGZ_PULL(4) AGZ_PULL(5) AGZ_PULL(6) AGZ_PULL(7)
                            AGZ_PULL(8) AGZ_PULL(9) AGZ_PULL(10) AGZ_PULL(11) AGZ_PULL(12) AGZ_PULL(13) AGZ_PULL(14) AGZ_PULL(15)
#undef AGZ_PULL
                            S = __int_as_float(grp_bcast<G>(__float_as_int(a))); gg = __int_as_float(grp_bcast<G>(__float_as_int(b)));
                            STAMP(1);
                        } else {
                            // more children than lanes: blocks of G consecutive elements (element j0 + sub in lane sub), each block
                            // pulled into the lead lane's running sums exactly like the single block of the fast path
                            float a = 0.0f, b = 0.0f;
                            for (int j0 = 0; j0 <= nch; j0 += G) {
                                const int c = j0 + sub;
                                float t = 0.0f, uu = 0.0f;
                                if (c <= nch) {
                                    float top = prior_rem, qv = 0.0f;
                                    if (c > 0) { top = lambda * tabp[c - 1]; qv = tabq[c - 1]; }
                                    const float bot = alpha - qv;
                                    div_pair(top, bot, -top, bot * bot, t, uu);
                                }
                                if (j0 == 0) { a = t; b = uu; } else { a += t; b += uu; }
#define AGZ_PULL(d) if (G > d) { a += lane_shl<d>(t); b += lane_shl<d>(uu); }
                                AGZ_PULL(1) AGZ_PULL(2) AGZ_PULL(3) AGZ_PULL(4) AGZ_PULL(5) AGZ_PULL(6) AGZ_PULL(7)
                                AGZ_PULL(8) AGZ_PULL(9) AGZ_PULL(10) AGZ_PULL(11) AGZ_PULL(12) AGZ_PULL(13) AGZ_PULL(14) AGZ_PULL(15)
#undef AGZ_PULL
                            }
                            S = __int_as_float(grp_bcast<G>(__float_as_int(a))); gg = __int_as_float(grp_bcast<G>(__float_as_int(b)));
                            STAMP(3);
                        }
                        const float newerr = S - 1.0f;
                        if (newerr < 0.001f || newerr == err) break;
                        alpha -= newerr / gg;
                        err = newerr;
                        STAMP(6);
                    }
                    STAMP(10);
#pragma unroll   // PHASE descent: policy row
                    for (int j = 0; j < KPL; j += 2)                       // :165-169, two exact quotients per call (agz_divpair.hpp)
                        div_pair(lambda * p[j], alpha - q[j], lambda * p[j + 1], alpha - q[j + 1], pol[j], pol[j + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < KPL; ++j) pol[j] = p[j];           // policy == prior since expand (:297-299)
                }
                if (__builtin_expect(node == 0 && SF.last, 0)) {            // copy_pol (:330-339) of the last descent   // PHASE descent: copy_pol
#pragma unroll
                    for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)slot * A + k0 + j] = pol[j];
                }
                STAMP(11);
                // ---- sample (:172-182): ordered prefix by turns, then every lane re-derives its own prefixes   // PHASE descent: sampling prefix
                float st0;
                (void)grp_ordered_sum<G, KPL>(pol, sub, st0);
                int jhit = KPL, jpos = -1;                                 // first j with prefix >= u ; last j <= jhit with policy > 0
                {
                    float a = st0;
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        a += pol[j];
                        if (jhit == KPL) { if (pol[j] > 0.0f) jpos = j; if (a >= u) jhit = j; }
                    }
                }
                const uint32_t hitbits = (uint32_t)((__ballot(jhit < KPL) & gmask) >> gbits_shift);
                const int hl = hitbits ? __builtin_ctz(hitbits) : G;       // first lane of the group whose block crosses u
                // bestmove = last positive action at or before the crossing (whole row if there is no crossing)
                int cand = (sub <= hl && jpos >= 0) ? k0 + jpos : -1;
                if (G > 1) {
                    int y;
                    if (G >= 2) { y = dpp_mov<DPP_XOR1, 0xF>(-1, cand); cand = y > cand ? y : cand; }
                    if (G >= 4) { y = dpp_mov<DPP_XOR2, 0xF>(-1, cand); cand = y > cand ? y : cand; }
                    if (G >= 8) { y = dpp_mov<DPP_HALF_MIRROR, 0xF>(-1, cand); cand = y > cand ? y : cand; }
                    if (G >= 16) { y = dpp_mov<DPP_MIRROR, 0xF>(-1, cand); cand = y > cand ? y : cand; }
                }
                const int bestmove = cand;
                STAMP(13);
                if (bestmove < 0) {   // PHASE descent: child lookup / step
                    descending = false;                                    // reference would index [-1]; leaf = node
                } else {
                    // child id of bestmove: the owning lane looks it up in its block
                    int cv = 0;
#pragma unroll
                    for (int j = 0; j < KPL; ++j) {
                        const uint32_t c = (j & 1) ? (vw[j / 2] >> 16) : (vw[j / 2] & 0xffffu);
                        cv = (k0 + j == bestmove) ? (int)c : cv;
                    }
                    cv = grp_sum<G>(cv);
                    const uint32_t child = (uint32_t)cv >> 8;
                    if (child == 0) {                                      // :183-191: a new child is never expanded -> descent ends
                        create_from = node; create_move = bestmove; create_vc = (uint32_t)cv;
                        descending = false;
                    } else {
                        mn = mymeta[child];
                        node = (int)child;                                 // :192
                        descending = (mn & M_EXPANDED) != 0;
                    }
                    ++depth;
                }
            }
            AGZ_WSYNC();
            STAMP(12);
        }
        {   // the parent of the new child loses one childless action: its prior_rem is re-summed here, once per rollout and   // PHASE prior_rem re-sum
            // at a point where the whole wave is converged, instead of at every later visit (mcts_gpu.jl:120-124)
            // (its row is re-read rather than kept in registers across the loop: the loads share the latency of the state load)
            const uint8_t* rec = myrecs + (size_t)(create_from >= 0 ? create_from : 0) * ROWS;
            float m[KPL];
#pragma unroll
            for (int j = 0; j < KPL; j += 4) {
                const float4 a = *reinterpret_cast<const float4*>(rec + (size_t)(k0 + j) * 4);
                const uint2 c = *reinterpret_cast<const uint2*>(rec + T.off_vc + (size_t)(k0 + j) * 2);
                const bool on = create_from >= 0;
                m[j] = (on && (c.x & 0xff00u) == 0 && k0 + j != create_move) ? a.x : 0.0f;            // +0 terms are exact
                m[j + 1] = (on && (c.x >> 24) == 0 && k0 + j + 1 != create_move) ? a.y : 0.0f;
                m[j + 2] = (on && (c.y & 0xff00u) == 0 && k0 + j + 2 != create_move) ? a.z : 0.0f;
                m[j + 3] = (on && (c.y >> 24) == 0 && k0 + j + 3 != create_move) ? a.w : 0.0f;
            }
            float st0;
            const float prem = grp_ordered_sum<G, KPL>(m, sub, st0);
            if (live && lead && create_from >= 0) reinterpret_cast<float*>(myaux + create_from)[0] = prem;
        }
        if (live && create_from >= 0) {                                    // :183-191 node creation (at most one per rollout)   // PHASE create child (play, isOver)
            const uint32_t child = ncount; ncount += 1;
            const WPos<NC> ps = grp_load_pos<NC, REV>(mystates + create_from);
            lst = GM::play(P, ps, create_move);
            have_state = true;
            int rr; const bool f = GM::isOver(P, lst, rr);
            uint32_t mc = (uint32_t)create_from | ((uint32_t)create_move << 8) | M_EXISTS | M_EVAL;
            if (f) mc |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
            if (lead) {
                ++add_new;
                reinterpret_cast<uint16_t*>(myrecs + (size_t)create_from * ROWS + T.off_vc)[create_move] = (uint16_t)(create_vc | (child << 8));
                mystates[child] = pack(lst);
                mymeta[child] = mc; gmeta[child] = mc;
            }
            mn = mc; node = (int)child;
        }
        if (live) {   // PHASE leaf: isOver(root), encode planes
            if (!(mn & M_EVAL)) {                                           // root on the first rollout
                lst = grp_load_pos<NC, REV>(mystates + node); have_state = true;
                int rr; const bool f = GM::isOver(P, lst, rr);
                mn |= M_EVAL;
                if (f) mn |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
                if (lead) { mymeta[node] = mn; gmeta[node] = mn; }
            }
            if (!have_state) lst = grp_load_pos<NC, REV>(mystates + node);
            // decoder (:202-223): 8 planes (16 bytes of bf16, or 32 of fp32) per store, chunks dealt round-robin to the group
            if (G == 8 && !planes_f32) {
                // the 2 VS plane bits as one bit string W (side to move, then opponent); lane sub takes byte sub of every 64-bit word
                constexpr int NW = 2 * NC;
                uint64_t W[NW];
                const int VS = P.VS, sw = VS >> 6, sb = VS & 63;
#pragma unroll
                for (int i = 0; i < NW; ++i) W[i] = 0;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const int lo = 64 * i;
                    const uint64_t m = VS >= lo + 64 ? ~0ull : (VS > lo ? ((1ull << (VS - lo)) - 1ull) : 0ull);
                    const uint64_t pc = lst.p.c[i] & m, oc = lst.o.c[i] & m;
                    W[i] |= pc;
                    if (sw == NC - 1) { W[i + NC - 1] |= oc << sb; W[i + NC] |= sb ? oc >> (64 - sb) : 0ull; }
                    else W[i + NC] |= oc;                                  // VS == 64 NC
                }
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const int j0 = 64 * k + 8 * sub;
                    if (j0 < T.INP) {
                        const uint32_t f = (uint32_t)(W[k] >> (8 * sub)) & 0xffu;
                        uint4 o;
                        o.x = ((f & 1u) ? 0x3F80u : 0u) | ((f & 2u) ? 0x3F800000u : 0u); o.y = ((f & 4u) ? 0x3F80u : 0u) | ((f & 8u) ? 0x3F800000u : 0u);
                        o.z = ((f & 16u) ? 0x3F80u : 0u) | ((f & 32u) ? 0x3F800000u : 0u); o.w = ((f & 64u) ? 0x3F80u : 0u) | ((f & 128u) ? 0x3F800000u : 0u);
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(T.planes) + (size_t)slot * T.INP + j0) = o;
                    }
                }
            } else
            for (int j0 = 8 * sub; j0 < T.INP; j0 += 8 * G) {
                uint32_t w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = j0 + e;
                    bool bit = false;
                    if (j < P.VS) bit = bb_get(lst.p, j);
                    else if (j < 2 * P.VS) bit = bb_get(lst.o, j - P.VS);
                    w[e] = bit ? 1u : 0u;
                }
                if (planes_f32) {
                    float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(T.planes) + (size_t)slot * T.INP + j0);
                    d[0] = make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
                    d[1] = make_float4((float)w[4], (float)w[5], (float)w[6], (float)w[7]);
                } else {
                    uint4 o;
                    o.x = (w[0] ? 0x3F80u : 0u) | (w[1] ? 0x3F800000u : 0u); o.y = (w[2] ? 0x3F80u : 0u) | (w[3] ? 0x3F800000u : 0u);
                    o.z = (w[4] ? 0x3F80u : 0u) | (w[5] ? 0x3F800000u : 0u); o.w = (w[6] ? 0x3F80u : 0u) | (w[7] ? 0x3F800000u : 0u);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(T.planes) + (size_t)slot * T.INP + j0) = o;
                }
            }
            leafn = (uint32_t)node;
        }
    }

    STAMP(14);   // PHASE bookkeeping
    // ---- bookkeeping (the meta words were written through as they changed) --------------------------
    if (live && lead) {
        T.ncount[slot] = ncount;
        T.leaf[slot] = leafn;
        if (SF.do_reset) { T.cnt_p[slot] = add_p; T.cnt_new[slot] = add_new; }
        else { T.cnt_p[slot] += add_p; T.cnt_new[slot] += add_new; }
    }
#ifdef AGZ_STAMPS
    STAMP(15);
    if (lane < 16 && T.dbg) T.dbg[(size_t)bidx * 16 + lane] += stamp_lds[lane];
#endif
    (void)AP;
}

}  // namespace agz
