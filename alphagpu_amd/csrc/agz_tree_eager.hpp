// agz_tree_eager.hpp — PUCT tree kernel, fifth generation: the regularised policy is computed EAGERLY, by the backup.
//
// The reference recomputes a node's policy row lazily, at the next visit after a backup has passed through it
// (mcts_gpu.jl:114-169: uptodate is cleared by backUp :321 and never set again), inside the descent.  A descent is a chain of
// dependent node visits whose length differs from game to game: with 8 games per wavefront the wave runs to the deepest of
// its games (8.4 rounds for a mean depth of 4.4 on Gobang 9x9), so nearly half of the lanes idle through the expensive part —
// lambda, alpha0, Newton, 81 IEEE divisions, the ordered prefix.  But the row a visit will find depends only on the node's own
// state after the LAST backup through it, and every node a backup passes through is known when the backup starts.  So:
//   * backup + recompute: every node on the path of the previous rollout is one independent WORK ITEM (update q / visits of
//     the edge taken, then lambda, alpha0, Newton, the policy row and its source-order running sums), and the items of all 8
//     games of the wave are dealt to the 8 lane-groups 8 at a time — ceil(sum of depths / 8) balanced rounds instead of
//     max(depth) divergent ones;
//   * the descent only reads the stored running sums: the sampled action of :172-182 is the number of entries of the
//     nondecreasing row cum[] that are < u (one compare per action, no ordered sum, no division), the child id comes from a
//     byte array stored next to it.  ~70 instructions per round instead of ~1000.
// Same arithmetic, same order of every fp32 operation as the reference (and as agz_tree_reg.hpp, which stays as the cross-check):
// only the time at which a row is computed changes.  Details:
//   rec[L][V]  [prior f32 x A2][q f32 x A2][vc u16 x A2], vc = visits | (creation rank of the child + 1) << 8
//   sel[L][V]  [cum f32 x A2][cid u8 x A2]: cum[k] = fl(cum[k-1] + policy[k]) (+inf for k >= A), cid = child node id | expanded << 7
//   aux[L][V]  {prior_rem before lambda (:120-124), -, npos | nvis << 8 | nch << 16 | lastpos << 24, -}
//   wl[block][8 V] work list of the wave: one word per expanded node passed below which the descent went on; sp[slot]: the last
//   expanded node of the path (the parent of the leaf) — these 8 items are processed together in the first round because they
//   alone may have a new child to register (creation rank, cid, re-summed prior_rem).
#pragma once
#include "agz_tree_reg.hpp"

namespace agz {

enum : uint32_t { SP_VALID = 1u << 24, SP_CREATED = 1u << 25 };

#ifdef AGZ_STAMPS
#define STAMPW(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); STAMP(i); } while (0)   // waits are charged to the phase that issued the loads
#else
#define STAMPW(i) do { } while (0)
#endif

struct EagerLds { int tabp, tabq, tstride, val, utab, total; };
__host__ __device__ inline EagerLds eager_lds_layout(int V) {
    EagerLds o;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    o.tabp = 0;                                                  // per lane-group: Newton inputs in creation order
    o.tabq = up16(V * 4);
    o.tstride = 2 * up16(V * 4);
    o.val = 8 * o.tstride;                                       // per game: {value_1, value_2, flags, -}
    o.utab = o.val + 8 * 16;                                     // per game: 32 uniforms (depths 0..31)
    o.total = o.utab + 8 * 128;
    return o;
}

// source-order running sums over the group's 8*KPL values (lane sub holds block sub): returns the sum of everything BEFORE the
// lane's own block — the lanes take turns, lane t adds its KPL values to what lane t-1 ended with (one DPP row_shr:1 per turn),
// bit-identical to the source-order loop.  The last lane's start needs no turn of its own; its end (the total) does.
template <int KPL, bool WANT_TOTAL>
__device__ __forceinline__ float grp_ordered_start(const float (&x)[KPL], int sub, float& total) {
    float a = 0.0f, st = 0.0f;
#pragma unroll 1
    for (int t = 0; t < (WANT_TOTAL ? 8 : 7); ++t) {
        const float carry = lane_shr1(a);                       // what the previous lane ended with
        const float s0 = sub == 0 ? 0.0f : carry;
        if (sub == t) st = s0;
        a = s0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) a += x[j];                // only lane t's result is final in turn t
    }
    if (WANT_TOTAL) total = grp_bcast_last<8>(a);
    else { const float carry = lane_shr1(a); if (sub == 7) st = carry; }
    return st;
}

template <int FAM, int NC, int KPL, bool LEAN>
__device__ __forceinline__ void rollout_eager_body(const TreePar& T, const StepFlags SF, uint8_t* const lds, const int bidx) {
    using GM = Game<FAM, NC>;
    constexpr bool REV = FAM == F_REV;
    constexpr int G = 8, NG = 8;
    static_assert(KPL % 4 == 0, "block of actions per lane must be a multiple of 4");
    const GamePar& P = T.G;
    int lane_ = lane_id();   // PHASE setup
    asm volatile("" : "+v"(lane_));                              // opaque per call (see rollout_reg_body)
    const int lane = lane_ & 63, g = lane / G, sub = lane % G;
    const int GPW = T.gpw;
    const int slot_base = T.slot0 + bidx * GPW;
    const int slot = slot_base + g;
    const bool live = g < GPW && slot < T.L;
    const bool lead = sub == 0;
    const int A = P.A, V = T.V, ROWS = (int)T.rec_bytes, SELB = (int)T.sel_bytes;
    const EagerLds LO = eager_lds_layout(V);
    float* const tabp = reinterpret_cast<float*>(lds + (size_t)g * LO.tstride + LO.tabp);
    float* const tabq = reinterpret_cast<float*>(lds + (size_t)g * LO.tstride + LO.tabq);
    float4* const valtab = reinterpret_cast<float4*>(lds + LO.val);
    float* const utab = reinterpret_cast<float*>(lds + LO.utab);
    const int sl = live ? slot : 0;
    const int k0 = sub * KPL;
    const bool inject = !LEAN && T.inject, capture = !LEAN && T.capture;
    const bool exact = !LEAN && T.exact, planes_f32 = !LEAN && T.planes_f32;
    const size_t wl_base = (size_t)(T.slot0 / NG + bidx) * (size_t)T.wl_cap;      // this wave's work list
    const int wl_block = T.slot0 / NG + bidx;
    uint32_t* const gmeta = T.meta + (size_t)sl * V;

#ifdef AGZ_STAMPS
    unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + LO.total);
    if (lane < 17) stamp_lds[lane] = lane == 16 ? __builtin_amdgcn_s_memtime() : 0ull;
    AGZ_WSYNC();
#endif
    uint32_t ncount = 1, leafn = 0;
    if (SF.do_reset) {
        if (live && lead) { gmeta[0] = M_EXISTS; T.sp[slot] = 0u; }
        if (lane == 0) T.wl_n[wl_block] = 0u;
    } else if (live) { ncount = T.ncount[slot]; leafn = T.leaf[slot]; }
    uint32_t add_p = 0, add_new = 0;
    STAMPW(0);

    // =============================================================================================
    // expand (mcts_gpu.jl:250-302) of the previous rollout's leaf, then backUp (:306-328) + the recomputation of every row
    // the backup makes stale (:114-169)
    // =============================================================================================
    if (SF.do_expand) {   // PHASE expand: load logits
        // ---------------------------------------------------------------------------- expand (lane-group g = game g)
        const int lf = (int)leafn;
        uint32_t ml = live ? gmeta[lf] : (uint32_t)M_TERM;
        const bool term = (ml & M_TERM) != 0;
        const bool doexp = live && !term;
        float vleaf = 0.0f;
        const uint32_t spw = live ? T.sp[slot] : 0u;
        if (doexp) {
            vleaf = T.v_eval[slot];
            const WPos<NC> st = grp_load_pos<NC, REV>(T.states + (size_t)sl * V + lf);
            float x[KPL];
            const float* src = inject ? T.prior_eval + (size_t)slot * A : T.logits + (size_t)slot * T.LGS;
            if constexpr (LEAN) {
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
                float s;
                (void)grp_ordered_start<KPL, true>(x, sub, s);
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
            float normalize;
            (void)grp_ordered_start<KPL, true>(x, sub, normalize);
            const bool rootmix = lf == 0 && T.training;               // :270-275 vs :277-279, :292-294   // PHASE expand: mix / divide
            const float Af = (float)nl;
            float qn_[KPL];
#pragma unroll
            for (int j = 0; j < KPL; j += 2)
                div_pair(rootmix ? 0.75f * x[j] : x[j], normalize, rootmix ? 0.75f * x[j + 1] : x[j + 1], normalize, qn_[j], qn_[j + 1]);
            int npos = 0, lastpos = -1;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                float pr = rootmix ? (lg[j] ? qn_[j] + 0.25f / Af : 0.0f) : qn_[j];
                if (k0 + j >= A) pr = 0.0f;
                x[j] = pr;
                npos += pr > 0.0f ? 1 : 0;
                lastpos = pr > 0.0f ? k0 + j : lastpos;
            }
            if (__builtin_expect(lf == 0, 0)) {                       // root expansion: policy == prior is what copy_pol sees for V <= 2
#pragma unroll
                for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)slot * A + k0 + j] = x[j];
            }
            npos = grp_sum<G>(npos);                                  // "A" of :125-131 never changes after the expansion
            {   int y;
                y = dpp_mov<DPP_XOR1, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos;
                y = dpp_mov<DPP_XOR2, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos;
                y = dpp_mov<DPP_HALF_MIRROR, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos; }
            // policy = prior (:297-299): the running sums the first revisit will sample from; their total is prior_rem (:120-124,   // PHASE expand: running sums + write rows
            // no child yet)
            float total;
            const float st0 = grp_ordered_start<KPL, true>(x, sub, total);
            uint8_t* rec = T.recs + ((size_t)sl * V + lf) * ROWS;
            uint8_t* srow = T.sel + ((size_t)sl * V + lf) * SELB;
            float c = st0;
#pragma unroll
            for (int j = 0; j < KPL; j += 4) {
                *reinterpret_cast<float4*>(rec + (size_t)(k0 + j) * 4) = make_float4(x[j], x[j + 1], x[j + 2], x[j + 3]);
                *reinterpret_cast<float4*>(rec + T.off_q + (size_t)(k0 + j) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<uint2*>(rec + T.off_vc + (size_t)(k0 + j) * 2) = make_uint2(0u, 0u);
                float4 cc;
                c += x[j]; cc.x = k0 + j < A ? c : __builtin_inff();
                c += x[j + 1]; cc.y = k0 + j + 1 < A ? c : __builtin_inff();
                c += x[j + 2]; cc.z = k0 + j + 2 < A ? c : __builtin_inff();
                c += x[j + 3]; cc.w = k0 + j + 3 < A ? c : __builtin_inff();
                *reinterpret_cast<float4*>(srow + (size_t)(k0 + j) * 4) = cc;
                *reinterpret_cast<uint32_t*>(srow + T.off_cid + (size_t)(k0 + j)) = 0u;
            }
            ml |= M_EXPANDED;                                         // :256
            if (lead) {
                gmeta[lf] = ml;
                T.aux4[(size_t)sl * V + lf] = make_uint4(__float_as_uint(total), 0u, (uint32_t)npos | ((uint32_t)(lastpos & 0xff) << 24), 0u);
            }
        } else if (__builtin_expect(live && lf == 0, 0)) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)slot * A + k0 + j] = 0.0f;   // terminal root
        }
        STAMPW(1);
        // ---- what the backup adds at the ancestors (:312-324): value_1 = 1 - v at even levels (the parent is level 0), value_2 =   // PHASE values of the backup
        // 1 - value_1 at odd ones — the alternation value <- 1 - value is 2-periodic from its first step (1 - x is exact for x in
        // [0.5, 1], and one of value_1, value_2 lies there); a terminal leaf starts from (1 + player*r)/2 in Float64 (:314)
        if (lead) {
            const int tv2 = (int)((ml >> M_TV_SHIFT) & 3u);
            const float v0 = term ? 0.5f * (float)tv2 : vleaf;
            const float v1 = 1.0f - v0, v2 = 1.0f - v1;
            valtab[g] = make_float4(v1, v2, __uint_as_float((term ? 1u : 0u) | (((spw >> 16) & 0xffu) << 8) | ((uint32_t)lf << 16) | (doexp ? 1u << 24 : 0u)), 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // rows written above are read by the items below (another lane-group may own them)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        AGZ_WSYNC();

        STAMPW(2);
        // ---------------------------------------------------------------------------- work items
        const uint32_t nwl = SF.do_reset ? 0u : ufirst(T.wl_n[wl_block]);   // PHASE items: loop control
        const int rounds = 1 + (int)((nwl + 7u) >> 3);
        const bool recompute = !(T.final_ || SF.fin);                             // after the last rollout of a search nobody descends again
#pragma unroll 1
        for (int r = 0; r < rounds; ++r) {
            // ---- which item does this lane-group take?  round 0: the parent of game g's leaf; later: entry 8 (r-1) + g of the list
            uint32_t ent = 0u; int gi = g; bool valid = false, special = r == 0;   // PHASE items: fetch item
            if (r == 0) { ent = spw; valid = live && (spw & SP_VALID); }
            else {
                const uint32_t idx = 8u * (uint32_t)(r - 1) + (uint32_t)g;
                if (idx < nwl) { ent = T.wl[wl_base + idx]; valid = true; gi = (int)(ent >> 24) & 7; }
            }
            if (!__ballot(valid)) continue;
            const int node = (int)(ent & 0xffu), move = (int)((ent >> 8) & 0xffu), dpt = (int)((ent >> 16) & 0xffu);
            const bool created = special && (ent & SP_CREATED);
            const int islot = valid ? slot_base + gi : sl;
            const float4 vt = valtab[gi];
            const uint32_t vflags = __float_as_uint(vt.z);
            const bool iterm = vflags & 1u;
            const int D = (int)((vflags >> 8) & 0xffu), ileaf = (int)((vflags >> 16) & 0xffu);
            const bool leaf_expanded = (vflags >> 24) & 1u;
            const int level = special ? 0 : D - 1 - dpt;
            const float w = (level & 1) ? vt.y : vt.x;                // 1 - value at this level
            uint8_t* const rec = T.recs + ((size_t)islot * V + node) * ROWS;
            uint8_t* const srow = T.sel + ((size_t)islot * V + node) * SELB;
            float p[KPL], q[KPL]; uint32_t vw[KPL / 2];   // PHASE items: row loads
            uint4 ax = make_uint4(0u, 0u, 0u, 0u);
            float pm = 0.0f, qm = 0.0f; uint32_t vcm = 0u;
            if (valid) {
                ax = T.aux4[(size_t)islot * V + node];
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(rec + (size_t)(k0 + j) * 4);
                    p[j] = a.x; p[j + 1] = a.y; p[j + 2] = a.z; p[j + 3] = a.w;
                    const float4 b = *reinterpret_cast<const float4*>(rec + T.off_q + (size_t)(k0 + j) * 4);
                    q[j] = b.x; q[j + 1] = b.y; q[j + 2] = b.z; q[j + 3] = b.w;
                    const uint2 c = *reinterpret_cast<const uint2*>(rec + T.off_vc + (size_t)(k0 + j) * 2);
                    vw[j / 2] = c.x; vw[j / 2 + 1] = c.y;
                }
                pm = reinterpret_cast<const float*>(rec)[move];
                qm = reinterpret_cast<const float*>(rec + T.off_q)[move];
                vcm = reinterpret_cast<const uint16_t*>(rec + T.off_vc)[move];
            } else {
#pragma unroll
                for (int j = 0; j < KPL; ++j) { p[j] = 0.0f; q[j] = 0.0f; }
#pragma unroll
                for (int j = 0; j < KPL / 2; ++j) vw[j] = 0u;
            }
            STAMPW(3);
            // ---- backUp of this edge (:319-320)   // PHASE items: backUp of the edge, prior_rem re-sum, q patch
            const float vis = (float)(vcm & 0xffu);
            float nq;
            if (__builtin_expect(__ballot(valid && iterm) != 0, 0)) {
                const float nqf = (vis * qm + w) / (vis + 1.0f);
                const float nqd = (float)(((double)(vis * qm) + (double)w) / (double)(vis + 1.0f));
                nq = iterm ? nqd : nqf;
            } else nq = (vis * qm + w) / (vis + 1.0f);
            const uint32_t npos = ax.z & 0xffu, nvis = ((ax.z >> 8) & 0xffu) + 1u, nch_old = (ax.z >> 16) & 0xffu;
            const uint32_t nch = nch_old + (created ? 1u : 0u);
            uint32_t nvc = vcm + 1u;
            if (created) nvc |= nch << 8;                             // creation rank + 1 (:183-191)
            float prem_raw = __uint_as_float(ax.x);                  // sum of the priors of childless actions, before lambda
            if (__builtin_expect(__ballot(valid && created) != 0, 0)) {
                // the node loses one childless action: re-sum prior_rem in source order (:120-124), once per rollout
                float m[KPL];
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    const uint32_t c = (j & 1) ? (vw[j / 2] >> 16) : (vw[j / 2] & 0xffffu);
                    m[j] = (created && (c >> 8) == 0 && k0 + j != move) ? p[j] : 0.0f;
                }
                float tot;
                (void)grp_ordered_start<KPL, true>(m, sub, tot);
                prem_raw = created ? tot : prem_raw;
            }
            if (valid && lead) {
                reinterpret_cast<float*>(rec + T.off_q)[move] = nq;
                reinterpret_cast<uint16_t*>(rec + T.off_vc)[move] = (uint16_t)nvc;
                if (special) srow[T.off_cid + move] = (uint8_t)((uint32_t)ileaf | (leaf_expanded ? 0x80u : 0u));
            }
            {   const int idx = move - k0;                            // the row in registers follows the update
#pragma unroll
                for (int j = 0; j < KPL; ++j) q[j] = (j == idx) ? nq : q[j];
            }
            if (!recompute) {
                if (valid && lead) T.aux4[(size_t)islot * V + node] = make_uint4(__float_as_uint(prem_raw), 0u, npos | (nvis << 8) | (nch << 16) | (ax.z & 0xff000000u), 0u);
                continue;
            }
            STAMPW(4);
            // ---- Newton inputs in creation order (:144-148): the rank stored with the child id is the place   // PHASE items: Newton inputs (rank scatter)
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const uint32_t c = (j & 1) ? (vw[j / 2] >> 16) : (vw[j / 2] & 0xffffu);
                const uint32_t rk = c >> 8;
                if (rk != 0) { tabp[rk - 1] = p[j]; tabq[rk - 1] = q[j]; }
            }
            AGZ_WSYNC();
            if (created && lead) { tabp[nch - 1] = pm; tabq[nch - 1] = nq; }
            AGZ_WSYNC();
            // ---- :116-138   // PHASE items: lambda, alpha0
            const float nf = 1.0f + (float)nvis, Af = (float)npos;
            const float lambda = T.cpuct * __builtin_sqrtf(nf) / (Af + nf);
            const float prior_rem = prem_raw * lambda;               // :134
            float am = 0.0f;
#pragma unroll
            for (int j = 0; j < KPL; ++j) {
                const float lp = lambda * p[j];
                const float gap = lp > 1e-4f ? lp : 1e-4f;
                const float c = q[j] + gap;
                am = c > am ? c : am;
            }
            float alpha = grp_max<G>(am);
            STAMPW(5);
            // ---- Newton (:141-162): element 0 is the prior_rem term, elements 1..nch the children in creation order   // PHASE items: Newton
            {
                float err = __builtin_inff();
                const bool fast = (int)nch < G;
                float top_l = 0.0f, qv_l = 0.0f;
                if (sub == 0) top_l = prior_rem;
                else if (sub <= (int)nch && fast) { top_l = lambda * tabp[sub - 1]; qv_l = tabq[sub - 1]; }
                for (int it = 0; it < 100; ++it) {
                    float S, gg;
                    if (fast) {
                        float t = 0.0f, uu = 0.0f;
                        if (sub <= (int)nch) { const float bot = alpha - qv_l; div_pair(top_l, bot, -top_l, bot * bot, t, uu); }
                        float a = t, b = uu;
#define AGZ_PULL(d) { a += lane_shl<d>(t); b += lane_shl<d>(uu); }
                        AGZ_PULL(1) AGZ_PULL(2) AGZ_PULL(3) AGZ_PULL(4) AGZ_PULL(5) AGZ_PULL(6) AGZ_PULL(7)
#undef AGZ_PULL
                        S = grp_bcast<G>(a); gg = grp_bcast<G>(b);
                    } else {
                        float a = 0.0f, b = 0.0f;
                        for (int j0 = 0; j0 <= (int)nch; j0 += G) {
                            const int c = j0 + sub;
                            float t = 0.0f, uu = 0.0f;
                            if (c <= (int)nch) {
                                float top = prior_rem, qv = 0.0f;
                                if (c > 0) { top = lambda * tabp[c - 1]; qv = tabq[c - 1]; }
                                const float bot = alpha - qv;
                                div_pair(top, bot, -top, bot * bot, t, uu);
                            }
                            if (j0 == 0) { a = t; b = uu; } else { a += t; b += uu; }
#define AGZ_PULL(d) { a += lane_shl<d>(t); b += lane_shl<d>(uu); }
                            AGZ_PULL(1) AGZ_PULL(2) AGZ_PULL(3) AGZ_PULL(4) AGZ_PULL(5) AGZ_PULL(6) AGZ_PULL(7)
#undef AGZ_PULL
                        }
                        S = grp_bcast<G>(a); gg = grp_bcast<G>(b);
                    }
                    const float newerr = S - 1.0f;
                    if (newerr < 0.001f || newerr == err) break;
                    alpha -= newerr / gg;
                    err = newerr;
                }
            }
            STAMPW(6);
            // ---- the policy row (:165-169) and its running sums (:172-181)   // PHASE items: policy row
            float pol[KPL];
#pragma unroll
            for (int j = 0; j < KPL; j += 2)
                div_pair(lambda * p[j], alpha - q[j], lambda * p[j + 1], alpha - q[j + 1], pol[j], pol[j + 1]);
            if (__builtin_expect(__ballot(valid && node == 0 && SF.last) != 0, 0)) {     // copy_pol (:330-339): the row the last descent samples from
                if (valid && node == 0) {
#pragma unroll
                    for (int j = 0; j < KPL; ++j) if (k0 + j < A) T.policy_final[(size_t)islot * A + k0 + j] = pol[j];
                }
            }
            int lastpos = -1;   // PHASE items: lastpos
#pragma unroll
            for (int j = 0; j < KPL; ++j) lastpos = pol[j] > 0.0f ? k0 + j : lastpos;
            {   int y;
                y = dpp_mov<DPP_XOR1, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos;
                y = dpp_mov<DPP_XOR2, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos;
                y = dpp_mov<DPP_HALF_MIRROR, 0xF>(-1, lastpos); lastpos = y > lastpos ? y : lastpos; }
            STAMPW(7);
            float dummy;   // PHASE items: running sums + stores
            float c = grp_ordered_start<KPL, false>(pol, sub, dummy);
            if (valid) {
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    float4 cc;
                    c += pol[j]; cc.x = k0 + j < A ? c : __builtin_inff();
                    c += pol[j + 1]; cc.y = k0 + j + 1 < A ? c : __builtin_inff();
                    c += pol[j + 2]; cc.z = k0 + j + 2 < A ? c : __builtin_inff();
                    c += pol[j + 3]; cc.w = k0 + j + 3 < A ? c : __builtin_inff();
                    *reinterpret_cast<float4*>(srow + (size_t)(k0 + j) * 4) = cc;
                }
                if (lead) T.aux4[(size_t)islot * V + node] = make_uint4(__float_as_uint(prem_raw), 0u, npos | (nvis << 8) | (nch << 16) | ((uint32_t)(lastpos & 0xff) << 24), 0u);
            }
            AGZ_WSYNC();                                              // the child table is rewritten by the next round
            STAMPW(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        AGZ_WSYNC();
        STAMPW(9);
    }

    // =============================================================================================
    // kdescendTree! (mcts_gpu.jl:100-199) over the stored running sums + decoder (:202-223)
    // =============================================================================================
    if (SF.do_select) {   // PHASE descent: uniforms (Philox)
        const uint32_t gid = live ? T.game_id[slot] : 0u;
        {   // prob[1..32, i] (:397): lane sub draws the uniforms of depths 4 sub .. 4 sub + 3
            float uq[4];
            uniform_search4(T.seed, gid, T.step, SF.rollout, (uint32_t)sub, uq);
            *reinterpret_cast<float4*>(utab + g * 32 + 4 * sub) = make_float4(uq[0], uq[1], uq[2], uq[3]);
        }
        AGZ_WSYNC();
        STAMPW(10);
        int node = 0, depth = 0;
        uint32_t mroot = live ? gmeta[0] : 0u;
        bool descending = live && (mroot & M_EXPANDED);
        int create_from = -1, create_move = 0;
        uint32_t spnew = 0u;
        uint32_t wcount = 0;                                          // wave-uniform: entries of the work list so far
        bool at_leaf_known = !descending;                             // leaf = root when the root is not expanded
        uint32_t mn = mroot;
        while (__ballot(descending)) {   // PHASE descent: rounds
            if (descending) {
                const uint8_t* srow = T.sel + ((size_t)sl * V + node) * SELB;
                float cum[KPL]; uint32_t cw[KPL / 4];
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(srow + (size_t)(k0 + j) * 4);
                    cum[j] = a.x; cum[j + 1] = a.y; cum[j + 2] = a.z; cum[j + 3] = a.w;
                    cw[j / 4] = *reinterpret_cast<const uint32_t*>(srow + T.off_cid + (size_t)(k0 + j));
                }
                STAMPW(11);
                float u;
                if (__builtin_expect(depth < 32, 1)) u = utab[g * 32 + depth];
                else u = uniform_search(T.seed, gid, T.step, SF.rollout, (uint32_t)depth);
                // bestmove (:172-182) = number of running sums below u (the row is nondecreasing; +inf beyond A)
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < KPL; ++j) cnt += cum[j] < u ? 1 : 0;
                int bestmove = grp_sum<G>(cnt);
                if (__builtin_expect(__ballot(bestmove >= A) != 0, 0)) {
                    // the row sums below u: the last positive action wins (:175-181)
                    if (bestmove >= A) {
                        const uint32_t lp = (T.aux4[(size_t)sl * V + node].z >> 24) & 0xffu;
                        bestmove = lp == 0xffu ? -1 : (int)lp;
                    }
                }
                if (lead) ++add_p;
                if (bestmove < 0) {                                   // reference would index [-1]; leaf = node
                    spnew = (uint32_t)depth << 16;
                    descending = false;
                } else {
                    const int idx = bestmove - k0;
                    uint32_t byte = 0u;
#pragma unroll
                    for (int j = 0; j < KPL; ++j) byte = (j == idx) ? ((cw[j / 4] >> (8 * (j & 3))) & 0xffu) : byte;
                    const int cv = grp_sum<G>((int)byte);
                    const int child = cv & 0x7f;
                    if (child == 0) {                                  // :183-191: a new child is never expanded -> the descent ends
                        create_from = node; create_move = bestmove;
                        spnew = (uint32_t)node | ((uint32_t)bestmove << 8) | ((uint32_t)(depth + 1) << 16) | SP_VALID | SP_CREATED;
                        descending = false;
                    } else if (cv & 0x80) {                            // expanded child: the descent goes on (:192)
                        const uint64_t app = __ballot(lead);           // (only lanes of descending groups are here)
                        if (lead) {
                            const uint32_t pos = wcount + (uint32_t)__popcll(app & ((1ull << lane) - 1ull));
                            T.wl[wl_base + pos] = (uint32_t)node | ((uint32_t)bestmove << 8) | ((uint32_t)depth << 16) | ((uint32_t)g << 24);
                        }
                        node = child;
                    } else {                                           // existing child that is not expanded: a terminal position
                        spnew = (uint32_t)node | ((uint32_t)bestmove << 8) | ((uint32_t)(depth + 1) << 16) | SP_VALID;
                        node = child;
                        at_leaf_known = true;
                        descending = false;
                    }
                    ++depth;
                }
            }
            // entries appended this round (wave-uniform): lead lanes of groups that are still descending
            wcount += (uint32_t)__popcll(__ballot(descending && lead));
            AGZ_WSYNC();
            STAMPW(12);
        }
        if (lane == 0) T.wl_n[wl_block] = wcount;
        if (live && lead) T.sp[slot] = spnew;
        (void)at_leaf_known;

        WPos<NC> lst; bool have_state = false;   // PHASE create child (play, isOver)
        for (int i = 0; i < NC; ++i) { lst.p.c[i] = 0; lst.o.c[i] = 0; lst.lg.c[i] = 0; }
        lst.player = 1; lst.aux = 0;
        if (live && create_from >= 0) {                                    // :183-191 node creation (at most one per rollout)
            const uint32_t child = ncount; ncount += 1;
            const WPos<NC> ps = grp_load_pos<NC, REV>(T.states + (size_t)sl * V + create_from);
            lst = GM::play(P, ps, create_move);
            have_state = true;
            int rr; const bool f = GM::isOver(P, lst, rr);
            uint32_t mc = (uint32_t)create_from | ((uint32_t)create_move << 8) | M_EXISTS | M_EVAL;
            if (f) mc |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
            if (lead) {
                ++add_new;
                T.states[(size_t)sl * V + child] = pack(lst);
                gmeta[child] = mc;
            }
            mn = mc; node = (int)child;
        } else if (live && node != 0) mn = gmeta[node];
        if (live) {
            if (!(mn & M_EVAL)) {                                           // root on the first rollout
                lst = grp_load_pos<NC, REV>(T.states + (size_t)sl * V + node); have_state = true;
                int rr; const bool f = GM::isOver(P, lst, rr);
                mn |= M_EVAL;
                if (f) mn |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
                if (lead) gmeta[node] = mn;
            }
            if (!have_state) lst = grp_load_pos<NC, REV>(T.states + (size_t)sl * V + node);
            // decoder (:202-223)   // PHASE encode planes
            if (!planes_f32) {
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
                    else W[i + NC] |= oc;
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
                float w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = j0 + e;
                    bool bit = false;
                    if (j < P.VS) bit = bb_get(lst.p, j);
                    else if (j < 2 * P.VS) bit = bb_get(lst.o, j - P.VS);
                    w[e] = bit ? 1.0f : 0.0f;
                }
                float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(T.planes) + (size_t)slot * T.INP + j0);
                d[0] = make_float4(w[0], w[1], w[2], w[3]);
                d[1] = make_float4(w[4], w[5], w[6], w[7]);
            }
            leafn = (uint32_t)node;
        }
    }

    STAMPW(13);
    // ---- bookkeeping ----------------------------------------------------------------------------------   // PHASE bookkeeping
    if (live && lead) {
        T.ncount[slot] = ncount;
        T.leaf[slot] = leafn;
        if (SF.do_reset) { T.cnt_p[slot] = add_p; T.cnt_new[slot] = add_new; }
        else { T.cnt_p[slot] += add_p; T.cnt_new[slot] += add_new; }
    }
#ifdef AGZ_STAMPS
    STAMPW(14);
    AGZ_WSYNC();
    if (lane < 15 && T.dbg) T.dbg[(size_t)(T.slot0 / NG + bidx) * 16 + lane] += stamp_lds[lane];
#endif
}

template <int FAM, int NC, int KPL, int WV = AGZ_REG_WAVES>
__global__ __launch_bounds__(64, WV) void k_rollout_eager(const TreePar T) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_eager[];
    const StepFlags SF = {T.rollout, T.do_reset, T.do_expand, T.do_select, T.last, T.final_};
    rollout_eager_body<FAM, NC, KPL, false>(T, SF, lds_eager, (int)blockIdx.x);
}

}  // namespace agz
