// agz_tree_grp.hpp — PUCT tree kernel, third generation: G lanes per game tree, 64/G trees per wavefront.
//
// The reference's per-node arithmetic has two kinds of work (mcts_gpu.jl:100-199, 250-302):
//   * order-free, O(A) wide   : divisions lambda*P/(alpha-Q), max, integer counts, exp, legality tests
//   * order-bound, O(A) long  : the fp32 sums prior_rem, S/g (Newton), the sampling prefix, normalize — these must be
//                               taken in source order to stay bit-identical to the reference.
// Measured extremes: G = 64 (agz_tree.hpp, one wave per tree) spends a whole wave instruction per step of every
// ordered sum (6900 instructions per game-rollout); G = 1 (one lane per tree, 64 trees per wave) makes the ordered
// sums free but leaves only 512 waves on 1024 SIMDs, each stalling on its own LDS/HBM latency.  Here the G lanes
// of a group split the order-free loops (stride G) and lane 0 of the group walks the ordered sums over values the
// group has parked in LDS; G = 8 gives 4096 waves at L = 32768.
//
// Node rows move as whole records: one LDS-DMA (global_load_lds_dwordx4, 16 B per lane, coalesced) per row and round;
// the LDS image of a game is an odd multiple of 16 B long, so lane-0 walks of different games never share a bank.
#pragma once
#include "agz_device.hpp"

namespace agz {

#ifdef AGZ_STAMPS
// diagnostic builds only: per-phase cycle sums kept in LDS (stamp_lds[0..15] sums, [16] last time stamp) and updated by the
// first ACTIVE lane, so that the attribution is right inside divergent code as well
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime();                                   \
                      if (lane_id() == (int)__builtin_ctzll(__ballot(1))) { stamp_lds[i] += t_ - stamp_lds[16]; stamp_lds[16] = t_; } } while (0)
#define CNT(i, n) do { if (lane_id() == (int)__builtin_ctzll(__ballot(1))) stamp_lds[i] += (n); } while (0)
#else
#define STAMP(i) do { } while (0)
#define CNT(i, n) do { } while (0)
#endif

#define AGZ_WSYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

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

struct GrpLds {                     // byte offsets inside one game's LDS image
    int row, meta, cbuf, cu, clist, stride;
};
__host__ __device__ inline GrpLds grp_lds_layout(int rec_bytes, int A4, int V) {
    GrpLds o;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    o.row = 0;                                                  // node record: prior | q (later: policy) | vc
    o.meta = up16(rec_bytes);
    o.cbuf = o.meta + up16((V + 1) * 4);                        // masked priors [A4], then Newton terms ct[V+1]
    const int cb = up16((V + 1) * 4);
    o.cu = o.cbuf + cb;                                         //                                  and cu[V+1]
    const int both = 2 * cb > up16(A4 * 4) ? 2 * cb : up16(A4 * 4);
    o.clist = o.cbuf + both;                                    // child actions in creation order [V]
    o.stride = o.clist + up16(V);
    if (((o.stride / 16) & 1) == 0) o.stride += 16;
    return o;
}

// LDS-DMA row gather: for every set bit j (a lane index with sub == 0) copy rowbytes from that lane's mysrc into
// game (j / G)'s image at byte offset dst_off.
template <int G>
__device__ __forceinline__ void grp_gather(uint64_t mask, const uint8_t* mysrc, uint8_t* lds, int gstride, int dst_off, int rowbytes) {
    const int lane = lane_id();
    const uint32_t lo = (uint32_t)(uintptr_t)mysrc, hi = (uint32_t)((uintptr_t)mysrc >> 32);
    while (mask) {
        const int j = __builtin_ctzll(mask); mask &= mask - 1;
        const uint8_t* s = reinterpret_cast<const uint8_t*>(((uint64_t)rdlane(hi, j) << 32) | rdlane(lo, j));
        uint8_t* d = lds + (size_t)(j / G) * gstride + dst_off;
        for (int c = 0; c < rowbytes; c += 1024)
            if (c + lane * 16 < rowbytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + c + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(d + c), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
template <int G>
__device__ __forceinline__ void grp_scatter(uint64_t mask, uint8_t* mydst, const uint8_t* lds, int gstride, int src_off, int rowbytes) {
    const int lane = lane_id();
    const uint32_t lo = (uint32_t)(uintptr_t)mydst, hi = (uint32_t)((uintptr_t)mydst >> 32);
    while (mask) {
        const int j = __builtin_ctzll(mask); mask &= mask - 1;
        uint8_t* d = reinterpret_cast<uint8_t*>(((uint64_t)rdlane(hi, j) << 32) | rdlane(lo, j));
        const uint8_t* s = lds + (size_t)(j / G) * gstride + src_off;
        for (int c = lane * 16; c < rowbytes; c += 1024) *reinterpret_cast<uint4*>(d + c) = *reinterpret_cast<const uint4*>(s + c);
    }
}

// group reductions / broadcast with DPP (a few cycles) instead of ds_bpermute (an LDS round trip each)
template <int CTRL, int BANK> __device__ __forceinline__ int dpp_mov(int old, int x) {
    return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xf, BANK, false);
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_QUAD_B0 = 0x00, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118 };
template <int G> __device__ __forceinline__ int grp_bcast(int x) {          // value of the group's lane 0
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "group size");
    if (G == 2) return dpp_mov<0xA0, 0xF>(x, x);                            // quad_perm [0,0,2,2]
    if (G == 16) x = dpp_mov<DPP_SHR8, 0xC>(x, x);                          // lanes 8..15 <- lanes 0..7
    if (G >= 8) x = dpp_mov<DPP_SHR4, 0xA>(x, x);                           // lanes 4..7 (12..15) <- lanes 0..3 (8..11)
    if (G >= 4) x = dpp_mov<DPP_QUAD_B0, 0xF>(x, x);
    return x;
}
template <int G> __device__ __forceinline__ float grp_bcast(float x) { return __int_as_float(grp_bcast<G>(__float_as_int(x))); }
template <int G> __device__ __forceinline__ int grp_sum(int x) {
    if (G >= 2) x += dpp_mov<DPP_XOR1, 0xF>(0, x);
    if (G >= 4) x += dpp_mov<DPP_XOR2, 0xF>(0, x);
    if (G >= 8) x += dpp_mov<DPP_HALF_MIRROR, 0xF>(0, x);
    if (G >= 16) x += dpp_mov<DPP_MIRROR, 0xF>(0, x);
    return x;
}
template <int G> __device__ __forceinline__ float grp_max(float x) {
    float y;
    if (G >= 2) { y = __int_as_float(dpp_mov<DPP_XOR1, 0xF>(0, __float_as_int(x))); x = y > x ? y : x; }
    if (G >= 4) { y = __int_as_float(dpp_mov<DPP_XOR2, 0xF>(0, __float_as_int(x))); x = y > x ? y : x; }
    if (G >= 8) { y = __int_as_float(dpp_mov<DPP_HALF_MIRROR, 0xF>(0, __float_as_int(x))); x = y > x ? y : x; }
    if (G >= 16) { y = __int_as_float(dpp_mov<DPP_MIRROR, 0xF>(0, __float_as_int(x))); x = y > x ? y : x; }
    return x;
}
// ordered sum of n floats (n multiple of 4, 16-B aligned LDS): 8 float4 reads in flight per batch, adds in source order
__device__ __forceinline__ float lds_ordered_sum(const float* p, int n, float acc) {
    for (int k0 = 0; k0 < n; k0 += 32) {
        float4 m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = (k0 + 4 * j < n) ? *reinterpret_cast<const float4*>(p + k0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc += m[j].x; acc += m[j].y; acc += m[j].z; acc += m[j].w; }
    }
    return acc;
}

template <int FAM, int NC, int G>
__global__ __launch_bounds__(64, G >= 16 ? 4 : (G >= 8 ? 2 : 1)) void k_rollout_grp(const TreePar T) {
    using GM = Game<FAM, NC>;
    constexpr bool REV = FAM == F_REV;
    constexpr int NG = 64 / G;
    const GamePar& P = T.G;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = lane_id(), g = lane / G, sub = lane % G;
    const int slot = (int)blockIdx.x * NG + g;
    const bool live = slot < T.L;
    const bool lead = sub == 0;
    const int A = P.A, A4 = (int)T.A2, V = T.V, ROWS = (int)T.rec_bytes;
    const GrpLds LO = grp_lds_layout(ROWS, A4, V);
    uint8_t* const mine = lds + (size_t)g * LO.stride;
    float* const rp = reinterpret_cast<float*>(mine + LO.row);
    float* const rq = reinterpret_cast<float*>(mine + LO.row + T.off_q);
    uint16_t* const rvc = reinterpret_cast<uint16_t*>(mine + LO.row + T.off_vc);
    float* const pol = rq;                                                   // the policy row overwrites q in place
    uint32_t* const mymeta = reinterpret_cast<uint32_t*>(mine + LO.meta);
    float* const mrow = reinterpret_cast<float*>(mine + LO.cbuf);            // masked priors (prior_rem), legal flags
    float* const ct = reinterpret_cast<float*>(mine + LO.cbuf);
    float* const cu = reinterpret_cast<float*>(mine + LO.cu);
    uint8_t* const clist = mine + LO.clist;
    const int sl = live ? slot : 0;
    uint8_t* const myrecs = T.recs + (size_t)sl * V * ROWS;
    Pos* const mystates = T.states + (size_t)sl * V;
    const uint32_t gbits_shift = (uint32_t)(g * G);
    const uint64_t gmask = G == 64 ? ~0ull : (((1ull << G) - 1ull) << gbits_shift);

    // ---- stage the meta rows of the wave's games (coalesced) ------------------------------------------
    uint32_t ncount = 1, leafn = 0;
    if (T.do_reset) {
        if (lead) mymeta[0] = M_EXISTS;
    } else {
        for (int j = 0; j < NG; ++j) {
            const int sj = (int)blockIdx.x * NG + j;
            if (sj >= T.L) break;
            uint32_t* dm = reinterpret_cast<uint32_t*>(lds + (size_t)j * LO.stride + LO.meta);
            for (int c = lane; c < V; c += 64) dm[c] = T.meta[(size_t)sj * V + c];
        }
        if (live) { ncount = T.ncount[slot]; leafn = T.leaf[slot]; }
    }
    AGZ_WSYNC();
    uint32_t add_p = 0, add_new = 0;

    // =============================================================================================
    // expand (mcts_gpu.jl:250-302) + backUp (:306-328) of the previous rollout's leaf
    // =============================================================================================
    if (T.do_expand) {
        const int lf = (int)leafn;
        uint32_t ml = live ? mymeta[lf] : (uint32_t)M_TERM;
        const bool term = (ml & M_TERM) != 0;
        const bool doexp = live && !term;
        const int inbytes = (A * 4 + 15) & ~15;
        if (T.inject) grp_gather<G>(__ballot(doexp && lead), reinterpret_cast<const uint8_t*>(T.prior_eval + (size_t)sl * A), lds, LO.stride, LO.row, inbytes);
        else grp_gather<G>(__ballot(doexp && lead), reinterpret_cast<const uint8_t*>(T.logits + (size_t)sl * T.LGS), lds, LO.stride, LO.row, inbytes);
        float vleaf = 0.0f;
        if (doexp) {
            vleaf = T.v_eval[slot];
            const WPos<NC> st = grp_load_pos<NC, REV>(mystates + lf);
            if (!T.inject) {                                          // softmax!(prior) (:417), source-order sum
                float mx = -__builtin_inff();
                for (int k = sub; k < A; k += G) mx = rp[k] > mx ? rp[k] : mx;
                mx = grp_max<G>(mx);
                for (int k = sub; k < A4; k += G) rp[k] = k < A ? (T.exact ? exp_spec(rp[k] - mx) : exp2_spec(rp[k] - mx)) : 0.0f;
                AGZ_WSYNC();
                float s = 0.0f;
                if (lead) s = lds_ordered_sum(rp, A4, 0.0f);
                s = grp_bcast<G>(s);
                for (int k = sub; k < A; k += G) {
                    const float p = rp[k] / s;
                    rp[k] = p;
                    if (T.capture) T.prior_eval[(size_t)slot * A + k] = p;
                }
            }
            int nl = 0;                                               // legal mask; masked priors (:260-268 / :284-290)
            for (int k = sub; k < A4; k += G) {                         // pad entries A..A4-1 become +0 (exact in the ordered sum)
                const bool lg = k < A && GM::canPlay(P, st, k);
                rp[k] = lg ? rp[k] : 0.0f;
                mrow[k] = lg ? 1.0f : 0.0f;
                nl += lg ? 1 : 0;
            }
            nl = grp_sum<G>(nl);
            AGZ_WSYNC();
            float normalize = 0.0f;
            if (lead) normalize = lds_ordered_sum(rp, A4, 0.0f);
            normalize = grp_bcast<G>(normalize);
            const bool rootmix = lf == 0 && T.training;               // :270-275 vs :277-279, :292-294
            const float Af = (float)nl;
            for (int k = sub; k < A4; k += G) {
                float pr = 0.0f;
                if (k < A) pr = rootmix ? (mrow[k] != 0.0f ? 0.75f * rp[k] / normalize + 0.25f / Af : 0.0f) : rp[k] / normalize;
                rp[k] = pr; rq[k] = 0.0f; rvc[k] = 0;
                if (lf == 0 && k < A) T.policy_final[(size_t)slot * A + k] = pr;
            }
            ml |= M_EXPANDED;                                         // :256
            if (lead) mymeta[lf] = ml;
        } else if (live && lf == 0) {
            for (int k = sub; k < A; k += G) T.policy_final[(size_t)slot * A + k] = 0.0f;   // terminal root
        }
        AGZ_WSYNC();
        grp_scatter<G>(__ballot(doexp && lead), myrecs + (size_t)lf * ROWS, lds, LO.stride, LO.row, ROWS);
        // ---- backUp (:306-328): the group walks the path together, lane (i mod G) updates ancestor i
        if (live) {
            const int tv2 = (int)((ml >> M_TV_SHIFT) & 3u);
            float valf = vleaf; double vald = 0.5 * (double)tv2;
            int cur = lf; uint32_t mcur = ml; int i = 0;
            while (cur != 0) {
                const int par = (int)(mcur & 0xffu), mv = (int)((mcur >> 8) & 0xffu);
                if ((i % G) == sub) {
                    uint8_t* rec = myrecs + (size_t)par * ROWS;
                    float* qp = reinterpret_cast<float*>(rec + T.off_q) + mv;
                    uint16_t* vp = reinterpret_cast<uint16_t*>(rec + T.off_vc) + mv;
                    const float q = *qp; const uint32_t vc = *vp;
                    const float vis = (float)(vc & 0xffu);
                    float nq;
                    if (term) nq = (float)(((double)(vis * q) + (1.0 - vald)) / (double)(vis + 1.0f));
                    else nq = (vis * q + (1.0f - valf)) / (vis + 1.0f);                  // :319
                    *qp = nq;
                    *vp = (uint16_t)(vc + 1u);                                           // :320
                }
                valf = 1.0f - valf; vald = 1.0 - vald;                                   // :324
                mcur = mymeta[par];
                if (lead) mymeta[par] = mcur | M_STALE;                                  // :321 uptodate = 0
                cur = par; ++i;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        AGZ_WSYNC();
    }

    // =============================================================================================
    // kdescendTree! (mcts_gpu.jl:100-199) + decoder (:202-223)
    // =============================================================================================
    if (T.do_select) {
        const uint32_t gid = live ? T.game_id[slot] : 0u;
        int node = 0, depth = 0;
        uint32_t mn = live ? mymeta[0] : 0u;
        WPos<NC> lst; bool have_state = false;
        for (int i = 0; i < NC; ++i) { lst.p.c[i] = 0; lst.o.c[i] = 0; lst.lg.c[i] = 0; }
        lst.player = 1; lst.aux = 0;
        bool descending = live && (mn & M_EXPANDED);
        int create_from = -1, create_move = 0; uint32_t create_vc = 0;
        float uq[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        uint64_t dmask = __ballot(descending && lead);
        while (dmask) {
            if ((depth & 3) == 0) uniform_search4(T.seed, gid, T.step, T.rollout, (uint32_t)depth >> 2, uq);   // independent of the rows: overlaps the DMA
            const int uw = depth & 3;
            const float u = uw == 0 ? uq[0] : (uw == 1 ? uq[1] : (uw == 2 ? uq[2] : uq[3]));
            grp_gather<G>(dmask, myrecs + (size_t)node * ROWS, lds, LO.stride, LO.row, ROWS);
            if (descending) {
                if (lead) ++add_p;
                float alpha = 0.0f, lambda = 0.0f;
                const bool stale = (mn & M_STALE) != 0;
                if (stale) {                                               // :114
                    int vs = 0, ac = 0;                                    // :120-131, order-free parts
                    for (int k = 4 * sub; k < A4; k += 4 * G) {
                        const float4 p4 = *reinterpret_cast<const float4*>(rp + k);
                        const uint2 v4 = *reinterpret_cast<const uint2*>(rvc + k);
                        vs += (int)(v4.x & 0xffu) + (int)((v4.x >> 16) & 0xffu) + (int)(v4.y & 0xffu) + (int)((v4.y >> 16) & 0xffu);
                        ac += (p4.x > 0.0f) + (p4.y > 0.0f) + (p4.z > 0.0f) + (p4.w > 0.0f);
                        float4 m4;                                         // prior of childless actions, +0 otherwise (adding +0 is exact)
                        m4.x = (v4.x & 0xff00u) == 0 ? p4.x : 0.0f; m4.y = (v4.x >> 24) == 0 ? p4.y : 0.0f;
                        m4.z = (v4.y & 0xff00u) == 0 ? p4.z : 0.0f; m4.w = (v4.y >> 24) == 0 ? p4.w : 0.0f;
                        *reinterpret_cast<float4*>(mrow + k) = m4;
                    }
                    vs = grp_sum<G>(vs); ac = grp_sum<G>(ac);
                    const float nf = 1.0f + (float)vs, Af = (float)ac;
                    AGZ_WSYNC();
                    float prior_rem = 0.0f;                                // ordered: childless priors in k order (:122-124)
                    if (lead) prior_rem = lds_ordered_sum(mrow, A4, 0.0f);
                    AGZ_WSYNC();
                    prior_rem = grp_bcast<G>(prior_rem);
                    lambda = T.cpuct * __builtin_sqrtf(nf) / (Af + nf);   // :132
                    prior_rem *= lambda;                                   // :134
                    float am = 0.0f;                                       // :133-138
                    for (int k = 4 * sub; k < A4; k += 4 * G) {
                        const float4 p4 = *reinterpret_cast<const float4*>(rp + k);
                        const float4 q4 = *reinterpret_cast<const float4*>(rq + k);
                        float lp, gap, c;
                        lp = lambda * p4.x; gap = lp > 1e-4f ? lp : 1e-4f; c = q4.x + gap; am = c > am ? c : am;
                        lp = lambda * p4.y; gap = lp > 1e-4f ? lp : 1e-4f; c = q4.y + gap; am = c > am ? c : am;
                        lp = lambda * p4.z; gap = lp > 1e-4f ? lp : 1e-4f; c = q4.z + gap; am = c > am ? c : am;
                        lp = lambda * p4.w; gap = lp > 1e-4f ? lp : 1e-4f; c = q4.w + gap; am = c > am ? c : am;
                    }
                    alpha = grp_max<G>(am);
                    // children in creation order = nodes i with parent(i) == node, ascending i (:144-146)
                    int nch = 0;
                    for (int base = 1; base < (int)ncount; base += 8 * G) {
                        uint32_t mi[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const int i = base + G * j + sub; mi[j] = i < (int)ncount ? mymeta[i] : 0xffffffffu; }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int i = base + G * j + sub;
                            const bool isc = i < (int)ncount && (int)(mi[j] & 0xffu) == node;
                            const uint32_t bits = (uint32_t)((__ballot(isc) & gmask) >> gbits_shift);
                            if (isc) {
                                const int pos = nch + __popc(bits & ((1u << sub) - 1u));
                                const int a = (int)((mi[j] >> 8) & 0xffu);
                                clist[pos] = (uint8_t)a;
                            }
                            nch += __popc(bits);
                        }
                    }
                    AGZ_WSYNC();
                    float err = __builtin_inff();
                    for (int j = 0; j < 100; ++j) {                        // :141-162
                        for (int c = sub; c <= nch; c += G) {
                            float top = prior_rem, qv = 0.0f;              // c == 0: S = prior_rem/alpha, g = -prior_rem/alpha^2 (:142-143)
                            if (c > 0) { const int a = clist[c - 1]; top = lambda * rp[a]; qv = rq[a]; }   // :147-148
                            const float bot = alpha - qv;
                            ct[c] = top / bot;
                            cu[c] = -top / (bot * bot);
                        }
                        AGZ_WSYNC();
                        float S = 0.0f, gg = 0.0f;
                        if (lead) {
                            S = ct[0]; gg = cu[0];
                            for (int c0 = 1; c0 <= nch; c0 += 4) {
                                float tv[4], uv[4];
#pragma unroll
                                for (int j = 0; j < 4; ++j) { tv[j] = c0 + j <= nch ? ct[c0 + j] : 0.0f; uv[j] = c0 + j <= nch ? cu[c0 + j] : 0.0f; }
#pragma unroll
                                for (int j = 0; j < 4; ++j) { S += tv[j]; gg += uv[j]; }
                            }
                        }
                        S = grp_bcast<G>(S); gg = grp_bcast<G>(gg);
                        AGZ_WSYNC();
                        const float newerr = S - 1.0f;
                        if (newerr < 0.001f || newerr == err) break;
                        alpha -= newerr / gg;
                        err = newerr;
                    }
                }
                // policy row (:165-169, or the prior while the node is up to date)
                const bool wr_final = node == 0 && T.last;                 // copy_pol (:330-339) of the last descent
                for (int k = 4 * sub; k < A4; k += 4 * G) {
                    float4 d = *reinterpret_cast<const float4*>(rp + k);
                    if (stale) {
                        const float4 q4 = *reinterpret_cast<const float4*>(rq + k);
                        d.x = lambda * d.x / (alpha - q4.x); d.y = lambda * d.y / (alpha - q4.y);
                        d.z = lambda * d.z / (alpha - q4.z); d.w = lambda * d.w / (alpha - q4.w);
                    }
                    *reinterpret_cast<float4*>(pol + k) = d;
                    if (wr_final) {
                        float* pf = T.policy_final + (size_t)slot * A;
                        if (k < A) pf[k] = d.x;
                        if (k + 1 < A) pf[k + 1] = d.y;
                        if (k + 2 < A) pf[k + 2] = d.z;
                        if (k + 3 < A) pf[k + 3] = d.w;
                    }
                }
                AGZ_WSYNC();
                // ---- sample (:172-182), ordered prefix on the group's lane 0
                int bestmove = -1;
                if (lead) {
                    float pr = 0.0f; int kb = A4;                           // kb = first k with running sum >= u (A4: none)
                    for (int k0 = 0; k0 < A4 && kb == A4; k0 += 16) {
                        float sp[16];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 d = (k0 + 4 * j < A4) ? *reinterpret_cast<const float4*>(pol + k0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
                            sp[4 * j] = d.x; sp[4 * j + 1] = d.y; sp[4 * j + 2] = d.z; sp[4 * j + 3] = d.w;
                        }
                        float mx = -__builtin_inff();
#pragma unroll
                        for (int j = 0; j < 16; ++j) { pr += sp[j]; sp[j] = pr; mx = pr > mx ? pr : mx; }
                        if (mx >= u) {                                      // some prefix of this batch reaches u: find the first
                            int f = 15;
#pragma unroll
                            for (int j = 14; j >= 0; --j) f = sp[j] >= u ? j : f;
                            kb = k0 + f;
                        }
                    }
                    if (kb > A4 - 1) kb = A4 - 1;
                    int k = kb;                                             // bestmove = last k' <= kb with policy > 0
                    for (; k >= 0; --k) if (pol[k] > 0.0f) { bestmove = k; break; }
                }
                bestmove = grp_bcast<G>(bestmove);
                if (bestmove < 0) {
                    descending = false;                                    // reference would index [-1]; leaf = node
                } else {
                    const uint32_t child = (uint32_t)rvc[bestmove] >> 8;
                    if (child == 0) {                                      // :183-191: a new child is never expanded -> descent ends;
                        create_from = node; create_move = bestmove;       //           it is materialised once, after the loop
                        create_vc = (uint32_t)rvc[bestmove];
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
            dmask = __ballot(descending && lead);
        }
        if (live && create_from >= 0) {                                    // :183-191 node creation (at most one per rollout)
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
                mymeta[child] = mc;
            }
            mn = mc; node = (int)child;
        }
        if (live) {
            if (!(mn & M_EVAL)) {                                           // root on the first rollout
                lst = grp_load_pos<NC, REV>(mystates + node); have_state = true;
                int rr; const bool f = GM::isOver(P, lst, rr);
                mn |= M_EVAL;
                if (f) mn |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
                if (lead) mymeta[node] = mn;
            }
            if (!have_state) lst = grp_load_pos<NC, REV>(mystates + node);
            // decoder (:202-223): 8 planes (16 bytes of bf16, or 32 of fp32) per store, chunks dealt round-robin to the group
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
                if (T.planes_f32) {
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
    // ---- write back bookkeeping ---------------------------------------------------------------------
    AGZ_WSYNC();
    for (int j = 0; j < NG; ++j) {
        const int sj = (int)blockIdx.x * NG + j;
        if (sj >= T.L) break;
        const int nj = (int)rdlane(ncount, j * G);
        const uint32_t* sm = reinterpret_cast<const uint32_t*>(lds + (size_t)j * LO.stride + LO.meta);
        for (int c = lane; c < nj; c += 64) T.meta[(size_t)sj * V + c] = sm[c];
    }
    if (live && lead) {
        T.ncount[slot] = ncount;
        T.leaf[slot] = leafn;
        if (T.do_reset) { T.cnt_p[slot] = add_p; T.cnt_new[slot] = add_new; }
        else { T.cnt_p[slot] += add_p; T.cnt_new[slot] += add_new; }
    }
}

}  // namespace agz
