// agz_tree.hpp — the PUCT tree kernels: one 64-lane wavefront per game tree.
//
// k_rollout fuses, per rollout k, the reference's  expand(k-1) -> backUp(k-1) -> kdescendTree!(k) -> decoder(k)
// (mcts_gpu.jl:250-302, 306-328, 100-199, 202-223) so that the only kernel boundary per rollout is the
// batched network evaluation.  Lanes <-> actions (k = 64 r + lane); node rows are read with coalesced
// loads; every fp32 sum the reference performs sequentially is reproduced in SOURCE ORDER by a DPP
// row-shift chain (agz_device.hpp chain64), everything order-free (max, integer counts, divisions) runs
// lane-parallel.  Game logic (play / isOver) is wave-uniform and runs on scalar registers.
#pragma once
#include "agz_device.hpp"

namespace agz {

template <int NRV> struct MetaRegs {
    uint32_t m[NRV];
    __device__ __forceinline__ uint32_t get(int n) const {     // n wave-uniform
        uint32_t v = rdlane(m[0], n & 63);
        if (NRV > 1) { uint32_t w = rdlane(m[1], n & 63); v = (n >> 6) == 1 ? w : v; }
        if (NRV > 2) {
            uint32_t w2 = rdlane(m[2 % NRV], n & 63), w3 = rdlane(m[3 % NRV], n & 63);
            v = (n >> 6) == 2 ? w2 : v; v = (n >> 6) == 3 ? w3 : v;
        }
        return v;
    }
    __device__ __forceinline__ void set(int n, uint32_t val) {
        const int lane = lane_id();
        for (int r = 0; r < NRV; ++r) m[r] = (lane == (n & 63) && r == (n >> 6)) ? val : m[r];
    }
    __device__ __forceinline__ void orbits(int n, uint32_t bits) {
        const int lane = lane_id();
        for (int r = 0; r < NRV; ++r) m[r] |= (lane == (n & 63) && r == (n >> 6)) ? bits : 0u;
    }
};

// value of register array element idx (wave-uniform index) at a wave-uniform action
template <int NR> __device__ __forceinline__ float pick(const float (&x)[NR], int k) {
    float v = rdlane(x[0], k & 63);
    for (int r = 1; r < NR; ++r) { float w = rdlane(x[r], k & 63); v = (k >> 6) == r ? w : v; }
    return v;
}
template <int NR> __device__ __forceinline__ uint32_t pick(const uint32_t (&x)[NR], int k) {
    uint32_t v = rdlane(x[0], k & 63);
    for (int r = 1; r < NR; ++r) { uint32_t w = rdlane(x[r], k & 63); v = (k >> 6) == r ? w : v; }
    return v;
}
// per-lane gather x[a] for a per-lane action a
template <int NR> __device__ __forceinline__ float gather(const float (&x)[NR], int a) {
    float v = __shfl(x[0], a & 63, 64);
    for (int r = 1; r < NR; ++r) { float w = __shfl(x[r], a & 63, 64); v = (a >> 6) == r ? w : v; }
    return v;
}

// softmax!(prior) (mcts_gpu.jl:417) over one row held lane-wise.  exact: exp_spec + source-order sum.
template <int NR> __device__ __forceinline__ void softmax_row(float (&x)[NR], int A, bool exact) {
    const int lane = lane_id();
    float m = -__builtin_inff();
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; m = (k < A && x[r] > m) ? x[r] : m; }
    m = ufirst(wave_max(m));
    // the sum is taken in source order in both modes (the lane-per-tree kernel sums sequentially anyway);
    // the modes differ only in the exponential: exp_spec (bit-identical to the oracle) or the hardware v_exp_f32.
    float carry = 0.0f; bool st = false;
    for (int r = 0; r < NR; ++r) {
        int k = 64 * r + lane;
        float e = exact ? exp_spec(x[r] - m) : exp2_spec(x[r] - m);
        x[r] = k < A ? e : 0.0f;
        int nr = A - 64 * r; uint64_t full = nr >= 64 ? ~0ull : ((1ull << nr) - 1ull);
        (void)chain64(x[r], full, carry, false, 0.0f, st);
    }
    const float s = carry;
    for (int r = 0; r < NR; ++r) x[r] = x[r] / s;
}

template <int FAM, int NR, int NC, int NRV>
__global__ __launch_bounds__(256) void k_rollout(const TreePar T) {
    using G = Game<FAM, NC>;
    const GamePar& P = T.G;
    const int lane = lane_id();
    const int slot = ufirst((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (slot >= T.L) return;
    const int A = P.A, V = T.V;
    uint8_t* const recs = T.recs + (size_t)slot * V * T.rec_bytes;
    Pos* const states = T.states + (size_t)slot * V;
    uint32_t* const metap = T.meta + (size_t)slot * V;

    MetaRegs<NRV> M;
    uint32_t ncount, leafn = 0;
    if (T.do_reset) {                                   // mcts_gpu.jl:380-387 + re_init :371-372, without touching the arrays
        for (int r = 0; r < NRV; ++r) M.m[r] = (r == 0 && lane == 0) ? M_EXISTS : 0u;
        ncount = 1;
    } else {
        for (int r = 0; r < NRV; ++r) M.m[r] = (64 * r + lane < V) ? metap[64 * r + lane] : 0u;
        ncount = ufirst(T.ncount[slot]);
        leafn = ufirst(T.leaf[slot]);
    }
    uint32_t add_p = 0, add_new = 0;

    // =============================================================================================
    // expand (mcts_gpu.jl:250-302) + backUp (:306-328) of the previous rollout's leaf
    // =============================================================================================
    if (T.do_expand) {
        const int lf = (int)leafn;
        uint32_t ml = M.get(lf);
        const bool term = (ml & M_TERM) != 0;
        float vleaf = 0.0f;
        if (!term) {
            WPos<NC> st = load_pos<NC>(states + lf);
            float pin[NR];
            if (T.inject) {
                for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; pin[r] = k < A ? T.prior_eval[(size_t)slot * A + k] : 0.0f; }
            } else {
                for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; pin[r] = k < A ? T.logits[(size_t)slot * T.LGS + k] : 0.0f; }
                softmax_row<NR>(pin, A, T.exact != 0);
                if (T.capture)
                    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) T.prior_eval[(size_t)slot * A + k] = pin[r]; }
            }
            vleaf = ufirst(T.v_eval[slot]);
            // legal mask, normalize = sum over legal j in source order (:260-268 / :284-290)
            bool legal[NR]; float masked[NR]; uint64_t lmask[NR];
            int nlegal = 0;
            float carry = 0.0f; bool stf = false;
            for (int r = 0; r < NR; ++r) {
                int k = 64 * r + lane;
                legal[r] = k < A && G::canPlay(P, st, k);
                lmask[r] = __ballot(legal[r]);
                nlegal += __popcll(lmask[r]);
                masked[r] = legal[r] ? pin[r] : 0.0f;
                uint64_t nz = __ballot(masked[r] != 0.0f);
                (void)chain64(masked[r], nz, carry, false, 0.0f, stf);
            }
            const float normalize = carry;
            float pr[NR];
            if (lf == 0 && T.training) {                                  // :270-275
                const float Af = (float)nlegal;
                for (int r = 0; r < NR; ++r) pr[r] = legal[r] ? 0.75f * masked[r] / normalize + 0.25f / Af : 0.0f;
            } else {                                                      // :277-279, :292-294
                for (int r = 0; r < NR; ++r) pr[r] = masked[r] / normalize;
            }
            // write the node's record: prior row, q = 0, visits = 0, no children (:297-299 policy := prior is implicit)
            uint8_t* rec = recs + (size_t)lf * T.rec_bytes;
            for (int r = 0; r < NR; ++r) {
                int k = 64 * r + lane;
                if (k < (int)T.A2) {
                    reinterpret_cast<float*>(rec)[k] = k < A ? pr[r] : 0.0f;
                    reinterpret_cast<float*>(rec + T.off_q)[k] = 0.0f;
                    reinterpret_cast<uint16_t*>(rec + T.off_vc)[k] = 0;
                }
            }
            if (lf == 0)
                for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) T.policy_final[(size_t)slot * A + k] = pr[r]; }
            ml |= M_EXPANDED;                                             // :256 expanded = 1 - f
            M.set(lf, ml);
        }
        // ---- backUp: the ancestors' (node, move) pairs come from the meta registers, so all read-modify-writes
        // of one path are issued together, one lane per ancestor (:318-325).
        {
            const int tv2 = (int)((ml >> M_TV_SHIFT) & 3u);
            int cur = lf;
            uint32_t mcur = ml;
            float valf = vleaf;                  // Float32 value path (:316)
            double vald = 0.5 * (double)tv2;     // Float64 value path (:314): (1 + player*r)/2
            while (cur != 0) {
                // collect up to 64 ancestors into lanes
                int mynode = 0, mymove = 0; float myvf = 0.0f; double myvd = 0.0;
                int cnt = 0;
                while (cur != 0 && cnt < 64) {
                    int par = (int)(mcur & 0xffu), mv = (int)((mcur >> 8) & 0xffu);
                    if (lane == cnt) { mynode = par; mymove = mv; myvf = valf; myvd = vald; }
                    valf = 1.0f - valf; vald = 1.0 - vald;               // :324
                    M.orbits(par, M_STALE);                               // :321 uptodate = 0
                    cur = par; mcur = M.get(par);
                    ++cnt;
                }
                if (lane < cnt) {
                    uint8_t* rec = recs + (size_t)mynode * T.rec_bytes;
                    float* qp = reinterpret_cast<float*>(rec + T.off_q) + mymove;
                    uint16_t* vp = reinterpret_cast<uint16_t*>(rec + T.off_vc) + mymove;
                    float q = *qp; uint32_t vc = *vp;
                    float vis = (float)(vc & 0xffu);
                    float nq;
                    if (term) nq = (float)(((double)(vis * q) + (1.0 - myvd)) / (double)(vis + 1.0f));
                    else nq = (vis * q + (1.0f - myvf)) / (vis + 1.0f);   // :319
                    *qp = nq;
                    *vp = (uint16_t)(vc + 1u);                            // :320
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // =============================================================================================
    // kdescendTree! (mcts_gpu.jl:100-199) + decoder (:202-223)
    // =============================================================================================
    if (T.do_select) {
        const uint32_t gid = ufirst(T.game_id[slot]);
        int node = 0, depth = 0;
        uint32_t mn = M.get(0);
        WPos<NC> lst; bool have_state = false;
        while (mn & M_EXPANDED) {
            ++add_p;
            uint8_t* rec = recs + (size_t)node * T.rec_bytes;
            float prior[NR], pol[NR]; uint32_t vc[NR];
            for (int r = 0; r < NR; ++r) {
                int k = 64 * r + lane;
                prior[r] = k < A ? reinterpret_cast<const float*>(rec)[k] : 0.0f;
                vc[r] = k < A ? (uint32_t) reinterpret_cast<const uint16_t*>(rec + T.off_vc)[k] : 0u;
            }
            if (mn & M_STALE) {                                            // :114 uptodate != 1
                float q[NR];
                int vsum = 0, npos = 0;
                float rem[NR]; float carry = 0.0f; bool stf = false;
                for (int r = 0; r < NR; ++r) {
                    int k = 64 * r + lane;
                    q[r] = k < A ? reinterpret_cast<const float*>(rec + T.off_q)[k] : 0.0f;
                    vsum += (int)(vc[r] & 0xffu);
                    npos += __popcll(__ballot(prior[r] > 0.0f));           // :128 A
                    rem[r] = (vc[r] >> 8) == 0 ? prior[r] : 0.0f;          // :122-124 prior_rem over childless actions
                    uint64_t nz = __ballot(rem[r] != 0.0f);
                    (void)chain64(rem[r], nz, carry, false, 0.0f, stf);
                }
                const float nf = 1.0f + (float)ufirst(wave_sum_i(vsum));  // :117,121 (integers: order-free)
                const float Af = (float)npos;
                const float lambda = T.cpuct * __builtin_sqrtf(nf) / (Af + nf); // :132
                const float prior_rem = carry * lambda;                    // :134
                float am = 0.0f;                                           // :133-138
                for (int r = 0; r < NR; ++r) {
                    int k = 64 * r + lane;
                    float lp = lambda * prior[r];
                    float gap = lp > 1e-4f ? lp : 1e-4f;
                    float c = q[r] + gap;
                    am = (k < A && c > am) ? c : am;
                }
                float alpha = ufirst(wave_max(am));
                // children of `node` in creation order = nodes i with parent(i) == node, ascending i (:144-146)
                float ctop[NRV], cq[NRV]; uint64_t cmask[NRV];
                for (int rv = 0; rv < NRV; ++rv) {
                    int i = 64 * rv + lane;
                    uint32_t mi = M.m[rv];
                    bool isc = (mi & M_EXISTS) && i > 0 && i < (int)ncount && (int)(mi & 0xffu) == node;
                    int a = (int)((mi >> 8) & 0xffu);
                    float pa = gather<NR>(prior, a), qa = gather<NR>(q, a);
                    ctop[rv] = isc ? lambda * pa : 0.0f;                   // :147 top
                    cq[rv] = qa;
                    cmask[rv] = __ballot(isc);
                }
                float err = __builtin_inff();
                for (int j = 0; j < 100; ++j) {                            // :141-162
                    float S = prior_rem / alpha;
                    float g = -prior_rem / (alpha * alpha);
                    for (int rv = 0; rv < NRV; ++rv) {
                        float bot = alpha - cq[rv];
                        bool isc = (cmask[rv] >> lane) & 1ull;
                        float t = isc ? ctop[rv] / bot : 0.0f;
                        float u = isc ? -ctop[rv] / (bot * bot) : 0.0f;
                        chain64x2(t, u, cmask[rv], S, g);
                    }
                    float newerr = S - 1.0f;
                    bool brk = newerr < 0.001f || newerr == err;
                    if (ufirst((int)brk)) break;
                    alpha = alpha - newerr / g;
                    err = newerr;
                }
                for (int r = 0; r < NR; ++r) pol[r] = lambda * prior[r] / (alpha - q[r]);   // :165-169
            } else {
                for (int r = 0; r < NR; ++r) pol[r] = prior[r];            // policy == prior since expand (:297-299)
            }
            if (node == 0 && T.last)                                        // copy_pol (:330-339) of the last descent
                for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) T.policy_final[(size_t)slot * A + k] = pol[r]; }

            // ---- sample: first k with running sum >= u, bestmove = last k' <= k with policy > 0 (:172-182)
            const float u = ufirst(uniform_search(T.seed, gid, T.step, T.rollout, (uint32_t)depth));
            int bestmove = -1;
            {
                float carry = 0.0f; bool stopped = false;
                for (int r = 0; r < NR; ++r) {
                    int k = 64 * r + lane;
                    float x = k < A ? pol[r] : 0.0f;
                    uint64_t pos = __ballot(k < A && pol[r] > 0.0f);
                    uint64_t nz = __ballot(x != 0.0f);
                    if (!stopped) {
                        float pre = chain64(x, nz, carry, true, u, stopped);
                        uint64_t ge = __ballot(((nz >> lane) & 1ull) && pre >= u);
                        if (ge) {
                            int kb = __builtin_ctzll(ge);
                            uint64_t cand = pos & (kb >= 63 ? ~0ull : ((2ull << kb) - 1ull));
                            if (cand) bestmove = 64 * r + 63 - __builtin_clzll(cand);
                            stopped = true;
                        } else if (pos) {
                            bestmove = 64 * r + 63 - __builtin_clzll(pos);
                        }
                    }
                }
            }
            if (bestmove < 0) break;                                        // reference would index [-1]; leaf = node
            uint32_t child = pick<NR>(vc, bestmove) >> 8;
            if (child == 0) {                                               // :183-191
                child = ncount; ncount += 1; ++add_new;
                if (lane == (bestmove & 63)) {
                    uint32_t mine = vc[0];
                    for (int r = 1; r < NR; ++r) mine = (bestmove >> 6) == r ? vc[r] : mine;
                    reinterpret_cast<uint16_t*>(rec + T.off_vc)[bestmove] = (uint16_t)(mine | (child << 8));
                }
                WPos<NC> ps = load_pos<NC>(states + node);
                lst = G::play(P, ps, bestmove);
                store_pos<NC>(states + child, lst);
                have_state = true;
                int rr; bool f = G::isOver(P, lst, rr);
                uint32_t mc = (uint32_t)node | ((uint32_t)bestmove << 8) | M_EXISTS | M_EVAL;
                if (f) mc |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
                M.set((int)child, mc);
            } else {
                have_state = false;
            }
            node = (int)child;                                              // :192
            mn = M.get(node);
            ++depth;
        }
        // leaf reached (:195)
        if (!(mn & M_EVAL)) {                                               // root on the first rollout
            lst = load_pos<NC>(states + node); have_state = true;
            int rr; bool f = G::isOver(P, lst, rr);
            mn |= M_EVAL;
            if (f) mn |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
            M.set(node, mn);
        }
        {                                                                   // decoder (:202-223): planes of the leaf
            if (!have_state) lst = load_pos<NC>(states + node);             // (terminal leaves too, as the reference does)
            for (int j0 = 0; j0 < T.INP; j0 += 64) {
                int j = j0 + lane;
                if (j < T.INP) {
                    bool bit = false;
                    if (j < P.VS) bit = bb_get(lst.p, j);
                    else if (j < 2 * P.VS) bit = bb_get(lst.o, j - P.VS);
                    if (T.planes_f32) reinterpret_cast<float*>(T.planes)[(size_t)slot * T.INP + j] = bit ? 1.0f : 0.0f;
                    else reinterpret_cast<uint16_t*>(T.planes)[(size_t)slot * T.INP + j] = bit ? 0x3F80 : 0;
                }
            }
        }
        leafn = (uint32_t)node;
    }

    // write back the slot's bookkeeping
    for (int r = 0; r < NRV; ++r) if (64 * r + lane < V) metap[64 * r + lane] = M.m[r];
    if (lane == 0) {
        T.ncount[slot] = ncount;
        T.leaf[slot] = leafn;
        if (T.do_reset) { T.cnt_p[slot] = add_p; T.cnt_new[slot] = add_new; }
        else { T.cnt_p[slot] += add_p; T.cnt_new[slot] += add_new; }
    }
}

// ---- standalone pieces for the stepwise API and getters -------------------------------------------
// softmax!(prior) as its own kernel (stepwise mode: agz_rollout_eval); same device function as the fused path
template <int NR>
static __global__ __launch_bounds__(256) void k_softmax(const float* logits, int LGS, float* prior_eval, int A, int L, int exact) {
    const int lane = lane_id();
    const int slot = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (slot >= L) return;
    float x[NR];
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; x[r] = k < A ? logits[(size_t)slot * LGS + k] : 0.0f; }
    softmax_row<NR>(x, A, exact != 0);
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) prior_eval[(size_t)slot * A + k] = x[r]; }
}

// decoder_roots (mcts_gpu.jl:225-246) / decoder for getters: fp32 planes of node `which` (0 = root, else leaf[slot])
static __global__ void k_planes(const Pos* states, const uint32_t* leaf, int use_leaf, int V, int VS, int L, float* out) {
    int slot = blockIdx.x;
    if (slot >= L) return;
    const Pos* s = states + (size_t)slot * V + (use_leaf ? leaf[slot] : 0u);
    for (int j = threadIdx.x; j < 2 * VS; j += blockDim.x) {
        int b = j < VS ? j : j - VS;
        const uint64_t* w = j < VS ? s->p : s->o;
        out[(size_t)slot * 2 * VS + j] = ((w[b >> 6] >> (b & 63)) & 1) ? 1.0f : 0.0f;
    }
}

// visits[:,1,:] and q[:,1,:] of the root as fp32 [L][A]
static __global__ void k_root_stats(const uint8_t* recs, const uint32_t* meta, int V, uint32_t rec_bytes, uint32_t off_q, uint32_t off_vc,
                             int A, int L, float* visits, float* q) {
    int slot = blockIdx.x;
    if (slot >= L) return;
    const uint8_t* rec = recs + (size_t)slot * V * rec_bytes;
    bool expanded = (meta[(size_t)slot * V] & M_EXPANDED) != 0;
    for (int k = threadIdx.x; k < A; k += blockDim.x) {
        float vv = 0.0f, qq = 0.0f;
        if (expanded) {
            vv = (float)(reinterpret_cast<const uint16_t*>(rec + off_vc)[k] & 0xffu);
            qq = reinterpret_cast<const float*>(rec + off_q)[k];
        }
        if (visits) visits[(size_t)slot * A + k] = vv;
        if (q) q[(size_t)slot * A + k] = qq;
    }
}

}  // namespace agz
