// agz_tree_eager.hpp — the PUCT tree kernel: policy rows AND the next sampled action are computed eagerly, by the backup.
//
// The reference recomputes a node's policy row lazily, at the next visit after a backup has passed through it
// (mcts_gpu.jl:114-169: uptodate is cleared by backUp :321 and never set again), inside the descent, and samples from it with
// a fresh uniform (:172-182).  A descent is a chain of dependent node visits whose length differs from game to game: with 8
// games per wavefront the wave runs to the deepest of its games (8.4 rounds for a mean depth of 4.4 on Gobang 9x9), so nearly
// half of the lanes idle through the expensive part — lambda, alpha0, Newton, 81 IEEE divisions, the ordered prefix.  But the
// row a visit will find depends only on the node's own state after the LAST backup through it, every node a backup passes
// through is known when the backup starts, and the uniform of the visit is keyed by that backup: U(seed; game, step, rollout of
// the backup, depth of the node) — one independent uniform per node visit, as the reference draws them (agz_device.hpp
// uniform_search).  So:
//   * backup + recompute: every node on the path of the previous rollout is one independent WORK ITEM — update q / visits of the
//     edge taken (:319-320), then lambda, alpha0, Newton, the policy row, its source-order running sums, and the action the
//     NEXT visit of the node will sample (:172-182: the number of running sums below u) with the child it leads to.  The items of
//     all 8 games of the wave are dealt to the 8 lane-groups 8 at a time: ceil(sum of depths / 8) balanced rounds instead of
//     max(depth) divergent ones.  Nothing of the row is stored: one 32-bit word per node (action | child << 8 | valid).
//   * the descent follows those words: one 4-byte load per level, no arithmetic.
// Same arithmetic, same order of every fp32 operation as the reference restated by the oracle; only the time at which a row is
// computed changes.  Layout (A2 = 8 KPL >= A):
//   rec[L][V]  [aux 16 bytes, below][prior f32 x A2][rank u8 x A2][cid u8 x A2] by action: creation rank + 1 and node id of the child under each action
//              (0 = none); [edge {q, prior} f32x2 x VL][visits u8 x VL] by creation rank (VL = V rounded up to 16): the running
//              mean and the visit count of the edge to the child — only as many entries as the node has children are ever
//              read, and an expansion writes the per-action rows only (a fresh node has no edge)
//   aux        (the head of the record: the descent's look at a node brings in the cache line its item reads first)
//              {prior_rem before lambda (:120-124), next word (action | child << 8 | valid << 16 | rank + 1 of the child << 17),
//              npos | nvis << 8 | nch << 16, -}
//   wl[block][8 V] work list of the wave (in LDS inside the whole-search kernel): one word per expanded node passed below which
//   the descent went on; sp[slot]: the last expanded node of the path (the parent of the leaf) — these 8 items are processed
//   together in the first round because they alone may have a new child to register (rank, cid, re-summed prior_rem).
#pragma once
#include <type_traits>
#include "agz_wave.hpp"
#include "agz_divpair.hpp"
#include "agz_fastdiv.hpp"

#ifndef AGZ_LATE_PRIO
#define AGZ_LATE_PRIO 0     // n > 0: a wave whose rollout has more than n item rounds runs its item loop at priority 1 (A/B)
#endif
#ifndef AGZ_FAST_COUNT
#define AGZ_FAST_COUNT 1    // 1: the sampled action of a work item from block-local running sums + a rounding margin (sample_count_fast); 0: always the source-order chain (A/B)
#endif

#define AGZ_LDSP __attribute__((address_space(3)))
#define AGZ_GLBP __attribute__((address_space(1)))

namespace agz {

enum : uint32_t { SP_VALID = 1u << 24, SP_CREATED = 1u << 25, NX_VALID = 1u << 16, AUX_SLOW = 1u << 24 };

// correctly rounded sqrt of a float in [1, 2^24] (here: 1 + the visit count of a node): v_sqrt_f32 is within 1 ulp, and the two
// residual tests are the compiler's own correction steps — without its scaling for tiny arguments and its inf / 0 test (8 instead
// of 20 instructions, same result)
__device__ __forceinline__ float sqrt_count(const float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;
    s = rp > 0.0f ? sp : s;
    return s;
}
// bits [bit0, bit0 + 32) of a bitboard (zeros beyond its end), bit0 per lane: two 32-bit words picked by the lane, one v_alignbit
template <int NC> __device__ __forceinline__ uint32_t bb_field32(const BB<NC>& b, const int bit0) {
    uint32_t w[2 * NC + 2];
#pragma unroll
    for (int i = 0; i < NC; ++i) { w[2 * i] = (uint32_t)b.c[i]; w[2 * i + 1] = (uint32_t)(b.c[i] >> 32); }
    w[2 * NC] = 0u; w[2 * NC + 1] = 0u;
    const int i = bit0 >> 5;
    uint32_t lo = w[0], hi = w[1];
#pragma unroll
    for (int k = 1; k <= 2 * NC; ++k) { lo = i == k ? w[k] : lo; hi = i == k ? w[k + 1] : hi; }
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(bit0 & 31));
}
// canPlay of the lane's block of KPL <= 24 actions k0 .. k0 + nval - 1 as a bit mask (mcts_gpu.jl:262, :286 call canPlay per action).
// Games whose action k is cell k (Gobang.jl:25-27: the cell is empty; Reversi8x8.jl:84-90: bit k of the cached legal set, the pass
// action = no legal move) read the block as one bit field; the others (Connect4's top cells, Hex's padded board) ask per action.
template <int FAM, int NC, int KPL>
__device__ __forceinline__ uint32_t legal_block(const GamePar& P, const WPos<NC>& st, const int k0, const int nval) {
    uint32_t m = 0u;
    if constexpr (FAM == F_LINE) m = ~bb_field32<NC>(bb_or(st.p, st.o), k0);
    else if constexpr (FAM == F_REV) {
        m = bb_field32<NC>(st.lg, k0);
        const int pj = P.pass_action - k0;
        if (pj >= 0 && pj < KPL && !bb_any(st.lg)) m |= 1u << pj;
    } else {
#pragma unroll
        for (int j = 0; j < KPL; ++j) m |= Game<FAM, NC>::canPlay(P, st, k0 + j) ? 1u << j : 0u;
    }
    return m & ((1u << nval) - 1u);
}

// bytes of a node record [aux uint4][prior f32 x A2][rank u8 x A2][cid u8 x A2][edge f32x2 x VL][visits u8 x VL] (A2 a multiple of 8;
// a node has at most one child per action and per rollout: VL = min(V, A2) rounded up; the record a multiple of 16 bytes)
__host__ __device__ constexpr int eager_vl(int V, int A2) { return ((V + 15) & ~15) < ((A2 + 7) & ~7) ? ((V + 15) & ~15) : ((A2 + 7) & ~7); }
__host__ __device__ constexpr int eager_rec_bytes(int A2, int V) { return (16 + 6 * A2 + 9 * eager_vl(V, A2) + 15) & ~15; }

struct EagerLds { int tab, tstride, val, utab, total; };
// NG = games (lane-groups) per wave = 64 / lanes per group; A2 = entries of a node row (lanes per tree x entries per lane): a node has at most
// one child per row entry and per rollout, so its edge table holds eager_vl(V, A2) entries — 16 for Connect4 (was V = 64), 96 for the 9x9
// boards at V = 128 (was 128: what lets the 128-game workgroups of the wide trunks keep their next-word tables in LDS at V = 128)
__host__ __device__ inline EagerLds eager_lds_layout(int V, int NG = 8, int A2 = 1 << 20) {
    EagerLds o;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    // per lane-group: the edges {q, prior} of the item's node in creation order, a zero pair in front of them (what an action
    // without a child reads)
    o.tab = 16;
    o.tstride = 16 + up16(eager_vl(V, A2) * 8);
    o.val = NG * o.tstride;                                      // per game: {value_1, value_2, flags, -}
    o.utab = o.val + NG * 16;                                    // per game: 32 uniforms (depths 0..31)
    o.total = o.utab + NG * 128;
    return o;
}

// what a game carries from one rollout to the next.  The stand-alone kernel keeps it in global memory (T.ncount, T.leaf, T.sp,
// T.cnt_*); the whole-search kernel (agz_search_small.hpp) keeps it in registers across its rollout loop.
struct EagerCarry { uint32_t ncount, leafn, spw, add_p, add_new, root_exp, leaf_meta; };   // leaf_meta: the leaf's meta word (whole-search kernels)

// rows of one work item, loaded one round ahead of their use
template <int KPL> struct ChildWords { uint32_t w[KPL / 4]; };       // child-id bytes of a lane's block, passed by value (registers)
template <int KPL> struct ItemRows {
    float p[KPL]; uint32_t rk[KPL / 4], cd[KPL / 4];              // priors; rank + 1 / child id bytes of the lane's KPL actions
    uint32_t ax_x, ax_z;                                          // aux: prior_rem bits, npos | nvis << 8 | nch << 16
    float pm, qm; uint32_t vism;                                  // the edge taken: prior of its action (a new edge), q, visits
    float2 e0, e1, e2, e3;                                        // edges sub, G + sub .. of the node's list (e1.. only for a root: the node with many children)
    uint32_t ent; int gi; bool valid;
};

// LEAN: the caller is a whole-search kernel: V a multiple of 4 (<= 128 in the engine's dispatch), bf16 network mode, no inject / capture, the carry
// lives in registers (C) and the wave's work list in LDS (wl_lds, wcount); otherwise both live in global memory.
// PFM = 2: the rows of the next work item are requested while the running sums of the current one are computed, and those of
// the first item before the leaf's rows are written (35 more live registers: only where the register budget has room for them
// — in the 128-register build the spills cost more than the latency saved, measured 12.5 vs 7.8 ms per 32768-game search).
// PFM = 1: the next item's record is only TOUCHED a round ahead (each of the 8 lanes of the group reads one dword of one of the
// record's 8 cache lines into a scratch register), so that the real loads find it in L2 instead of paying an HBM miss.
// wl_cap_lds: entries of the work list that fit the LDS region (the rest, rare, goes to the global list).
// io_blk (LIO builds): this wave's block of the LDS hand-over window of k_search_small — the network body left the logits
// of game g in row g (io_lgs floats, the value in column A) and takes the leaf's planes from row g (io_prowb bytes, zero padded to
// whole k-rows): one round trip through L2 less in each direction on the rollout's chain.  The global arrays are not written then.
// ROLE (whole-search kernel with idle network-only waves: the 16-game workgroups): the expansion of a rollout's leaf and the
// backup work items of its path are independent of each other (the first needs the logits, the second the value), so two waves
// run them side by side — ROLE_EXPAND: the game's tree wave expands (and, in a second call after a workgroup barrier, descends);
// ROLE_ITEMS: a helper wave processes the work items of that tree wave's games from the carry the tree wave published in `xch`
// (per game {leaf, special item, leaf meta word}, then the length of the work list).  ROLE_ALL: one wave does everything.
enum { ROLE_ALL = 0, ROLE_EXPAND = 1, ROLE_ITEMS = 2 };

// The tree parameters are READ FROM THE KERNEL-ARGUMENT SEGMENT at the head of every phase (scalar loads through a pointer made
// opaque there) instead of living in scalar registers from the kernel's entry on: some 60 words of parameters next to the lane
// masks and wave-uniform addresses of the tree step are more than the 102 scalar registers of a wave, and every spilled one costs
// v_writelane / v_readlane — 4-cycle VECTOR instructions (13 % of the vector instructions of the 9x9 kernel before this).
// Every kernel that runs rollout_eager_body has its TreePar at the START of its argument segment (static_assert at each).
typedef const TreePar __attribute__((address_space(4)))* TreeKArg;
__device__ __forceinline__ const TreePar& tree_par() {
    TreeKArg p = (TreeKArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *(const TreePar*)p;
}

// KPR_ (legal-compacted rows, Gobang / Hex in the whole-search kernels): a stone is never removed, so every node of a search has its legal
// actions among the ROOT's legal actions; with KPR_ set the node rows are indexed by the root's legal RANK (the r-th legal action
// of the root, in action order) and hold KPR_ < KPL entries per lane — from ply 17 of a 9x9 game 64 ranks cover every tree.  The
// ordered sums of the reference run over the actions in action order and an illegal action contributes an exact zero, so the
// sums over the ranks are the same bits; the softmax denominator still runs over all A logits (action space, KPL per lane) and the
// expansion compacts the masked numerators into rank order through LDS.  policy_final is written in rank order too (the engine
// spreads it back over the actions after the search: k_spread_policy).
// G_ = lanes per game tree (8, or 4 / 2 in the whole-search kernels): a wave walks NG = 64 / G_ trees, lane sub of a group owns the KPL
// actions sub KPL .. of every row.  Fewer lanes per tree = more trees per wave: the per-round fixed work of the item loop (Newton, the
// turns of the ordered sums, reductions, sampling, the backup) is shared by twice / four times the items, and a game with few actions
// (Connect4: 7) does not leave six of eight lanes without an action.
// nxw != nullptr (round 6; the whole-search kernels, wherever the workgroup's LDS has 2 V bytes per game left): the NEXT WORDS of the wave's trees
// live in LDS as well — nxw[game in wave][node], 16 bits: child id (7 bits) | valid << 7 | (child ? creation rank + 1 of its edge : the action) << 8
// — and the descent follows THOSE: one LDS read per level (~100 cycles) instead of one dependent global load per level (an L2 hit at best,
// 8.4 levels per wave and rollout: 8 % of a wave's time on the headline shape; measured +6 % rollouts/s).  Every word a search reads was
// written by a wave of the same workgroup during the same search (the root's at its expansion, a child's at its creation; the helper waves
// of the 16-game workgroups write the table of their tree wave, a workgroup barrier in between), so the table needs no clearing and
// nothing outside a search depends on it; the global next word is still written with the aux words (same 16-byte store).
template <int FAM, int NC, int KPL, bool LEAN, int PFM, bool LIO = false, int ROLE = ROLE_ALL, int KPR_ = 0, int G_ = 8>
__device__ __forceinline__ void rollout_eager_body(const StepFlags SF, uint8_t* const lds, const int bidx,
                                                   EagerCarry& C, uint32_t* const wl_lds, const uint32_t wl_cap_lds, uint32_t& wcount,
                                                   uint8_t* const io_blk = nullptr, const int io_prowb = 0, const int io_lgs = 0,
                                                   uint32_t* const xch = nullptr, const uint32_t dead_mask = 0u, const uint32_t rec_stride = 0u,
                                                   uint16_t* const nxw = nullptr) {
    using GM = Game<FAM, NC>;
    constexpr bool REV = FAM == F_REV;
    constexpr int G = G_, NG = 64 / G_;
    static_assert(G == 8 || G == 4 || G == 2, "lanes per game tree");
    static_assert(G == 8 || (LEAN && PFM == 2), "narrow lane-groups: whole-search kernels with register prefetch");
    static_assert(KPL % 4 == 0 && KPL <= 24, "block of actions per lane: a multiple of 4, one 32-bit legal mask");
    constexpr int KPR = KPR_ ? KPR_ : KPL;                        // entries per lane of the node ROWS
    constexpr bool CMP = KPR != KPL;                              // rows by the root's legal rank
    static_assert(!CMP || (LEAN && KPR % 4 == 0 && KPR < KPL && (FAM == F_LINE || FAM == F_HEX)), "legal-compacted rows: lean builds of the stone-placing games");
    const bool NXL = LEAN && nxw != nullptr;                      // (wave-uniform)
    // the 32-bit next word (as stored in a record's aux) -> the 16-bit form of the LDS table
    auto nx16 = [](const uint32_t nx) -> uint16_t {
        const uint32_t child = (nx >> 8) & 0xffu;
        return (nx & NX_VALID) ? (uint16_t)(child | 0x80u | ((child ? (nx >> 17) & 0xffu : nx & 0xffu) << 8)) : (uint16_t)0u;
    };
    const TreePar& T = tree_par();
    const GamePar& P = T.G;
    int lane_ = lane_id();
    asm volatile("" : "+v"(lane_));                              // opaque per call (see rollout_reg_body)   // PHASE setup
    const int lane = lane_ & 63, g = lane / G, sub = lane % G;
    const int GPW = T.gpw;
    const int slot_base = T.slot0 + __builtin_amdgcn_readfirstlane(bidx) * GPW;   // (wave-uniform: the wave's arrays get scalar bases)
    const int slot = slot_base + g;
    // (dead_mask, wave-uniform: bit g set = the wave's slot g holds no game — the persistent self-play kernels, whose slots are never compacted)
    const bool live = g < GPW && slot < T.L && !((dead_mask >> g) & 1u);
    const bool lead = sub == 0;
    // the row geometry follows from KPL alone (the engine lays the records out with the same formulas): compile-time offsets,
    // so that every load / store of a row is one base address + an immediate
    constexpr int A2 = G * KPR;
    constexpr int OFF_P = 16, OFF_RK = 16 + 4 * A2, OFF_CID = 16 + 5 * A2, OFF_EL = 16 + 6 * A2;   // aux, per-action rows, then the edge list by rank ...
    const int A = P.A, V = T.V;
    // (rec_stride != 0: the records keep the stride the engine allocated them with although this build's rows are narrower — the persistent
    //  self-play kernel runs workgroups with rows by action and workgroups with rows by legal rank side by side on one record buffer)
    const uint32_t OFF_VIS = (uint32_t)(OFF_EL + 8 * eager_vl(V, A2)), ROWS = rec_stride ? rec_stride : (uint32_t)eager_rec_bytes(A2, V);   // ... and the visit bytes by rank
    const EagerLds LO = eager_lds_layout(V, NG, A2);
    float2* const tab = reinterpret_cast<float2*>(lds + (size_t)g * LO.tstride + LO.tab);   // tab[-1] = {0, 0}
    float4* const valtab = reinterpret_cast<float4*>(lds + LO.val);
    float* const utab = reinterpret_cast<float*>(lds + LO.utab);
    const int sl = live ? slot : 0;
    const int k0 = sub * KPL;
    const int nlanes = (A + KPL - 1) / KPL;                            // lanes of a group whose block holds real actions (wave-uniform)
    const int nval = A - k0 < 0 ? 0 : (A - k0 > KPL ? KPL : A - k0);   // real actions in this lane's block (padding sits at the end of the last blocks)
    // the same three for the lane's block of a node ROW (compacted rows: every rank slot takes part — slots past the root's legal
    // count hold zeros like illegal actions do)
    const int r0 = sub * KPR;
    const int nlr = CMP ? G : nlanes;
    const int nvr = CMP ? KPR : nval;
    const bool inject = !LEAN && T.inject, capture = !LEAN && T.capture;
    const bool exact = !LEAN && T.exact, planes_f32 = !LEAN && T.planes_f32;
    // FD: quotients by agz_fastdiv.hpp (same bits as '/', half the instructions) wherever the operands are inside its range by
    // construction — bf16 mode with cpuct in [2^-10, 2^10] (T.fastdiv), no injected priors:
    // the logits of the leaf spread over at most 55 (checked at the expansion; a node that fails the check is flagged AUX_SLOW and
    // keeps '/' for good):
    //   softmax     x / s          x = exp2_spec(.) is 0 (padding) or in [2^-80, 1];  s in [1, 8 KPL]
    //   normalize   x' / norm      x' in {0} U [2^-87, 1] (x / s, times 0.75 at the root);  norm = sum of those, checked >= 2^-100
    //   policy      lambda p / (alpha - q):  p >= 2^-87 or 0, lambda in [2^-19, 2^10];  alpha - q >= 1e-4 for every action (alpha
    //               starts at max(q + max(lambda p, 1e-4)) and Newton only moves it up), <= 2^4
    //   Newton      top / bot, -top / bot^2: same bounds;  newerr / g: newerr in [1e-3, 2^25] and then |g| in [2^-10, 2^39]
    // The backup's own quotient (vis q + 1 - v) / (vis + 1) keeps '/': a value head output may be arbitrarily small.
    const bool FD = T.fastdiv && !inject && !exact;
    const int wl_block = T.slot0 / NG + __builtin_amdgcn_readfirstlane(bidx);
    uint32_t* const wl_g = T.wl + (size_t)wl_block * (size_t)T.wl_cap * (size_t)(NG / 8);   // this wave's work list (global form; wl_cap entries per 8 games)
    // the wave's 8 games lie next to each other in every per-node array: ONE wave-uniform base per array (scalar registers) and
    // 32-bit offsets (game-in-wave, node) from it — every load / store of the item loop is base + 32-bit offset + immediate,
    // without 64-bit vector address arithmetic
    // (a wave of the ragged last workgroup may own no game at all: its idle loads then go to the launch's first game)
    const int mem_base = slot_base < T.L ? slot_base : T.slot0;
    uint8_t* const wrecs = T.recs + (size_t)mem_base * (size_t)V * ROWS;
    auto auxp = [&](const uint32_t nd_) -> uint4* { return reinterpret_cast<uint4*>(wrecs + __umul24(nd_, ROWS)); };   // a node's aux words
    uint32_t* const wmeta = T.meta + (size_t)mem_base * (size_t)V;
    Pos* const wstates = T.states + (size_t)mem_base * (size_t)V;
    const int gl = live ? g : 0;                                                  // this group's game inside the wave (0 when it has none)
    const uint32_t gnode0 = (uint32_t)(gl * V);                                   // index of its root in the wave's node arrays
    uint32_t* const gmeta = wmeta + gnode0;
#ifdef AGZ_STAMPS
    unsigned long long* const stamp_lds = reinterpret_cast<unsigned long long*>(lds + LO.total);
    if (lane < 17) stamp_lds[lane] = lane == 16 ? __builtin_amdgcn_s_memtime() : 0ull;
    AGZ_WSYNC();
#endif
    if (lead) tab[-1] = make_float2(0.0f, 0.0f);
    if (SF.do_reset) {
        C.ncount = 1; C.leafn = 0; C.spw = 0; C.add_p = 0; C.add_new = 0; C.root_exp = 0; C.leaf_meta = M_EXISTS;
        wcount = 0;
        if (live && lead) gmeta[0] = M_EXISTS;
    } else if constexpr (!LEAN) {
        C.ncount = 1; C.leafn = 0; C.spw = 0; C.root_exp = 0;
        if (live) { C.ncount = T.ncount[slot]; C.leafn = T.leaf[slot]; C.spw = T.sp[slot]; C.root_exp = (gmeta[0] & M_EXPANDED) ? 1u : 0u; }
        C.add_p = 0; C.add_new = 0;
        wcount = ufirst(T.wl_n[wl_block]);
    }
    STAMPW(0);

    // The action the next visit of a row samples (:172-182), given the row pol[] (block sub of the group), the running sum st before
    // the block, the visit's uniform u and the child bytes of the block: the number of running sums below u — the row of sums is
    // nondecreasing — or, when the whole row sums below u, the last positive action.  Returns the next word.
    // next word: action | child << 8 | NX_VALID | creation rank + 1 of the child << 17 (0: no child yet)
    // cnt = the lane's count of running sums below u (the caller's); the rest of :172-182: the action, its child and the child's rank
    auto sample_finish = [&](const float (&pol)[KPR], int cnt, const ChildWords<KPR> cdw, const ChildWords<KPR> rkw,
                             const int fix_move, const uint32_t fix_child) -> uint32_t {
        cnt = cnt < nvr ? cnt : nvr;                                           // (padded actions never count)
        int bestmove = grp_sum<G>(cnt);
        // (compacted rows: the slots past the root's legal count repeat the row's total, so a row that sums below u counts all G KPR)
        const int AE = CMP ? G * KPR : A;
        if (__builtin_expect(wballot(bestmove >= AE) != 0, 0)) {              // the row sums below u: the last positive action wins (:175-181)
            int lastpos = -1;
#pragma unroll
            for (int j = 0; j < KPR; ++j) lastpos = pol[j] > 0.0f ? r0 + j : lastpos;
            lastpos = grp_max_i<G>(lastpos);
            bestmove = bestmove >= AE ? lastpos : bestmove;
        }
        if (bestmove < 0) return 0u;                                           // (the reference would index [-1]: the visit ends here)
        const uint32_t idx = (uint32_t)(bestmove - r0);
        uint32_t wsel = cdw.w[0], rsel = rkw.w[0];
#pragma unroll
        for (int j = 1; j < KPR / 4; ++j) { wsel = (idx >> 2) == (uint32_t)j ? cdw.w[j] : wsel; rsel = (idx >> 2) == (uint32_t)j ? rkw.w[j] : rsel; }
        // (child id in bits 0..7, rank in bits 8..15 of one sum: both are 0 in every lane but the one that owns the action)
        const uint32_t both = idx < (uint32_t)KPR ? __builtin_amdgcn_ubfe(wsel, (idx & 3u) * 8u, 8u) | (__builtin_amdgcn_ubfe(rsel, (idx & 3u) * 8u, 8u) << 8) : 0u;
        const uint32_t cr = (uint32_t)grp_sum<G>((int)both);
        uint32_t child = cr & 0xffu;
        const uint32_t rank = cr >> 8;                                         // (the rank bytes of this item's own new edge are already in rkw)
        if (bestmove == fix_move) child = fix_child;                           // the child registered by this very item
        return (uint32_t)bestmove | (child << 8) | NX_VALID | (rank << 17);
    };
    auto sample_next = [&](const float (&pol)[KPR], const float st, const float u, const ChildWords<KPR> cdw, const ChildWords<KPR> rkw,
                           const int fix_move, const uint32_t fix_child) -> uint32_t {
        float c = st; int cnt = 0;
#pragma unroll
        for (int j = 0; j < KPR; ++j) { c += pol[j]; cnt += c < u ? 1 : 0; }
        return sample_finish(pol, cnt, cdw, rkw, fix_move, fix_child);
    };
    // The same count WITHOUT the source-order chain over the group's lanes (round 6).  What the visit needs of the running sums S_k (:172-181) is
    // how many of them lie below u; the sums themselves are used for nothing else.  Every lane sums its own block from 0, an exclusive prefix over
    // the group's lanes (three DPP steps) gives the block's offset, and d_k = (offset - u) + local sum is S_k - u up to rounding: for
    // non-negative terms the source-order sum and ANY other order of the same terms differ from the exact sum by at most (n - 1) eps each
    // (relative to the sum, eps = 2^-24, n = G KPR terms) — together with the two roundings of d_k less than (n + KPR + 4) eps max(total, 1).  So
    // whenever |d_k| exceeds TWICE that for every k, the sign of d_k IS the outcome of the reference's comparison `S_k < u` and the count is exact;
    // otherwise (some running sum within ~1.4e-5 of u on a 9x9 board: ~0.2 % of the items) the caller takes the source-order chain.  7 turns of KPR additions
    // (all 64 lanes execute every turn) become KPR + 9 instructions.  Returns false (wave-uniform) when some item of the wave must take the chain.
    auto sample_count_fast = [&](const float (&pol)[KPR], const float u, int& cnt_out) -> bool {
        float inc = pol[0];                                                     // the block's total (the same additions again below: no array of sums is kept live)
#pragma unroll
        for (int j = 1; j < KPR; ++j) inc += pol[j];
        // inclusive prefix of the block totals over the lanes of the group, then the exclusive one (the value of the lane before)
        // (the lane tests are made HERE, on a copy of `sub` the compiler cannot see through: hoisted out of the item loop they are three 64-bit
        //  lane masks held in scalar registers for the whole loop — and scalar registers are what the loop is short of)
        int sb = sub;
        asm volatile("" : "+v"(sb));
        if constexpr (G >= 2) { const float t = __int_as_float(dpp_mov<0x111, 0xF>(0, __float_as_int(inc))); inc += sb >= 1 ? t : 0.0f; }
        if constexpr (G >= 4) { const float t = __int_as_float(dpp_mov<0x112, 0xF>(0, __float_as_int(inc))); inc += sb >= 2 ? t : 0.0f; }
        if constexpr (G >= 8) { const float t = __int_as_float(dpp_mov<0x114, 0xA>(0, __float_as_int(inc))); inc += t; }   // (bank mask: the lanes 4 .. 7 of a group)
        const float prev = __int_as_float(dpp_mov<0x111, 0xF>(0, __float_as_int(inc)));
        const float off = sb >= 1 ? prev : 0.0f;
        const float total = grp_bcast_last<G>(inc);
        const float base = off - u;
        int cnt = 0; float mn = __builtin_inff(), c = 0.0f;
#pragma unroll
        for (int j = 0; j < KPR; ++j) {
            c += pol[j];
            const float d = base + c;
            cnt += d < 0.0f ? 1 : 0;
            mn = __builtin_fminf(mn, __builtin_fabsf(d));
        }
        cnt_out = cnt;
        // twice the bound: (G KPR - 1) eps of the source-order sum + (KPR + 2) eps of the block-local sum and the prefix + 2 eps of d_k's own roundings
        constexpr float MARGIN = (float)(2 * (G * KPR + KPR + 4)) * 5.9604644775390625e-8f;
        const float margin = MARGIN * __builtin_fmaxf(total, 1.0f);
        return wballot(!(mn > margin)) == 0ull;                               // (a NaN anywhere takes the chain too)
    };

    // one work item's rows -> registers (zeros for a lane-group without an item).  entry: node | move << 8 | depth << 16 | game << 24
    // part: 0 = everything; 1 = what the round needs FIRST (the entry, aux words, the edge taken, the rank bytes, the head of the edge
    // list: the backup and the edge table), 2 = the rest (priors and child bytes, used after Newton) of the item whose first part R holds.
    // PFM = 3 prefetches part 1 a round ahead into registers (19 instead of 34) and requests part 2 at the head of the round, where the
    // backup and Newton cover its latency.
    // the work list's two homes by their address spaces (a select of two generic pointers becomes ONE flat access: the LDS entries of the list
    // then travel through the texture path)
    auto wl_lds_ld = [&](const uint32_t i) -> uint32_t { return *((const AGZ_LDSP uint32_t*)wl_lds + i); };
    auto wl_g_ld = [&](const uint32_t i) -> uint32_t { return *((const AGZ_GLBP uint32_t*)wl_g + i); };
    auto item_fetch = [&](ItemRows<KPR>& R, const int r, const uint32_t nwl, const int part = 0) {
        if (part != 2) {
        R.ent = 0u; R.gi = g; R.valid = false;
        // round 0: the lane-groups that own a game take its special item, the others (sparse waves: g >= GPW) already take list entries
        if (r == 0 && g < GPW) { R.ent = C.spw; R.valid = live && (C.spw & SP_VALID); }
        else {
            const uint32_t idx = r == 0 ? (uint32_t)(g - GPW) : (uint32_t)(NG - GPW) + (uint32_t)NG * (uint32_t)(r - 1) + (uint32_t)g;
            if (idx < nwl) {
                if (LEAN && idx < wl_cap_lds) R.ent = wl_lds_ld(idx); else R.ent = wl_g_ld(idx);
                R.valid = true; R.gi = (int)(R.ent >> 24) & (NG - 1);
            }
        }
        }
        // entry: node | mr << 8 | ...: mr = the ACTION of a new edge (special item with SP_CREATED), else the creation rank + 1 of
        // the edge taken
        const uint32_t node = R.ent & 0xffu, mr = (R.ent >> 8) & 0xffu;
        const bool crt = r == 0 && g < GPW && (R.ent & SP_CREATED);
        // a lane-group without an item reads the root record of its own game (ent == 0: node 0) — finite numbers, the engine
        // clears the records once at creation — with prior_rem = 0 and no children: its Newton loop ends in the first iteration and
        // nothing it computes is stored.  The loads are unconditional (no branch, no zero fill of 30 registers per round).
        const uint32_t nd = (uint32_t)((R.valid ? R.gi : gl) * V) + node;
        const uint8_t* const rec = wrecs + __umul24(nd, ROWS);
        if (part == 2) {
#pragma unroll
            for (int j = 0; j < KPR; j += 4) {
                const float4 a = *reinterpret_cast<const float4*>(rec + OFF_P + (uint32_t)(r0 + j) * 4u);
                R.p[j] = a.x; R.p[j + 1] = a.y; R.p[j + 2] = a.z; R.p[j + 3] = a.w;
                R.cd[j / 4] = *reinterpret_cast<const uint32_t*>(rec + OFF_CID + (uint32_t)(r0 + j));
            }
        }
        if (part != 2) {
            const uint4 ax = *reinterpret_cast<const uint4*>(rec);
            R.ax_x = ax.x; R.ax_z = ax.z;                             // (raw: nothing here may wait for a load — the item body masks them)
#pragma unroll
            for (int j = 0; j < KPR; j += 4) {
                if (part == 0) {
                    const float4 a = *reinterpret_cast<const float4*>(rec + OFF_P + (uint32_t)(r0 + j) * 4u);
                    R.p[j] = a.x; R.p[j + 1] = a.y; R.p[j + 2] = a.z; R.p[j + 3] = a.w;
                }
                R.rk[j / 4] = *reinterpret_cast<const uint32_t*>(rec + OFF_RK + (uint32_t)(r0 + j));
                if (part == 0) R.cd[j / 4] = *reinterpret_cast<const uint32_t*>(rec + OFF_CID + (uint32_t)(r0 + j));
            }
            // the edge taken: an existing one is entry mr - 1 of the list; a new one has q = 0, no visit, and the prior of its action
            const uint32_t er = (crt || mr == 0u) ? 0u : mr - 1u;
            const float2 em = *reinterpret_cast<const float2*>(rec + OFF_EL + er * 8u);
            const uint32_t vm = rec[OFF_VIS + er];
            R.pm = reinterpret_cast<const float*>(rec + OFF_P)[crt ? mr : 0u];
            R.qm = em.x; R.vism = vm;                                 // (raw: a new edge ignores them)
            // the node's edges for the Newton sums and the per-action q: entry sub of every item; a root (node 0: the node that collects
            // children) also entries G + sub, 2 G + sub, 3 G + sub — what lies beyond is fetched when the item is processed
            R.e0 = *reinterpret_cast<const float2*>(rec + OFF_EL + (uint32_t)sub * 8u);
            R.e1 = R.e2 = R.e3 = make_float2(0.0f, 0.0f);
            if (node == 0u) {
                const int VLc = eager_vl(V, A2);                      // (entries of the list: reads stay inside the record)
                if (VLc > G) R.e1 = *reinterpret_cast<const float2*>(rec + OFF_EL + (uint32_t)(G + sub) * 8u);
                if (VLc > 2 * G) R.e2 = *reinterpret_cast<const float2*>(rec + OFF_EL + (uint32_t)(2 * G + sub) * 8u);
                if (VLc > 3 * G) R.e3 = *reinterpret_cast<const float2*>(rec + OFF_EL + (uint32_t)(3 * G + sub) * 8u);
            }
        }
    };

    constexpr bool PF = PFM == 2, PF3 = PFM == 3;
    // PFM = 4 (round 6): the first round's rows are touched at the head of the expansion and fetched at the head of the round (as PFM = 1); the rows of
    // every later round are requested into registers behind the policy row of the round before (as PFM = 2)
    constexpr bool PF4 = PFM == 4;
    // touch the record (and the aux word) of the item of round r: one dword per cache line
    auto item_touch = [&](const int r, const uint32_t nwl) -> uint32_t {
        uint32_t ent = 0u; int gi = g; bool valid = false;
        if (r == 0 && g < GPW) { ent = C.spw; valid = live && (C.spw & SP_VALID); }
        else {
            const uint32_t idx = r == 0 ? (uint32_t)(g - GPW) : (uint32_t)(NG - GPW) + (uint32_t)NG * (uint32_t)(r - 1) + (uint32_t)g;
            if (idx < nwl) {
                if (LEAN && idx < wl_cap_lds) ent = wl_lds_ld(idx); else ent = wl_g_ld(idx);
                valid = true; gi = (int)(ent >> 24) & (NG - 1);
            }
        }
        uint32_t v = 0u;
        if (valid) {
            const uint32_t nd = (uint32_t)(gi * V) + (ent & 0xffu);
            // aux + the per-action rows and the head of the edge list (16 + 6 A2 + 128 bytes), then the line of the visit bytes
            const uint32_t toff = (uint32_t)sub * 128u < (uint32_t)OFF_EL + 128u ? (uint32_t)sub * 128u : OFF_VIS;
            v = *reinterpret_cast<const uint32_t*>(wrecs + __umul24(nd, ROWS) + toff);
        }
        return v;
    };

    // =============================================================================================
    // expand (mcts_gpu.jl:250-302) of the previous rollout's leaf, then backUp (:306-328) + the recomputation of every row
    // the backup makes stale (:114-169)
    // =============================================================================================
    if constexpr (ROLE == ROLE_ITEMS) {                              // the tree wave's carry (published by its select call)
        C.leafn = xch[4 * gl]; C.spw = xch[4 * gl + 1]; C.leaf_meta = xch[4 * gl + 2];
        wcount = ufirst(xch[4 * NG]);
    }
    if (SF.do_expand) {   // PHASE expand: load logits
        const TreePar& T = tree_par();
        const GamePar& P = T.G;
        const uint32_t nwl = wcount;
        const uint32_t free0 = (uint32_t)(NG - GPW);                 // list entries that round 0 already takes (sparse waves)
        const int rounds = 1 + (nwl > free0 ? (int)((nwl - free0 + (uint32_t)(NG - 1)) / (uint32_t)NG) : 0);
        ItemRows<KPR> R;
        const uint32_t gid = live ? T.game_id[slot] : 0u;
        const uint32_t stp = live ? T.slot_ply[slot] : 0u;           // the game's ply: part of the key of its uniforms
        {   // the uniforms of the rows this call makes: U(seed; game, step, rollout whose leaf is expanded / backed up, depth);
            // lane sub draws depths 4 sub .. 4 sub + 3 (and 4 (sub + G) .. in a narrow group; deeper nodes, rare: drawn where they are needed)
#pragma unroll
            for (int b = 0; b < 8 / G; ++b) {
                float uq[4];
                uniform_search4(T.seed, gid, stp, SF.rollout - 1u, (uint32_t)(sub + b * G), uq);
                *reinterpret_cast<float4*>(utab + g * 32 + 4 * (sub + b * G)) = make_float4(uq[0], uq[1], uq[2], uq[3]);
            }
        }
        uint32_t sink = 0u;
        if constexpr ((PFM == 1 || PFM == 3 || PFM == 4) && ROLE != ROLE_EXPAND) sink = item_touch(0, nwl);   // the first item's record starts travelling towards L2 now
        // ---------------------------------------------------------------------------- expand (lane-group g = game g)
        const int lf = (int)C.leafn;
        uint32_t ml = live ? (LEAN ? C.leaf_meta : gmeta[lf]) : (uint32_t)M_TERM;   // (one memory round trip less on the rollout's chain)
        const bool term = (ml & M_TERM) != 0;
        const bool doexp = live && !term;
        float vleaf = 0.0f;
        const uint32_t spw = C.spw;
        float x[KPL], xr[KPR], sden = 1.0f; int npos = 0;   // x: the lane's block of logits / softmax numerators by ACTION; xr: its block of the node row
        bool wide = false, fdx = false;                               // (fdx is wave-uniform)
        if constexpr (ROLE == ROLE_ITEMS) {                           // (the helper only needs the value of the leaf)
            if (doexp) vleaf = (LEAN && LIO) ? (reinterpret_cast<const float*>(io_blk) + (size_t)g * io_lgs)[A] : T.v_eval[slot];
        }
        if (ROLE != ROLE_ITEMS && doexp) {
            constexpr bool lio = LEAN && LIO;
            const float* src = inject ? T.prior_eval + (size_t)slot * A
                                      : (lio ? reinterpret_cast<const float*>(io_blk) + (size_t)g * io_lgs : T.logits + (size_t)slot * T.LGS);
            vleaf = lio ? src[A] : T.v_eval[slot];
            const WPos<NC> st = grp_load_pos<NC, REV>(wstates + (gnode0 + (uint32_t)lf));
            if constexpr (LEAN) {
#pragma unroll
                for (int j = 0; j < KPL; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(src + k0 + j);
                    x[j] = a.x; x[j + 1] = a.y; x[j + 2] = a.z; x[j + 3] = a.w;
                }
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (j < nval) ? x[j] : -__builtin_inff();
            } else {
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (j < nval) ? src[k0 + j] : (inject ? 0.0f : -__builtin_inff());
            }
            if (!inject) {                                            // softmax!(prior) (:417), source-order sum   // PHASE expand: softmax
                float mx = -__builtin_inff(), mnn = -__builtin_inff();         // mnn = max of -x = -min over the real actions
#pragma unroll
                for (int j = 0; j < KPL; ++j) { mx = x[j] > mx ? x[j] : mx; const float nx_ = (j < nval) ? -x[j] : -__builtin_inff(); mnn = nx_ > mnn ? nx_ : mnn; }
                mx = grp_max<G>(mx); mnn = grp_max<G>(mnn);
                wide = !(mx + mnn <= 55.0f);                          // softmax numerators may fall below 2^-80: no fast quotients on this node
#pragma unroll
                for (int j = 0; j < KPL; ++j) x[j] = (j < nval) ? (exact ? exp_spec(x[j] - mx) : exp2_spec(x[j] - mx)) : 0.0f;
                float s;
                (void)grp_ordered_start<KPL, true, G>(x, sub, s, nlanes);
                fdx = FD && !wballot(wide);
                sden = s;
                if constexpr (!CMP) {
                    if (fdx) {
                        const float rs = fd_rcp(s);
#pragma unroll
                        for (int j = 0; j < KPL; j += 2) fd_div2(x[j], s, rs, x[j + 1], s, rs, x[j], x[j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < KPL; j += 2) div_pair(x[j], s, x[j + 1], s, x[j], x[j + 1]);
                    }
                    if (capture) {
#pragma unroll
                        for (int j = 0; j < KPL; ++j) if (j < nval) T.prior_eval[(size_t)slot * A + k0 + j] = x[j];
                    }
                }
            }
            bool lg[KPR];                                             // legal mask; masked priors (:260-268 / :284-290)   // PHASE expand: legal mask + normalize sum
            const uint32_t lmask = legal_block<FAM, NC, KPL>(P, st, k0, nval);
            const int nl = grp_sum<G>(__builtin_popcount(lmask));
            if constexpr (CMP) {
                // the masked softmax NUMERATORS of the lane's actions go to the slots of their ranks among the root's legal actions
                // (an action of the root's legal set that is taken at this node leaves the illegal mark -0 in its slot, and so do the
                // slots past the root's legal count), then every lane takes its KPR slots and divides what is legal by the denominator:
                // the same quotients as x / s of the action form, for fewer actions
                const WPos<NC> rst = grp_load_pos<NC, REV>(wstates + gnode0);
                const uint32_t rmask = legal_block<FAM, NC, KPL>(P, rst, k0, nval);
                int rank = grp_excl_prefix8<G>(__builtin_popcount(rmask), sub);
                float* const cs = reinterpret_cast<float*>(tab);      // (the group's edge table is idle during the expansion: 2 V >= G KPR floats)
                const float nz = __uint_as_float(0x80000000u);
#pragma unroll
                for (int j = 0; j < KPR; j += 4) *reinterpret_cast<float4*>(cs + r0 + j) = make_float4(nz, nz, nz, nz);
                AGZ_WSYNC();
#pragma unroll
                for (int j = 0; j < KPL; ++j) {
                    if ((rmask >> j) & 1u) { if (rank < G * KPR) cs[rank] = ((lmask >> j) & 1u) ? x[j] : nz; ++rank; }   // (the engine guarantees rank < G KPR: A - ply legal actions)
                }
                // ... and a root that breaks the guarantee is REPORTED, not served with a truncated row (the ply loop fails the generation)
                if (__builtin_expect(rank > G * KPR, 0)) atomicAdd(T.rank_fault, 1ull);
                AGZ_WSYNC();
#pragma unroll
                for (int j = 0; j < KPR; j += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(cs + r0 + j);
                    xr[j] = a.x; xr[j + 1] = a.y; xr[j + 2] = a.z; xr[j + 3] = a.w;
                }
                AGZ_WSYNC();
#pragma unroll
                for (int j = 0; j < KPR; ++j) { lg[j] = !(__float_as_uint(xr[j]) >> 31); xr[j] = lg[j] ? xr[j] : 0.0f; }
                if (!inject) {
                    if (fdx) {
                        const float rs = fd_rcp(sden);
#pragma unroll
                        for (int j = 0; j < KPR; j += 2) fd_div2(xr[j], sden, rs, xr[j + 1], sden, rs, xr[j], xr[j + 1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < KPR; j += 2) div_pair(xr[j], sden, xr[j + 1], sden, xr[j], xr[j + 1]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < KPR; ++j) {
                    lg[j] = (lmask >> j) & 1u;
                    xr[j] = lg[j] ? x[j < KPL ? j : 0] : 0.0f;
                }
            }
            float normalize;
            (void)grp_ordered_start<KPR, true, G>(xr, sub, normalize, nlr);
            const bool rootmix = lf == 0 && T.training;               // :270-275 vs :277-279, :292-294   // PHASE expand: mix / divide
            const float Af = (float)nl;
            float qn_[KPR];
            // (the root is expanded in the FIRST rollout of a search only: the other 63 take the plain quotients — no 0.75 x, no select per action,
            //  no 0.25 / A — behind one wave-uniform test)
            const bool fdn = fdx && !wballot(doexp && !(normalize >= 7.8886090522101181e-31f));        // 2^-100
            auto quotients = [&](auto MIX) {
                constexpr bool mix = decltype(MIX)::value;
                if (fdn) {
                    const float rn = fd_rcp(normalize);
#pragma unroll
                    for (int j = 0; j < KPR; j += 2)
                        fd_div2((mix && rootmix) ? 0.75f * xr[j] : xr[j], normalize, rn, (mix && rootmix) ? 0.75f * xr[j + 1] : xr[j + 1], normalize, rn, qn_[j], qn_[j + 1]);
                } else {
#pragma unroll
                    for (int j = 0; j < KPR; j += 2)
                        div_pair((mix && rootmix) ? 0.75f * xr[j] : xr[j], normalize, (mix && rootmix) ? 0.75f * xr[j + 1] : xr[j + 1], normalize, qn_[j], qn_[j + 1]);
                }
#pragma unroll
                for (int j = 0; j < KPR; ++j) {
                    float pr = (mix && rootmix) ? (lg[j] ? qn_[j] + 0.25f / Af : 0.0f) : qn_[j];
                    if (j >= nvr) pr = 0.0f;
                    xr[j] = pr;
                    npos += pr > 0.0f ? 1 : 0;
                }
            };
            if (__builtin_expect(wballot(rootmix) != 0ull, 0)) quotients(std::true_type{}); else quotients(std::false_type{});
            if (__builtin_expect(lf == 0, 0)) {                       // root expansion: policy == prior is what copy_pol sees for V <= 2
#pragma unroll
                for (int j = 0; j < KPR; ++j) if (j < nvr) T.policy_final[(size_t)slot * A + r0 + j] = xr[j];
                C.root_exp = 1u;
            }
            npos = grp_sum<G>(npos);                                  // "A" of :125-131 never changes after the expansion
        }
        if constexpr ((PF || PF3) && ROLE != ROLE_EXPAND) item_fetch(R, 0, nwl, PF3 ? 1 : 0);   // the first round's rows travel while the leaf's rows are written
        if (ROLE != ROLE_ITEMS && doexp) {
            // policy = prior (:297-299): the first revisit samples from these running sums; their total is prior_rem (:120-124, no   // PHASE expand: running sums + write rows
            // child yet)
            float total;
            const float st0 = grp_ordered_start<KPR, true, G>(xr, sub, total, nlr);
            const int Dl = (int)((spw >> 16) & 0xffu);                // depth of the leaf = expanded nodes above it
            const float ul = Dl < 32 ? utab[g * 32 + Dl] : uniform_search(T.seed, gid, stp, SF.rollout - 1u, (uint32_t)Dl);
            ChildWords<KPR> nocd;
#pragma unroll
            for (int j = 0; j < KPR / 4; ++j) nocd.w[j] = 0u;
            const uint32_t nx = sample_next(xr, st0, ul, nocd, nocd, -1, 0u);
            uint8_t* rec = wrecs + __umul24(gnode0 + (uint32_t)lf, ROWS);
#pragma unroll
            for (int j = 0; j < KPR; j += 4) {                        // (the per-action rows only: a fresh node has no edge)
                *reinterpret_cast<float4*>(rec + OFF_P + (size_t)(r0 + j) * 4) = make_float4(xr[j], xr[j + 1], xr[j + 2], xr[j + 3]);
                *reinterpret_cast<uint32_t*>(rec + OFF_RK + (size_t)(r0 + j)) = 0u;
                *reinterpret_cast<uint32_t*>(rec + OFF_CID + (size_t)(r0 + j)) = 0u;
            }
            ml |= M_EXPANDED;                                         // :256
            if (lead) {
                gmeta[lf] = ml;
                *reinterpret_cast<uint4*>(rec) = make_uint4(__float_as_uint(total), nx, (uint32_t)npos | (wide ? AUX_SLOW : 0u), 0u);
                if (NXL) nxw[gnode0 + (uint32_t)lf] = nx16(nx);
            }
        } else if (__builtin_expect(ROLE != ROLE_ITEMS && live && lf == 0, 0)) {
#pragma unroll
            for (int j = 0; j < KPL; ++j) if (j < nval) T.policy_final[(size_t)slot * A + k0 + j] = 0.0f;   // terminal root
        }
        STAMPW(1);
        // ---- what the backup adds at the ancestors (:312-324): value_1 = 1 - v at even levels (the parent is level 0), value_2 =   // PHASE values of the backup
        // 1 - value_1 at odd ones — the alternation value <- 1 - value is 2-periodic from its first step (1 - x is exact for x in
        // [0.5, 1], and one of value_1, value_2 lies there); a terminal leaf starts from (1 + player*r)/2 in Float64 (:314)
        if (ROLE != ROLE_EXPAND && lead) {
            const int tv2 = (int)((ml >> M_TV_SHIFT) & 3u);
            const float v0 = term ? 0.5f * (float)tv2 : vleaf;
            const float v1 = 1.0f - v0, v2 = 1.0f - v1;
            valtab[g] = make_float4(v1, v2, __uint_as_float((term ? 1u : 0u) | (((spw >> 16) & 0xffu) << 8) | ((uint32_t)lf << 16) | (doexp ? 1u << 24 : 0u)), 0.0f);
        }
        AGZ_WSYNC();

        STAMPW(2);
        // ---------------------------------------------------------------------------- work items
        const TreePar& TI = tree_par();                               // (the item loop's own view of the parameters)
#if AGZ_LATE_PRIO
        // a wave with many item rounds is the one its workgroup will wait for in front of the network phase: it goes first (A/B switch)
        if constexpr (LEAN && ROLE == ROLE_ALL) { if (rounds > AGZ_LATE_PRIO) __builtin_amdgcn_s_setprio(1); }
#endif
        const bool recompute = !(TI.final_ || SF.fin);                 // after the last rollout of a search nobody descends again   // PHASE items: loop control
#pragma unroll 1
        for (int r = 0; r < (ROLE == ROLE_EXPAND ? 0 : rounds); ++r) {
            if constexpr (!PF && !PF3 && !PF4) item_fetch(R, r, nwl);
            if constexpr (PF4) { if (r == 0) item_fetch(R, 0, nwl); }
            if constexpr (PF3) item_fetch(R, r, nwl, 2);              // priors and child bytes of THIS item: used after Newton
            const bool valid = R.valid, special = r == 0 && g < GPW;
            const int gi = R.gi;   // PHASE items: fetch item
            const int node = (int)(R.ent & 0xffu), mr = (int)((R.ent >> 8) & 0xffu), dpt_e = (int)((R.ent >> 16) & 0xffu);
            const bool created = special && (R.ent & SP_CREATED);
            const int move = created ? mr : -1;                       // the action of a NEW edge (an old one is known by its rank)
            const uint32_t ind = (uint32_t)((valid ? gi : gl) * V) + (uint32_t)(R.ent & 0xffu);   // the item's node in the wave's arrays
            const float4 vt = valtab[gi];
            const uint32_t vflags = __float_as_uint(vt.z);
            const bool iterm = vflags & 1u;
            const int D = (int)((vflags >> 8) & 0xffu), ileaf = (int)((vflags >> 16) & 0xffu);
            const int dpt = special ? D - 1 : dpt_e;                  // depth of the item's node
            const int level = D - 1 - dpt;
            const float w = (level & 1) ? vt.y : vt.x;                // 1 - value at this level
            uint8_t* const rec = wrecs + __umul24(ind, ROWS);
            STAMPW(3);
            // ---- backUp of this edge (:319-320)   // PHASE items: backUp of the edge, prior_rem re-sum
            const uint32_t ax_z = valid ? R.ax_z : 0u;                // (a group without an item: no visits, no children, prior_rem 0)
            const uint32_t vism = created ? 0u : R.vism;              // a new edge: q = 0, no visit yet
            const float qm = created ? 0.0f : R.qm;
            const float vis = (float)vism;
            float nq;
            if (__builtin_expect(wballot(valid && iterm) != 0, 0)) {
                const float nqf = (vis * qm + w) / (vis + 1.0f);
                const float nqd = (float)(((double)(vis * qm) + (double)w) / (double)(vis + 1.0f));
                nq = iterm ? nqd : nqf;
            } else nq = (vis * qm + w) / (vis + 1.0f);
            const uint32_t npos = ax_z & 0xffu, nvis = ((ax_z >> 8) & 0xffu) + 1u, nch_old = (ax_z >> 16) & 0xffu;
            const uint32_t nch = nch_old + (created ? 1u : 0u);
            const uint32_t rank1 = created ? nch : (uint32_t)mr;      // creation rank + 1 of the edge taken (:183-191 for a new one)
            ChildWords<KPR> rkw;                                      // rank bytes of the block, the new edge registered
            {   const uint32_t idx = (uint32_t)(move - r0);
#pragma unroll
                for (int j = 0; j < KPR / 4; ++j) rkw.w[j] = (created && (idx >> 2) == (uint32_t)j) ? (R.rk[j] | (nch << ((idx & 3u) * 8u))) : R.rk[j];
            }
            float prem_raw = valid ? __uint_as_float(R.ax_x) : 0.0f;   // sum of the priors of childless actions, before lambda
            if (__builtin_expect(wballot(valid && created) != 0, 0)) {
                // the node loses one childless action: re-sum prior_rem in source order (:120-124), once per rollout
                float m[KPR];
#pragma unroll
                for (int j = 0; j < KPR; ++j) {
                    const uint32_t rk = (rkw.w[j / 4] >> (8 * (j & 3))) & 0xffu;
                    m[j] = (created && rk == 0) ? R.p[j] : 0.0f;
                }
                float tot;
                (void)grp_ordered_start<KPR, true, G>(m, sub, tot, nlr);
                prem_raw = created ? tot : prem_raw;
            }
            if (valid && lead) {
                float2* const el = reinterpret_cast<float2*>(rec + OFF_EL) + (rank1 - 1u);
                if (created) {
                    *el = make_float2(nq, R.pm);                      // the new edge: q, prior of its action
                    rec[OFF_RK + move] = (uint8_t)nch; rec[OFF_CID + move] = (uint8_t)ileaf;   // creation rank + 1, node id (:183-191)
                } else el->x = nq;
                rec[OFF_VIS + (rank1 - 1u)] = (uint8_t)(vism + 1u);
            }
            const uint32_t auxz = npos | (nvis << 8) | (nch << 16) | (ax_z & AUX_SLOW);
            const bool FDr = FD && !wballot(ax_z & AUX_SLOW);      // (wave-uniform)
            if (!recompute) {
                if (valid && lead) *reinterpret_cast<uint4*>(rec) = make_uint4(__float_as_uint(prem_raw), 0u, auxz, 0u);   // (nobody descends again: the LDS word is not read)
                if constexpr (PF || PF3 || PF4) { if (r + 1 < rounds) item_fetch(R, r + 1, nwl, PF3 ? 1 : 0); }
                continue;
            }
            STAMPW(4);
            // ---- the node's edges in creation order -> the group's table (Newton's sums :144-148 read them by rank, the per-action   // PHASE items: edge table
            // q by the rank byte of the action)
            {
                const uint32_t no = valid ? nch_old : 0u;              // (a group without an item has no edge)
                tab[sub] = (uint32_t)sub < no ? R.e0 : make_float2(0.0f, 0.0f);
                if (__builtin_expect(wballot(no > (uint32_t)G) != 0, 0)) {
                    // a root's entries G .. 4 G - 1 arrived with the item; any other node with more than G children, and a root's
                    // entries from 4 G on, are read here (rare)
                    const bool pre = node == 0;
                    float2 e[3] = {R.e1, R.e2, R.e3};
#pragma unroll
                    for (int b = 1; b < 4; ++b) {
                        const uint32_t i = (uint32_t)(G * b + sub);
                        if (wballot(no > (uint32_t)(G * b)) == 0) break;
                        float2 ev = e[b - 1];
                        if (!pre && i < no) ev = *reinterpret_cast<const float2*>(rec + OFF_EL + i * 8u);
                        if (i < no) tab[i] = ev;
                    }
                    for (uint32_t b8 = (uint32_t)(4 * G); wballot(no > b8) != 0; b8 += (uint32_t)G) {
                        const uint32_t i = b8 + (uint32_t)sub;
                        if (i < no) tab[i] = *reinterpret_cast<const float2*>(rec + OFF_EL + i * 8u);
                    }
                }
            }
            AGZ_WSYNC();
            if (valid && lead) { if (created) tab[rank1 - 1u] = make_float2(nq, R.pm); else tab[rank1 - 1u].x = nq; }   // the edge just updated
            AGZ_WSYNC();
            float qa[KPR];                                            // q by action: an action without a child reads the zero pair
#pragma unroll
            for (int j = 0; j < KPR; ++j) {
                const int rk = (int)((rkw.w[j / 4] >> (8 * (j & 3))) & 0xffu);
                qa[j] = tab[rk - 1].x;
            }
            // ---- :116-138   // PHASE items: lambda, alpha0
            const float nf = 1.0f + (float)nvis, Af = (float)npos;
            const float lnum = TI.cpuct * sqrt_count(nf), lden = Af + nf;
            const float lambda = FDr ? fd_div(lnum, lden, fd_rcp(lden)) : lnum / lden;   // :132
            const float prior_rem = prem_raw * lambda;               // :134
            float am = 0.0f;
#pragma unroll
            for (int j = 0; j < KPR; ++j) {
                const float lp = lambda * R.p[j];
                const float gap = lp > 1e-4f ? lp : 1e-4f;
                const float c = qa[j] + gap;
                am = __builtin_fmaxf(c, am);                          // (no NaN can occur: one v_max_f32)
            }
            float alpha = grp_max<G>(am);
            STAMPW(5);
            // ---- Newton (:141-162): element 0 is the prior_rem term, elements 1..nch the children in creation order   // PHASE items: Newton
            {
                float err = __builtin_inff();
                // block 0 (the prior_rem term and the first G - 1 children) stays in registers for every iteration; further blocks of G
                // children are read from the group's table as long as ANY group of the wave still has children left (a wave-uniform
                // loop: groups with fewer children add zeros)
                const bool v0 = sub <= (int)nch;
                float top0 = 0.0f, qv0 = 0.0f;
                if (sub == 0) top0 = prior_rem;
                else if (v0) { const float2 e = tab[sub - 1]; top0 = lambda * e.y; qv0 = e.x; }
                for (int it = 0; it < 100; ++it) {
                    float t, uu;
                    {
                        const float bot = alpha - qv0;
                        if (FDr) fd_div_pair(top0, bot, -top0, bot * bot, t, uu); else div_pair(top0, bot, -top0, bot * bot, t, uu);
                        t = v0 ? t : 0.0f; uu = v0 ? uu : 0.0f;
                    }
                    float a = t, b = uu;
                    grp_pull_sums<G>(a, t, b, uu);
                    for (int j0 = G; wballot(j0 <= (int)nch) != 0; j0 += G) {
                        const int c = j0 + sub;
                        const bool vc = c <= (int)nch;
                        const int ci = vc ? c - 1 : 0;
                        const float2 e = tab[ci];
                        const float top = lambda * e.y, bot = alpha - e.x;
                        if (FDr) fd_div_pair(top, bot, -top, bot * bot, t, uu); else div_pair(top, bot, -top, bot * bot, t, uu);
                        t = vc ? t : 0.0f; uu = vc ? uu : 0.0f;
                        a += t; b += uu;
                        grp_pull_sums<G>(a, t, b, uu);
                    }
                    const float S = grp_bcast<G>(a), gg = grp_bcast<G>(b);
                    const float newerr = S - 1.0f;
                    if (newerr < 0.001f || newerr == err) break;
                    alpha -= FDr ? fd_div(newerr, gg, fd_rcp(gg)) : newerr / gg;
                    err = newerr;
                }
            }
            STAMPW(6);
            // the 128-register build touches the next item's record HERE, not at the head of the round: a full round ahead, the lines
            // touched by the 512 waves of an XCD (5.2 MB per round) do not survive in its 4 MB L2 until they are read; half a round
            // (~6 us) still covers an HBM miss (first ply at 32768 games 4.22 -> 4.15 ms)
            if constexpr (PFM == 1 || PFM == 3 || PFM == 4) {
                asm volatile("" :: "v"(sink));                        // (keeps the touch loads alive; they completed long ago)
                if (r + 1 < rounds) sink = item_touch(r + 1, nwl);
            }
            // ---- the policy row (:165-169) and its running sums (:172-181)   // PHASE items: policy row
            float pol[KPR];
            if (FDr) {
#pragma unroll
                for (int j = 0; j < KPR; j += 2)
                    fd_div_pair(lambda * R.p[j], alpha - qa[j], lambda * R.p[j + 1], alpha - qa[j + 1], pol[j], pol[j + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < KPR; j += 2)
                    div_pair(lambda * R.p[j], alpha - qa[j], lambda * R.p[j + 1], alpha - qa[j + 1], pol[j], pol[j + 1]);
            }
            ChildWords<KPR> cdk;
#pragma unroll
            for (int j = 0; j < KPR / 4; ++j) cdk.w[j] = R.cd[j];
            // the rows of this item are dead: the next item's start travelling now (its table entries are written after the
            // AGZ_WSYNC below)
            AGZ_WSYNC();
            if constexpr (PF || PF3 || PF4) { if (r + 1 < rounds) item_fetch(R, r + 1, nwl, PF3 ? 1 : 0); }
            if (__builtin_expect(wballot(valid && node == 0 && SF.last) != 0, 0)) {     // copy_pol (:330-339): the row the last descent samples from
                if (valid && node == 0) {
                    int nv_ = nvr;
                    asm volatile("" : "+v"(nv_));                     // (left visible, the KPR compares are hoisted out of the item loop: 2 KPR scalar registers held for a branch taken once per search)
#pragma unroll
                    for (int j = 0; j < KPR; ++j) if (j < nv_) TI.policy_final[(size_t)(slot_base + gi) * A + r0 + j] = pol[j];
                }
            }
            // the child bytes are needed past the prefetch of the next item: kept aside   // PHASE items: running sums + sampling
            STAMPW(7);
            const float u = dpt < 32 ? utab[gi * 32 + dpt] : uniform_search(TI.seed, TI.game_id[valid ? slot_base + gi : sl], TI.slot_ply[valid ? slot_base + gi : sl], SF.rollout - 1u, (uint32_t)dpt);
            int cnt = 0;
            if (!AGZ_FAST_COUNT || __builtin_expect(!sample_count_fast(pol, u, cnt), 0)) {
                // some running sum of the wave's items lies within rounding of its uniform: the source-order chain decides
                float dummy;
                float c = grp_ordered_start<KPR, false, G>(pol, sub, dummy, nlr);
                cnt = 0;
#pragma unroll
                for (int j = 0; j < KPR; ++j) { c += pol[j]; cnt += c < u ? 1 : 0; }
            }
            const uint32_t nx = sample_finish(pol, cnt, cdk, rkw, move, (uint32_t)ileaf);
            if (valid && lead) {
                *reinterpret_cast<uint4*>(rec) = make_uint4(__float_as_uint(prem_raw), nx, auxz, 0u);
                if (NXL) nxw[ind] = nx16(nx);
            }
            STAMPW(8);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // rows written by one lane-group are read by the descent of another
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        AGZ_WSYNC();
        STAMPW(9);
    }

    // =============================================================================================
    // kdescendTree! (mcts_gpu.jl:100-199) over the stored running sums + decoder (:202-223)
    // =============================================================================================
    if (SF.do_select) {   // PHASE descent: rounds
        const TreePar& T = tree_par();
        const GamePar& P = T.G;
        if constexpr (LEAN) __builtin_amdgcn_s_setprio(2);
        // every expanded node carries the action its next visit samples and the child under it: the descent follows the words
        int node = 0, depth = 0;
        int create_from = -1, create_move = 0;
        uint32_t spnew = 0u;
        wcount = 0;                                                   // wave-uniform: entries of the work list so far
        // two copies of the loop: over the 16-bit words of the LDS table (nxw: child | valid << 7 | mr << 8, mr = the action of a new edge or the
        // rank + 1 of the edge taken — the fields are used as they lie: one AND or one shift each on the chain from level to level) or over the
        // records' 32-bit next words
        auto descend = [&](auto nxl_) {
            constexpr bool NX = decltype(nxl_)::value;
            auto word = [&](const uint32_t nd) -> uint32_t { if constexpr (NX) return (uint32_t)nxw[nd]; else return auxp(nd)->y; };
            auto valid_of = [](const uint32_t w) -> bool { return NX ? (w & 0x80u) != 0u : (w & NX_VALID) != 0u; };
            auto child_of = [](const uint32_t w) -> uint32_t { return NX ? w & 0x7fu : (w >> 8) & 0xffu; };
            auto action_of = [](const uint32_t w) -> uint32_t { return NX ? w >> 8 : w & 0xffu; };           // (of a word without a child)
            auto rank_of = [](const uint32_t w) -> uint32_t { return NX ? w >> 8 : (w >> 17) & 0xffu; };      // (of a word with a child: creation rank + 1)
            uint32_t nx = 0u;
            if (live && C.root_exp) nx = word(gnode0);
            bool descending = valid_of(nx);
            STAMPW(10);
            // (the lead lanes of the wave as a constant: a ballot of a predicate that is no compare costs a 0/1 select and a compare)
            constexpr uint64_t LEADS = ~0ull / ((G >= 64 ? 0ull : 1ull << G) - 1ull);   // bit 0 of every group of G lanes
            uint64_t dmask = wballot(descending);
            while (dmask) {
                if (descending) {
                    const uint32_t child = child_of(nx);
                    ++depth;
                    if (child == 0u) {                                 // :183-191: a new child is never expanded -> the descent ends
                        const uint32_t move = action_of(nx);
                        create_from = node; create_move = (int)move;
                        spnew = (uint32_t)node | (move << 8) | ((uint32_t)depth << 16) | SP_VALID | SP_CREATED;
                        descending = false;
                    } else {
                        const uint32_t nxc = word(gnode0 + child);    // (cleared when the child was created, set by its expansion)
                        STAMPW(11);
                        if (valid_of(nxc)) {                           // expanded child: the descent goes on (:192)
                            const uint64_t app = __builtin_amdgcn_read_exec() & LEADS;   // (only lanes of groups that go on are here)
                            if (lead) {
                                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(app >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)app, wcount));   // wcount + the list's new entries of lower lanes
                                const uint32_t e = (uint32_t)node | (rank_of(nx) << 8) | ((uint32_t)(depth - 1) << 16) | ((uint32_t)g << 24);   // (an old edge goes by its rank)
                                if (LEAN && pos < wl_cap_lds) *((AGZ_LDSP uint32_t*)wl_lds + pos) = e; else *((AGZ_GLBP uint32_t*)wl_g + pos) = e;
                            }
                            node = (int)child; nx = nxc;
                        } else {                                       // existing child that was never expanded: a terminal position
                            spnew = (uint32_t)node | (rank_of(nx) << 8) | ((uint32_t)depth << 16) | SP_VALID;
                            node = (int)child;
                            descending = false;
                        }
                    }
                }
                dmask = wballot(descending);
                wcount += (uint32_t)__popcll(dmask & LEADS);          // entries appended this round
                AGZ_WSYNC();
                STAMPW(12);
            }
        };
        if (NXL) descend(std::true_type{}); else descend(std::false_type{});
        if (lead) C.add_p += depth;                                   // one expanded node per level (the device counter of SURVEY 8(d)'s p)
        C.spw = spnew;

        WPos<NC> lst; bool have_state = false;   // PHASE create child (play, isOver)
        for (int i = 0; i < NC; ++i) { lst.p.c[i] = 0; lst.o.c[i] = 0; lst.lg.c[i] = 0; }
        lst.player = 1; lst.aux = 0;
        uint32_t mn = 0u;
        if (live && create_from >= 0) {                                    // :183-191 node creation (at most one per rollout)
            const uint32_t child = C.ncount; C.ncount += 1;
            const WPos<NC> ps = grp_load_pos<NC, REV>(wstates + (gnode0 + (uint32_t)create_from));
            int amove = create_move;
            if constexpr (CMP) {
                // rows by rank: the move is the create_move-th legal action of the ROOT — the lane whose block holds that rank finds the
                // action (the rel-th set bit of its part of the root's legal mask), the group sums the one non-zero answer
                const WPos<NC> rst = grp_load_pos<NC, REV>(wstates + gnode0);
                const uint32_t rmask = legal_block<FAM, NC, KPL>(P, rst, k0, nval);
                const int cnt = __builtin_popcount(rmask);
                const int rel = create_move - grp_excl_prefix8<G>(cnt, sub);
                uint32_t m = rmask;
#pragma unroll
                for (int i = 0; i < KPL - 1; ++i) m = i < rel ? m & (m - 1u) : m;
                amove = grp_sum<G>((rel >= 0 && rel < cnt) ? k0 + (int)__builtin_ctz(m) : 0);
            }
            lst = GM::play(P, ps, amove);
            have_state = true;
            int rr; const bool f = GM::isOver(P, lst, rr);
            uint32_t mc = (uint32_t)create_from | ((uint32_t)create_move << 8) | M_EXISTS | M_EVAL;
            if (f) mc |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
            if (lead) {
                ++C.add_new;
                wstates[gnode0 + child] = pack(lst);
                gmeta[child] = mc;
                reinterpret_cast<uint32_t*>(auxp(gnode0 + child))[1] = 0u;      // not expanded: no next word yet
                if (NXL) nxw[gnode0 + child] = (uint16_t)0u;
            }
            mn = mc; node = (int)child;
        } else if (live) mn = gmeta[node];
        if (live) {
            if (!(mn & M_EVAL)) {                                           // root on the first rollout
                lst = grp_load_pos<NC, REV>(wstates + (gnode0 + (uint32_t)node)); have_state = true;
                int rr; const bool f = GM::isOver(P, lst, rr);
                mn |= M_EVAL;
                if (f) mn |= M_TERM | ((uint32_t)(1 + lst.player * rr) << M_TV_SHIFT);
                if (lead) gmeta[node] = mn;
            }
            if (!have_state) lst = grp_load_pos<NC, REV>(wstates + (gnode0 + (uint32_t)node));
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
                constexpr bool lio = LEAN && LIO;                    // (the hand-over rows hold INP columns: the network skips the k-rows past them, agz_nn_wave.hpp KR0)
#pragma unroll
                for (int kk = 0; kk < NW * (8 / G); ++kk) {          // (a lane encodes 8 cells at a time; 8 / G bytes of every word in a narrow group)
                    const int k = kk / (8 / G), sb = sub + G * (kk % (8 / G));
                    const int j0 = 64 * k + 8 * sb;
                    if (j0 < T.INP) {
                        const uint32_t f = (uint32_t)(W[k] >> (8 * sb)) & 0xffu;
                        uint4 o;
                        o.x = ((f & 1u) ? 0x3F80u : 0u) | ((f & 2u) ? 0x3F800000u : 0u); o.y = ((f & 4u) ? 0x3F80u : 0u) | ((f & 8u) ? 0x3F800000u : 0u);
                        o.z = ((f & 16u) ? 0x3F80u : 0u) | ((f & 32u) ? 0x3F800000u : 0u); o.w = ((f & 64u) ? 0x3F80u : 0u) | ((f & 128u) ? 0x3F800000u : 0u);
                        if (j0 < T.INP && !lio) *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(T.planes) + (size_t)slot * T.INP + j0) = o;
                        if (lio) *reinterpret_cast<uint4*>(io_blk + (size_t)g * io_prowb + (size_t)j0 * 2) = o;
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
            C.leafn = (uint32_t)node; C.leaf_meta = mn;
        }
    }

    STAMPW(13);
    if constexpr (ROLE == ROLE_EXPAND) {                             // what the helper wave needs for the backup of this rollout
        if (SF.do_select) {
            if (lead) { xch[4 * g] = C.leafn; xch[4 * g + 1] = C.spw; xch[4 * g + 2] = C.leaf_meta; }
            if (lane == 0) xch[4 * NG] = wcount;
        }
    }
    // ---- bookkeeping: the stand-alone kernel hands the carry over through global memory; the whole-search kernel only at its end   // PHASE bookkeeping
    if (ROLE != ROLE_ITEMS && (!LEAN || SF.fin)) {
        const TreePar& T = tree_par();
        if (live && lead) {
            T.ncount[slot] = C.ncount;
            T.leaf[slot] = C.leafn;
            if constexpr (!LEAN) T.sp[slot] = C.spw;
            if (LEAN || SF.do_reset) { T.cnt_p[slot] = C.add_p; T.cnt_new[slot] = C.add_new; }
            else { T.cnt_p[slot] += C.add_p; T.cnt_new[slot] += C.add_new; }
        }
        if constexpr (!LEAN) { if (lane == 0) T.wl_n[wl_block] = wcount; }
    }
#ifdef AGZ_STAMPS
    STAMPW(14);
    AGZ_WSYNC();
    if (lane < 15 && T.dbg) T.dbg[(size_t)(T.slot0 / NG + bidx) * 16 + lane] += stamp_lds[lane];
#endif
}

#ifndef AGZ_EAGER_PF24
#define AGZ_EAGER_PF24 0    // 1: the register prefetch also with 24 actions per lane (the build the note below is about)
#endif
template <int FAM, int NC, int KPL, int WV = 4>
__global__ __launch_bounds__(64, WV) void k_rollout_eager(const TreePar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_eager[];
    const TreePar& T = tree_par();
    const StepFlags SF = {T.rollout, T.do_reset, T.do_expand, T.do_select, T.last, T.final_};
    EagerCarry C; uint32_t wcount = 0;
    // (register prefetch of the item rows in the 3-waves-per-SIMD build up to 16 actions per lane.  With 24 the build spills 19 vector
    //  registers, and this compiler placed four of the spill stores — the node count among them — at the head of the block that JOINS
    //  the expansion's `if (doexp)`, in front of its `s_or_b64 exec`: they ran for the games whose leaf was being expanded only, the item
    //  loop then reused the registers in every lane, and a game with a terminal leaf got back what an earlier launch had left in its
    //  scratch slot — one node-count increment lost (round-4 fuzz, Gobang 13x13; scratch/repro_13.py).  tests/test_code_objects.py scans
    //  every kernel of the library for that placement (scratch/spill_exec_check.py); -DAGZ_EAGER_PF24=1 rebuilds the case.)
    rollout_eager_body<FAM, NC, KPL, false, ((WV < 4 && (KPL <= 16 || AGZ_EAGER_PF24)) ? 2 : 1)>(SF, lds_eager, (int)blockIdx.x, C, nullptr, 0u, wcount);
}

}  // namespace agz
