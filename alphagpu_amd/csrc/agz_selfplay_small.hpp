// agz_selfplay_small.hpp — the self-play loop of mcts(actor,visits,ngames,buffer) (mcts_gpu.jl:494-561) as ONE launch per call, without a
// barrier between workgroups: a workgroup keeps its slots for the whole call and loops over the plies of ITS games by itself —
//     search (the body of k_search_small: V x {tree step, network forward})  ->  the root policy of each of its games (copy_pol :330-339)
//     ->  sample capture, move choice (tau rule, ordered sum :518-524), play / isOver (:530-531)  ->  a finished game's slot takes the
//     next game that has not started (an atomic counter), or goes dead when none waits
// and never meets the other workgroups.  With one launch per ply (agz_selfplay.hpp k_advance / k_scan_alive / k_compact around
// k_search_small) every ply ends on the slowest workgroup of 512 — measured on the headline shape: a workgroup's search takes 3.24 to
// 3.76 ms inside a 3.76 ms launch, mean / span 0.93 — then runs three small kernels and wakes the host.  Here the host only waits for the
// launch to end: the call is over when its own games are (a device counter every workgroup looks at between two plies) or when no slot
// holds a game any more.  Results are keyed by (game id, ply) as before — the uniforms, the tau rule, the sample index — so every game's
// samples are bit for bit those of the lock-step run, whatever slot and whatever moment it is played in (tests/test_gpu_parity.py,
// tests/test_gpu_scale_parity.py run the same oracle comparisons over both forms).
#pragma once
#include "agz_search_small.hpp"
#include "agz_plystep.hpp"

#ifndef AGZ_PERSIST_BP
#define AGZ_PERSIST_BP 0     // 1: the network phase of the persistent 128-wide kernels reads a group's operands ahead of its MFMAs (agz_nn_wave.hpp BP) also at four waves per SIMD (A/B)
#endif

#ifndef AGZ_PFM_RANKED
#define AGZ_PFM_RANKED AGZ_PFM_LOW
#endif
#ifndef AGZ_PERSIST_NXL
#define AGZ_PERSIST_NXL 1    // 1: the descent of the whole-search kernels follows next words kept in LDS (agz_tree_eager.hpp nxw) wherever they fit; 0: the records' (A/B)
#endif

namespace agz {

// what the persistent kernels share beside their search parameters (k_selfplay_small: SmallPar, k_selfplay_big: BigSearchPar — each with its
// TreePar at offset 0 of the argument segment)
struct PersistTail {
    PlyPar P;                 // the ply step (refill_total = games that may be started, ring / k_cur_end for chained calls, the migration queue);
                              // P.L = the slots of the launch, P.game_id / slot_ply / states = the search's
    uint32_t ngames_cur;      // chained call: stop once stats[8] (finished games of the running call) reaches this; 0: run until no slot holds a game
    int32_t flag_off;         // two LDS words of the workgroup: "some wave still has a game" / "stop" / "some game is young"
    unsigned long long* acc;  // [0] expanded nodes traversed, [1] nodes created (roofline bookkeeping), [2] searches of a game (x V = rollouts), [3] slots with a game at the end,
                              // [4] searches with rows by legal rank, [5] games pushed to the migration queue
    // age classes (AGE builds): a workgroup PREFERS old games (ply >= P.mq.age: rows by legal rank) if hash(its CU pair) < old16 (of 16), young ones otherwise;
    // class_by_block != 0 (tests): odd workgroups prefer old games
    uint32_t old16, class_by_block;
};
struct PersistPar { SmallPar S; PersistTail X; };

// The ply step of ALL the games of a wave at once, one game per lane-group (the form advance_slot has, agz_plystep.hpp, takes the whole
// wave for one game and the wave's games one after the other: 1.1 - 1.4 % of the persistent kernel's time on the 9x9 shapes, more with 16 games
// per wave).  Lane sub of group g holds the KPL actions sub KPL .. of game slot0 + g, as in the tree step; the ordered sum of the move
// choice is the tree step's turn-taking sum (bit-identical to the source-order loop: a zero weight adds nothing), the walk
// "first action whose running sum reaches u x total" (:518-524) is the number of running sums below it.  Same stores, same counters as
// advance_slot; a slot whose game has ended (or was abandoned after an illegal move) is reported in the returned mask and refilled by
// the caller (take_game is a wave-level operation).  Returns: bits 0 .. NG-1 games that go on, bits 16 .. 16+NG-1 games that ended.
template <int FAM, int NC, int KPL, int G>
__device__ __forceinline__ uint32_t advance_groups(const PlyPar& T, const int slot0, const uint32_t amask, const bool ranked) {
    using GM = Game<FAM, NC>;
    constexpr bool REV = FAM == F_REV;
    constexpr int NG = 64 / G;
    const GamePar& P = T.G;
    int lane = lane_id();
    asm volatile("" : "+v"(lane));                                // opaque once per ply: what depends on the lane only is not carried (in scratch) through the search
    const int g = lane / G, sub = lane % G;
    const bool lead = sub == 0;
    const bool active = (amask >> g) & 1u;
    const int slot = slot0 + (active ? g : (int)__builtin_ctz(amask | 0x80000000u) % NG);   // (an idle group reads the wave's first game: finite values, nothing stored)
    const int A = P.A;
    const int k0 = sub * KPL;
    const int nlanes = (A + KPL - 1) / KPL;
    const int nval = A - k0 < 0 ? 0 : (A - k0 > KPL ? KPL : A - k0);
    const uint32_t gid = T.game_id[slot];
    const uint32_t kg = gid - T.game_id_base;
    const int gi = T.ring ? (int)(kg % (uint32_t)T.sample_games) : (int)kg;
    const int ply = (int)T.slot_ply[slot];
    const WPos<NC> root = grp_load_pos<NC, REV>(T.states + (size_t)slot * T.V);
    const float* const row = T.policy_final + (size_t)slot * A;
    float pol[KPL];
    if (ranked) {       // the row is in the order of the root's legal ranks: entry k of the policy = the rank(k)-th entry if action k is legal, else 0
        const uint32_t lmask = legal_block<FAM, NC, KPL>(P, root, k0, nval);
        int rank = grp_excl_prefix8<G>(__builtin_popcount(lmask), sub);
#pragma unroll
        for (int j = 0; j < KPL; ++j) { const bool lg = (lmask >> j) & 1u; pol[j] = lg ? row[rank] : 0.0f; rank += lg ? 1 : 0; }
    } else {
#pragma unroll
        for (int j = 0; j < KPL; ++j) pol[j] = j < nval ? row[k0 + j] : 0.0f;
    }
    const bool in_range = T.ring || (gi >= 0 && gi < T.sample_games);
    const bool keep = active && in_range && ply < T.max_plies;
    const int np_end = ply + 1 < T.max_plies ? ply + 1 : T.max_plies;
    const size_t sidx = (size_t)gi * T.max_plies + ply;
    if (keep) {                                                   // push_buffer: root boards + policy
#pragma unroll
        for (int j = 0; j < KPL; ++j) if (j < nval) T.s_policy[sidx * A + k0 + j] = pol[j];
        if (lead) {
#pragma unroll
            for (int i = 0; i < 3; ++i) { T.s_boards[sidx * 6 + i] = i < NC ? root.p.c[i < NC ? i : 0] : 0ull; T.s_boards[sidx * 6 + 3 + i] = i < NC ? root.o.c[i < NC ? i : 0] : 0ull; }
            T.s_net[sidx] = (uint8_t)T.net_tag;
        }
    }
    // ---- move choice (:518-524)
    int c;
    bool anynz = false;
#pragma unroll
    for (int j = 0; j < KPL; ++j) anynz |= pol[j] != 0.0f;
    anynz = grp_max_i<G>(anynz ? 1 : 0) != 0;
    if (ply < T.tau_plies) {
        float total;
        const float st = grp_ordered_start<KPL, true, G>(pol, sub, total, nlanes);
        const float tt = uniform_move(T.seed, gid, (uint32_t)ply) * total;
        float run = st; int cnt = 0;
#pragma unroll
        for (int j = 0; j < KPL; ++j) { run += pol[j]; cnt += (j < nval && run < tt) ? 1 : 0; }
        c = grp_sum<G>(cnt);
        if (c >= A) {                                             // (cannot happen: u < 1, so the last running sum is not below u x total; kept like the walk's `c == last`)
            int last = -1;
#pragma unroll
            for (int j = 0; j < KPL; ++j) last = (j < nval && pol[j] != 0.0f) ? k0 + j : last;
            c = grp_max_i<G>(last);
        }
        if (!anynz) c = -1;
    } else {                                                      // argmax(pol): the first maximum
        float best = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < KPL; ++j) best = (j < nval && pol[j] > best) ? pol[j] : best;
        best = grp_max<G>(best);
        int first = 1 << 20;
#pragma unroll
        for (int j = KPL - 1; j >= 0; --j) first = (j < nval && pol[j] == best) ? k0 + j : first;
        first = -grp_max_i<G>(-first);
        c = first < A ? first : 0;
    }
    bool fault = c < 0 || ply >= 254;
    if (!fault) fault = !GM::canPlay(P, root, c);                 // "faute" guard (:526-529)
    const WPos<NC> np = GM::play(P, root, fault ? 0 : c);
    int res = 0;
    const bool f = !fault && GM::isOver(P, np, res);
    if (active && lead) {
        if (keep) T.s_move[sidx] = (int16_t)c;
        if (fault) {
            atomicAdd(&T.stats[4], 1ull);
            if (T.ring) atomicAdd(&T.stats[kg < T.k_cur_end ? 8 : 9], 1ull);
            if (in_range) { T.g_nplies[gi] = np_end; T.g_result[gi] = 0; T.g_final[gi] = pack(root); }
        } else if (f) {
            if (in_range) { T.g_nplies[gi] = np_end; T.g_result[gi] = (int8_t)res; T.g_final[gi] = pack(np); }
            atomicAdd(&T.stats[res == 1 ? 0 : (res == 0 ? 1 : 2)], 1ull);     // :541-547
            atomicAdd(&T.stats[3], (unsigned long long)ply);                   // tot_length += round (:535)
            if (T.ring) atomicAdd(&T.stats[kg < T.k_cur_end ? 8 : 9], 1ull);
        } else {
            T.states[(size_t)slot * T.V] = pack(np);
            T.slot_ply[slot] = (uint32_t)ply + 1u;
        }
    }
    // one bit per game: lane g of the result collects the lead lanes' verdicts
    const uint64_t ended = wballot(active && lead && (f || fault)), goes = wballot(active && lead && !(f || fault));
    uint32_t out = 0u;
#pragma unroll
    for (int i = 0; i < NG; ++i) out |= (uint32_t)((goes >> (i * G)) & 1ull) << i | (uint32_t)((ended >> (i * G)) & 1ull) << (16 + i);
    return out;
}

#ifdef AGZ_PSTAMPS
#define PSTAMP(x) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); x += n_ - ps_t; ps_t = n_; } while (0)
#else
#define PSTAMP(x) do { } while (0)
#endif

// The loop of a persistent self-play workgroup.  TW waves, each with the NG = 64 / G games of slots (blockIdx.x TW + wave) NG ...;
// tail() -> const PersistTail& (read from the kernel-argument segment where it is needed); search(amask, ranked, C): one mcts_single of the
// workgroup's games (all waves call it; it contains the workgroup barriers of the network phase), C = the games' carry, whose descent
// counters are read afterwards.
// KPR2 != 0 — AGE CLASSES (stone-placing games: Gobang, Hex).  A root at ply p has A - p legal actions, and a search whose roots all have
// at most 8 KPR2 of them may index its node rows by the root's legal RANK (agz_tree_eager.hpp KPR_: a third less arithmetic and traffic per
// work item on a 9x9 board from ply 17 on; per launch 3.41 instead of 4.07 ms).  With one launch per ply that needs every game of the
// BATCH to be old — lock-step generations only, the refilled batch is a mix of all ages.  Here it only needs every game of a WORKGROUP
// to be old, and workgroups trade games: a workgroup that prefers YOUNG games hands a game that has reached ply age = A - 8 KPR2 to a
// queue in device memory and starts a new game of the pool in its slot; a workgroup that prefers OLD games fills a slot whose game has
// ended from that queue.  Before every search a workgroup looks at its own games: all of them old -> the body with rows by rank
// (policy_final then leaves the search in rank order and the ply step reads it through the root's legal mask), else rows by action.
// The preference is a function of the CU (both workgroups of a CU, and the two CUs that share an instruction cache, run the same body
// most of the time).  Nothing depends on where a game is played: its uniforms, tau rule and sample index are keyed by (game id, ply).
//   * a young-preferring workgroup only lets a game go when a new game is there to take the slot (the pool has not run dry) and the
//     queue holds fewer than backlog_max games; otherwise it keeps the old game (and runs rows by action while it has young ones);
//   * an old-preferring workgroup takes from the queue only; a wave that has had an empty slot for three plies in a row also takes from
//     the pool (a run whose games end before they are old must not starve half the chip);
//   * slots empty at the entry (a chain's next call) and slots of a pool that has run dry take whatever waits, the queue included,
//     whatever the preference: the queue drains before the launch can end.
// GROUPSTEP: the ply step of all the wave's games at once, one per lane-group (advance_groups) — the 128-wide kernels; the wide-trunk
// kernels keep the one-game-after-the-other form (their ply step is 0.3 % of the time and the group form's registers cost the 128-register
// build of the 64-leaf network pass another 28 spilled registers)
// GPW_ (round 6): games of a tree wave when that is FEWER than its 64 / G lane-groups (0: all of them) — a game with few actions on narrow lane-groups
// fills a wave with 16 trees, and 32768 games are then two waves per SIMD: too few to hide a rollout's dependent chain.  With 8 games per
// wave the same games are four waves per SIMD, and the lane-groups without a game take work items of the wave's games (agz_tree_eager.hpp: sparse waves).
template <int FAM, int NC, int KPL, int G, int TW, bool AGE, bool GROUPSTEP, int GPW_ = 0, typename TailFn, typename SearchFn>
__device__ __forceinline__ void persist_loop(uint8_t* const lds, const TailFn tail, const SearchFn search) {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    constexpr int NG = GPW_ ? GPW_ : 64 / G;                      // games of a tree wave
    constexpr int NR = (G * KPL + 63) / 64;                       // 64-action rows of the ply step (rows past the game's actions are empty)
    static_assert(!AGE || G == 8, "age classes: 8 lanes per tree");
    const int lane = lane_id();
    // ---- which games does this workgroup prefer?  (AGE builds)
    bool pref_old = false;
    if constexpr (AGE) {
        const PersistTail& Q = tail();
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // CU_ID[11:8] (the pair 2k, 2k + 1 shares an instruction cache), SH_ID[12], SE_ID[15:13]; XCC_ID[3:0]
        const uint32_t key = ((xcc & 15u) << 8) | (((hw >> 12) & 15u) << 4) | ((hw >> 9) & 7u);
        const uint32_t hsh = (key * 2654435761u) >> 28;           // 0 .. 15
        pref_old = (Q.class_by_block & 1u) ? ((blockIdx.x & 1u) != 0u) : (hsh < Q.old16);
        pref_old = pref_old && Q.P.mq.buf != nullptr;
    }
    uint32_t starve = 0u;                                         // plies in a row this wave has had a slot without a game
    // ---- the games of this wave: slots slot0 .. slot0 + NG - 1; a slot without a game takes one that waits (a chain's next call, or slots
    // the last call left empty when its pool ran dry)
    uint32_t amask = 0u;                                          // bit g: slot slot0 + g holds a game (wave-uniform)
    {
        const PersistTail& Q = tail();
        const int slot0 = ((int)blockIdx.x * TW + wave) * NG;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int slot = slot0 + g;
            if (slot >= Q.P.L) break;
            uint32_t a = ufirst(Q.P.alive[slot]);
            if (!a) {
                a = take_game<true>(Q.P, slot, pref_old ? TAKE_QUEUE_THEN_POOL : TAKE_POOL_THEN_QUEUE) & 1u;
                if (a && lane == 0) Q.P.alive[slot] = 1u;
            }
            amask |= a << g;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    {
        uint32_t* const flag = reinterpret_cast<uint32_t*>(lds + tail().flag_off);
        if (threadIdx.x == 0) { flag[0] = 0u; flag[1] = 0u; }
    }
#ifdef AGZ_PSTAMPS
    unsigned long long ps_t = __builtin_amdgcn_s_memtime(), ps_flag = 0, ps_search = 0, ps_cnt = 0, ps_adv = 0;
#endif
#pragma unroll 1
    for (uint32_t it = 0;; ++it) {
        // ---- does the workgroup go on?  Some wave of it still has a game, and the call's own games are not all over.  Are all its games old?
        bool ranked = false;
        {
            const PersistTail& Q = tail();
            uint32_t* const flag = reinterpret_cast<uint32_t*>(lds + Q.flag_off);
            uint32_t* const fw = flag + (it & 1u);
            uint32_t young = 0u;
            if constexpr (AGE) {
                const int slot0 = ((int)blockIdx.x * TW + wave) * NG;
                const uint32_t p = (lane < NG && ((amask >> lane) & 1u)) ? Q.P.slot_ply[slot0 + lane] : 0xffffffffu;
                young = wballot(p < Q.P.mq.age) != 0ull ? 4u : 0u;
            }
            __syncthreads();                                      // (the word was cleared a whole ply ago / at the entry)
            if (lane == 0) {
                uint32_t f = (amask ? 1u : 0u) | young;
                if (wave == 0 && Q.ngames_cur) {
                    const unsigned long long fin = __hip_atomic_load(Q.P.stats + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (fin >= (unsigned long long)Q.ngames_cur) f |= 2u;
                }
                if (f) atomicOr(fw, f);
                if (wave == 0) flag[(it + 1u) & 1u] = 0u;         // (read for the last time before the barrier above)
            }
            __syncthreads();
            const uint32_t f = ufirst(*fw);
            if (!(f & 1u) || (f & 2u)) break;
            // (class_by_block bit 1: every WAVE decides for its own games — a young game makes its wave search with rows by action, not the whole workgroup;
            //  the two forms of the rollout loop meet at the same workgroup barriers)
            ranked = AGE && ((Q.class_by_block & 2u) ? !young : !(f & 4u));
        }
        PSTAMP(ps_flag);
        // ---- mcts_single (:376-462) for the games of this workgroup
        EagerCarry C = {1u, 0u, 0u, 0u, 0u, 0u, 0u};
        search(amask, ranked, C);
        // (policy_final of this wave's games was written by lanes of this wave: the root's work item of the last-but-one rollout)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        PSTAMP(ps_search);
        // ---- roofline bookkeeping: descent counters of the search (the lead lane of a game's lane-group holds them)
        {
            const PersistTail& Q = tail();
            if (Q.acc) {
                uint32_t ap = (lane % G) == 0 ? C.add_p : 0u, an = (lane % G) == 0 ? C.add_new : 0u;
                for (int o = 32; o > 0; o >>= 1) { ap += (uint32_t)__shfl_xor((int)ap, o, 64); an += (uint32_t)__shfl_xor((int)an, o, 64); }
                if (lane == 0) {
                    atomicAdd(Q.acc + 0, (unsigned long long)ap); atomicAdd(Q.acc + 1, (unsigned long long)an);
                    atomicAdd(Q.acc + 2, (unsigned long long)__builtin_popcount(amask));
                    if (ranked) atomicAdd(Q.acc + 4, (unsigned long long)__builtin_popcount(amask));
                }
            }
        }
        PSTAMP(ps_cnt);
        // ---- the ply step of each game (:513-561), one after the other, the whole wave on one game
        {
            const PersistTail& Q = tail();
            const int slot0 = ((int)blockIdx.x * TW + wave) * NG;
            const bool pool_open = Q.P.refill_total != 0u &&
                                   __hip_atomic_load(Q.P.next_game, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)Q.P.refill_total;
            // where a slot whose game ends looks for the next one: see the rules above
            // (a pool that has run dry is not asked again: every failed draw is two atomic operations on one address)
            int order = TAKE_POOL;
            if constexpr (AGE) order = !pool_open ? TAKE_QUEUE : (pref_old ? (starve >= 3u ? TAKE_QUEUE_THEN_POOL : TAKE_QUEUE) : TAKE_POOL_THEN_QUEUE);
            uint32_t next_mask = 0u, dead = 0u;
            // all the wave's games at once, one per lane-group; then the slots whose game has ended (and the empty ones), one after the other
            uint32_t adv = 0u;
            if constexpr (GROUPSTEP) {
                if (amask) adv = advance_groups<FAM, NC, KPL, G>(Q.P, slot0, amask, ranked);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the roots the lead lanes have written are read by whole waves below)
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                const int slot = slot0 + g;
                if (slot >= Q.P.L) break;
                uint32_t r;
                if (GROUPSTEP && ((adv >> g) & 1u)) r = 1u | (ufirst(Q.P.slot_ply[slot]) << 8);   // the game goes on
                else if (GROUPSTEP && ((amask >> g) & 1u)) {      // the game has ended: the slot takes a game that waits, if there is one
                    r = take_game<true>(Q.P, slot, order);
                    if (lane == 0) { if (!r) atomicAdd(&Q.P.stats[7], 1ull); Q.P.alive[slot] = r & 1u; }
                } else if ((amask >> g) & 1u) r = advance_slot<FAM, NR, NC, true>(Q.P, slot, ranked, order);
                else {                                            // a slot without a game: something may wait for it now
                    r = AGE ? take_game<true>(Q.P, slot, order) : 0u;
                    if (r && lane == 0) Q.P.alive[slot] = 1u;
                }
                if constexpr (AGE) {
                    // a young-preferring workgroup lets an old game go — when a new game can take its slot and the queue is not long
                    if ((r & 1u) && !pref_old && pool_open && Q.P.mq.buf && (r >> 8) >= Q.P.mq.age) {
                        uint32_t go = 0u;
                        if (lane == 0) {
                            const unsigned long long t = __hip_atomic_load(Q.P.mq.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long h = __hip_atomic_load(Q.P.mq.ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            go = t - h < (unsigned long long)Q.P.mq.backlog_max ? 1u : 0u;
                        }
                        unsigned long long k;
                        if (ufirst(go) && draw_from_pool(Q.P, k)) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // (the root lane 0 has just written is read by lanes 0..19)
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                            push_migrating(Q.P, slot, k);
                            if (lane == 0 && Q.acc) atomicAdd(Q.acc + 5, 1ull);
                        }
                    }
                }
                next_mask |= (r & 1u) << g;
                dead += (r & 1u) ^ 1u;
            }
            amask = next_mask;
            starve = dead ? starve + 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the new roots are read by the other lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        PSTAMP(ps_adv);
    }
#ifdef AGZ_PSTAMPS
    if (lane == 0 && tail().acc) { unsigned long long* a = tail().acc; atomicAdd(a + 8, ps_flag); atomicAdd(a + 9, ps_search); atomicAdd(a + 10, ps_cnt); atomicAdd(a + 11, ps_adv); }
#endif
    {
        const PersistTail& Q = tail();
        if (lane == 0 && Q.acc && amask) atomicAdd(Q.acc + 3, (unsigned long long)__builtin_popcount(amask));
    }
}

// one mcts_single (:376-462) of the workgroup's games inside the persistent kernel: the rollout loop of k_search_small.  KPR: rows per lane
// by the root's legal rank (0: rows by action); the records keep the stride they were allocated with either way.
template <int FAM, int NC, int KPL, int H, int TW, int WV, int G, int KPR, int GPW_ = 0>
__device__ __forceinline__ void persist_search(uint8_t* const lds_small, const uint32_t amask, EagerCarry& C) {
    constexpr int NWV = TW == 8 ? 8 : NW_WAVES, NG = 64 / G, GPW = GPW_ ? GPW_ : NG;   // (GPW games per wave: the rows of its block of the hand-over window and of the network's tile)
    // (rows by legal rank hold 8 instead of 12 entries per lane: AGZ_PFM_RANKED = 2 gives THAT form of the rollout loop the register prefetch of the item rows)
    constexpr int PFM_ = (G < 8 || WV < 3 || (WV == 3 && KPL <= 16) || KPL <= 4) ? 2 : ((KPR != 0 && KPR <= 8) ? AGZ_PFM_RANKED : AGZ_PFM_LOW);
    typedef const PersistPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto spar = [&]() -> const SmallPar& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistPar*)p)->S; };
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint32_t wcount = 0;
    const int V_ = spar().V;
#pragma unroll 1
    for (int k = 0; k <= V_; ++k) {
        int bx = (int)blockIdx.x;
        asm volatile("" : "+s"(bx));                              // opaque once per rollout (see k_search_small)
        const SmallPar& S = spar();
        uint8_t* const tree_lds = lds_small + (size_t)wave * S.tree_lds;
        uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_small + S.wl_off + (size_t)wave * S.wl_bytes);
        uint8_t* const io_blk = lds_small + S.io_off + (size_t)wave * S.io_bw;
        uint16_t* const nxw = (AGZ_PERSIST_NXL && S.nxw_off) ? reinterpret_cast<uint16_t*>(lds_small + S.nxw_off) + (size_t)wave * (size_t)(NG * S.V) : nullptr;   // next words of this wave's trees
        const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
        rollout_eager_body<FAM, NC, KPL, true, PFM_, true, ROLE_ALL, KPR, G>(
            SF, tree_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount, io_blk, S.io_prowb, S.io_lgs, nullptr, ~amask, KPR ? S.T.rec_bytes : 0u, nxw);
        if (k < S.V) {
            __builtin_amdgcn_s_setprio(3);
            const SmallPar& S = spar();
            mlp_wave_body<H, TW * GPW / 16, 2, true, true, (WV < 4), (WV < 3) || AGZ_PERSIST_BP, NWV>(S.F, lds_small, bx, lds_small + S.io_off, S.io_bw, S.io_lgs);
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
        }
    }
}

// The same template parameters as k_search_small; only the shapes whose every wave is a full tree wave (TW = 4 or 8, ROLE_ALL) are built.
// KPR2 != 0: age classes (above).
template <int FAM, int NC, int KPL, int H, int TW, int WV, int G = 8, int KPR2 = 0, int GPW_ = 0>
__global__ __launch_bounds__(64 * (TW == 8 ? 8 : NW_WAVES), WV) void k_selfplay_small(const PersistPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_small[];
    static_assert(offsetof(PersistPar, S) == 0 && offsetof(SmallPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    static_assert(TW == 4 || TW == 8, "every wave of the workgroup is a tree wave");
    static_assert(TW == (TW == 8 ? 8 : NW_WAVES), "tree waves == waves");
    constexpr bool AGE = KPR2 != 0;
    typedef const PersistPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto tail = [=]() -> const PersistTail& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistPar*)p)->X; };
    // (one copy of the rollout loop per row form: the two tree bodies inside ONE loop cost 38 spilled registers)
#ifdef AGZ_PLYSTEP_SERIAL
    constexpr bool GS = false;
#else
    constexpr bool GS = true;
#endif
    static_assert(!GPW_ || (!AGE && GPW_ * 2 == 64 / G && (TW * GPW_) % 16 == 0), "sparse waves: half of the lane-groups hold a game");
    persist_loop<FAM, NC, KPL, G, TW, AGE, GS, GPW_>(lds_small, tail, [&](const uint32_t amask, const bool ranked, EagerCarry& C) {
        if constexpr (AGE) {
            if (ranked) persist_search<FAM, NC, KPL, H, TW, WV, G, KPR2>(lds_small, amask, C);
            else persist_search<FAM, NC, KPL, H, TW, WV, G, 0>(lds_small, amask, C);
        } else persist_search<FAM, NC, KPL, H, TW, WV, G, 0, GPW_>(lds_small, amask, C);
    });
}

// the shapes of AGZ_SMALL_SHAPES in their full-batch form (64-game workgroups of eight tree waves, four waves per SIMD) and the narrow
// form of the few-action games (Connect4: 4 lanes per tree, 16 trees per wave, workgroups of four waves, two waves per SIMD)
// two workgroup shapes: 32 games (four waves, four workgroups per CU: half the waves at every barrier of the network phase — the default
// since it measured +1.7 % without and +2 % with age classes on the headline shape, although every layer's weights stream twice per 64
// games) and 64 games (eight waves, two per CU; AGZ_PERSIST_TW=8)
#define AGZ_PERSIST_VARIANTS(F, C, K, KW) KW template __global__ void k_selfplay_small<F, C, K, 128, 8, 4>(const PersistPar); \
                                          KW template __global__ void k_selfplay_small<F, C, K, 128, 4, 4>(const PersistPar);
// ... with age classes: (family, chunks, actions per lane, rows per lane by legal rank) — the 9x9 boards (81 actions, old from ply 17 on)
#define AGZ_PERSIST_AGE_VARIANTS(F, C, K, R, KW) KW template __global__ void k_selfplay_small<F, C, K, 128, 8, 4, 8, R>(const PersistPar); \
                                                 KW template __global__ void k_selfplay_small<F, C, K, 128, 4, 4, 8, R>(const PersistPar);
#define AGZ_PERSIST_AGE_SHAPES(X) X(F_LINE, 2, 12, 8) X(F_HEX, 2, 12, 8)
#define AGZ_PERSIST_NARROW_VARIANTS(F, C, K, GG, KW) KW template __global__ void k_selfplay_small<F, C, K, 128, 4, 2, GG>(const PersistPar); \
                                                     KW template __global__ void k_selfplay_small<F, C, K, 128, 4, 4, GG, 0, 32 / GG>(const PersistPar);   /* sparse: half of the lane-groups hold a game, four waves per SIMD */
#define AGZ_PERSIST_NARROW_SHAPES(X) X(F_C4, 1, 4, 4)

}  // namespace agz
