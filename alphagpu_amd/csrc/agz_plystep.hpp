// agz_plystep.hpp — the ply step of ONE slot of the self-play loop (mcts_gpu.jl:513-549): sample capture (push_buffer, mainGobang.jl:54-68),
// move choice (:518-524), play / isOver (:530-531), and the next game for a slot whose game has ended.  Shared by k_advance (one launch
// per ply, agz_selfplay.hpp) and the persistent self-play kernels (agz_selfplay_small.hpp: a workgroup steps its own slots).
#pragma once
#include "agz_device.hpp"

namespace agz {

// Games that MIGRATE between the workgroups of the persistent self-play kernel (agz_selfplay_small.hpp, age classes): a first-in first-out
// queue of {root position, game id, ply} in device memory.  ctr[0] = tickets handed to producers (tail), ctr[1] = tickets handed to
// consumers (head); entry t & mask belongs to ticket t; `ready` == t + 1 once the producer of ticket t has filled it, 0 once its consumer
// has copied it out.  The entry's words are written and read with agent-scope relaxed atomics (the XCDs' L2s are not coherent with each
// other for plain accesses inside a launch) around which the wave waits for its own memory operations: no cache-wide write-back or
// invalidation per game.
struct MigEntry { uint32_t w[24]; };                               // w[0..19] the Pos, w[20] game id, w[21] ply, w[22] ready, w[23] -
struct MigQ { unsigned long long* ctr; MigEntry* buf; uint32_t mask, backlog_max, age, rsv; };
enum { TAKE_POOL = 0, TAKE_POOL_THEN_QUEUE = 1, TAKE_QUEUE = 2, TAKE_QUEUE_THEN_POOL = 3 };

struct PlyPar {
    GamePar G;
    int32_t L, V, ply, tau_plies, all_actions;   // ply: the round of the lock-step loop (diagnostics; every game's own ply is slot_ply[slot])
    uint64_t seed;
    uint32_t game_id_base;
    Pos* states;              // [L][V] roots at node 0
    uint32_t* game_id;        // [L]
    uint32_t* slot_ply;       // [L] plies played so far by the slot's game (the reference's `round`, mcts_gpu.jl:484,556, per game)
    // more games than slots (agz_selfplay with ngames > max_games): a slot whose game has ended takes the next game that has not
    // started yet — game id game_id_base + k for the k-th start, Position(), ply 0 — until refill_total games have been started
    // (0: no refill).  Results are keyed by game id and ply, never by slot or by the round a game happens to start in.
    uint32_t refill_total;
    unsigned long long* next_game;   // games started so far (device counter)
    // chained calls (agz_selfplay_chain): game ids run on from call to call (game k of the chain has id game_id_base + k), the sample store
    // is a RING over k (k mod sample_games), and a call may start games of the NEXT call in slots that would otherwise idle.  k_cur_end:
    // the games k < k_cur_end belong to the call that is running (a finished one counts in stats[8] — what the call waits for —, a later one in stats[9]).
    int32_t ring; uint32_t k_cur_end;
    const uint32_t* identity; // k_compact: k_scan_alive's count words ([1] != 0: every slot keeps its place)
    const float* policy_final;// [L][A]
    // per-slot scratch
    Pos* newpos;              // [L]
    uint32_t* alive;          // [L]
    // sample store, indexed by local game g = game_id - game_id_base (< sample_games)
    int32_t sample_games, max_plies;
    uint64_t* s_boards;       // [G][max_plies][6]  bplayer[3], bopponent[3] of the root
    float* s_policy;          // [G][max_plies][A]
    int16_t* s_move;          // [G][max_plies]
    uint8_t* s_net;           // [G][max_plies] the tag of the network that searched the ply (agz_set_network_tag; byte 17 of a packed record)
    uint32_t net_tag;
    int32_t* g_nplies;        // [G]
    int8_t* g_result;         // [G]
    Pos* g_final;             // [G]
    unsigned long long* stats;// [0] wins [1] draws [2] losses [3] total_plies [4] faults ... [7] slots left without a game by this ply's k_advance (reset by k_scan_alive)
    MigQ mq;                  // games on their way from one workgroup to another (persistent self-play kernel with age classes); buf == nullptr: none
};

// A slot without a game takes one that waits (the whole wave calls; the result is wave-uniform): from the POOL — the next game that has
// not started yet: game id game_id_base + k for the k-th start, Position(), ply 0 — or from the migration QUEUE, in the order `order`
// (TAKE_*).  Writes the slot's root (over states[slot][0] when INPLACE, else to newpos[slot]), game id and ply; returns 1 | the ply of
// the game << 8 if the slot has a game now, else 0.
// (a draw past the end of the pool is given back: once nobody draws any more the counter IS the number of games started, and a chain's
//  later calls go on from it with a larger pool.  Draws below refill_total are unique: a failed draw only happens once all of them have
//  been handed out.  One add per finished game — a compare-and-swap loop on this one address cost 2 ms per ply.)
__device__ __forceinline__ uint32_t draw_from_pool(const PlyPar& T, unsigned long long& k) {   // lane 0's result, broadcast
    uint32_t ok = 0u; uint32_t klo = 0u;
    if (lane_id() == 0) {
        const unsigned long long kk = atomicAdd(T.next_game, 1ull);
        if (kk >= (unsigned long long)T.refill_total) atomicAdd(T.next_game, ~0ull); else { ok = 1u; klo = (uint32_t)kk; }
    }
    ok = ufirst(ok); k = (unsigned long long)ufirst(klo);          // (game numbers fit 32 bits: agz_selfplay_chain checks)
    return ok;
}
template <bool INPLACE>
__device__ __forceinline__ void write_start_position(const PlyPar& T, const int slot, const unsigned long long k) {
    if (lane_id() == 0) {
        const GamePar& P = T.G;
        Pos next;
        for (int i = 0; i < 3; ++i) { next.p[i] = P.start_p[i]; next.o[i] = P.start_o[i]; next.lg[i] = P.start_lg[i]; }
        next.player = (int8_t)P.start_player; next.aux = (int8_t)P.start_aux;
        for (int i = 0; i < 6; ++i) next.pad[i] = 0;
        if (INPLACE) T.states[(size_t)slot * T.V] = next; else T.newpos[slot] = next;
        T.game_id[slot] = T.game_id_base + (uint32_t)k;
        T.slot_ply[slot] = 0u;
    }
}
// Memory ordering of the queue: the payload words and the per-entry ready word are written and read with AGENT-scope RELAXED atomics
// around `s_waitcnt vmcnt(0)`, not with release / acquire: on gfx942 / gfx950 an agent-scope atomic access is an sc1 access that bypasses the
// per-XCD L2 (the L2s of the eight XCDs are not coherent for plain accesses inside a launch), and vmcnt(0) means the wave's stores have been
// acknowledged by the memory side — so "payload, wait, ready word" is ordered for every other XCD.  A release / acquire pair would write back and
// invalidate an L2 that holds a gigabyte of tree records in flight, once per migrated game.  That is a property of THESE targets:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "agz_plystep.hpp: the migration queue relies on gfx942 / gfx950 semantics of agent-scope relaxed atomics (sc1, L2 bypass) and of s_waitcnt vmcnt(0)"
#endif
// Ring bound: a producer takes its ticket before it looks at the backlog of the NEXT push, so the backlog can exceed backlog_max by one entry per
// wave at most; the engine refuses a launch unless backlog_max + (waves of the launch) < ring entries (agz_engine.hip run_games_persist), which is
// what keeps two producers from meeting on one entry.
template <bool INPLACE>
__device__ __forceinline__ uint32_t pop_migrated(const PlyPar& T, const int slot) {
    const MigQ& q = T.mq;
    if (!q.buf) return 0u;
    const int lane = lane_id();
    uint32_t ok = 0u, hlo = 0u;
    if (lane == 0) {
        for (int tries = 0; tries < 4 && !ok; ++tries) {
            const unsigned long long h = __hip_atomic_load(q.ctr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t = __hip_atomic_load(q.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h >= t) break;
            if (atomicCAS(q.ctr + 1, h, h + 1ull) == h) { ok = 1u; hlo = (uint32_t)h; }
        }
    }
    ok = ufirst(ok); hlo = ufirst(hlo);
    if (!ok) return 0u;
    MigEntry* const e = q.buf + (hlo & q.mask);
    // the producer of this ticket may still be writing (it took its ticket before the head could pass it)
    while (ufirst(__hip_atomic_load(&e->w[22], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != hlo + 1u) __builtin_amdgcn_s_sleep(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t v = 0u;
    if (lane < 22) v = __hip_atomic_load(&e->w[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < 20) { uint32_t* d = reinterpret_cast<uint32_t*>(INPLACE ? T.states + (size_t)slot * T.V : T.newpos + slot); d[lane] = v; }
    if (lane == 20) T.game_id[slot] = v;
    if (lane == 21) T.slot_ply[slot] = v;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the entry has been read: its place is free again
    if (lane == 0) __hip_atomic_store(&e->w[22], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 1u | (rdlane(v, 21) << 8);                              // ... | the game's ply << 8
}
template <bool INPLACE>
__device__ __forceinline__ uint32_t take_game(const PlyPar& T, const int slot, const int order) {
    if (!T.refill_total) return 0u;
    unsigned long long k;
    if (order == TAKE_QUEUE || order == TAKE_QUEUE_THEN_POOL) {
        const uint32_t r = pop_migrated<INPLACE>(T, slot);
        if (r) return r;
        if (order == TAKE_QUEUE) return 0u;
    }
    if (draw_from_pool(T, k)) { write_start_position<INPLACE>(T, slot, k); return 1u; }
    if (order == TAKE_POOL_THEN_QUEUE) return pop_migrated<INPLACE>(T, slot);
    return 0u;
}
// The game of slot `slot` (INPLACE form: root at states[slot][0]) leaves for the queue and the slot starts game k of the pool, which the
// caller has drawn.  The whole wave calls.
__device__ __forceinline__ void push_migrating(const PlyPar& T, const int slot, const unsigned long long k) {
    const MigQ& q = T.mq;
    const int lane = lane_id();
    uint32_t tlo = 0u;
    if (lane == 0) tlo = (uint32_t)atomicAdd(q.ctr, 1ull);
    tlo = ufirst(tlo);
    MigEntry* const e = q.buf + (tlo & q.mask);
    // (the place is free unless the queue has wrapped onto an entry whose consumer is still copying: the backlog is bounded far below the ring)
    while (ufirst(__hip_atomic_load(&e->w[22], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) __builtin_amdgcn_s_sleep(4);
    uint32_t v = 0u;
    if (lane < 20) v = reinterpret_cast<const uint32_t*>(T.states + (size_t)slot * T.V)[lane];
    if (lane == 20) v = T.game_id[slot];
    if (lane == 21) v = T.slot_ply[slot];
    if (lane < 22) __hip_atomic_store(&e->w[lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the entry is complete (and the slot's old root has been read) ...
    if (lane == 0) __hip_atomic_store(&e->w[22], tlo + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before it is published
    write_start_position<true>(T, slot, k);
}

// The ply step of ONE slot by one wavefront: sample capture, move choice, play / isOver, and — with refilled slots — the next game.
// INPLACE = false (k_advance, one launch per ply): the slot's next root goes to newpos[] and k_compact moves it to the slot's new place;
// INPLACE = true (the persistent self-play kernels: a workgroup keeps its slots for the whole call, nothing is compacted): the next
// root is written over the slot's root.  Returns whether the slot still holds a game (wave-uniform) | the ply of that game << 8.
// ranked: the search left the slot's policy_final row in the order of the root's legal RANKS (node rows by legal rank, agz_tree_eager.hpp
// KPR_): entry k of the policy is the rank(k)-th entry of the row if action k is legal at the root, else 0 (what k_spread_policy does
// in place for the one-launch-per-ply form).  order: where a slot whose game has ended looks for its next game (TAKE_*).
template <int FAM, int NR, int NC, bool INPLACE>
__device__ __forceinline__ uint32_t advance_slot(const PlyPar& T, const int slot, const bool ranked = false, const int order = TAKE_POOL) {
    using G = Game<FAM, NC>;
    const GamePar& P = T.G;
    const int lane = lane_id();
    const int A = P.A;
    const uint32_t gid = ufirst(T.game_id[slot]);
    const uint32_t kg = gid - T.game_id_base;                      // the game's number in the call (in the chain of calls)
    const int g = T.ring ? (int)(kg % (uint32_t)T.sample_games) : (int)kg;
    WPos<NC> root = load_pos<NC>(T.states + (size_t)slot * T.V);
    const int ply = (int)ufirst(T.slot_ply[slot]);                 // this game's round (:484, :556)
    float pol[NR];
    if (ranked) {
        int base = 0;
        for (int r = 0; r < NR; ++r) {
            const int k = 64 * r + lane;
            const bool legal = k < A && G::canPlay(P, root, k);
            const uint64_t m = wballot(legal);
            const int rank = base + (int)__popcll(m & ((1ull << lane) - 1ull));
            pol[r] = legal ? T.policy_final[(size_t)slot * A + rank] : 0.0f;
            base += (int)__popcll(m);
        }
    } else
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; pol[r] = k < A ? T.policy_final[(size_t)slot * A + k] : 0.0f; }
    const bool in_range = T.ring || (g >= 0 && g < T.sample_games);
    const bool keep = in_range && ply < T.max_plies;
    const int np_end = ply + 1 < T.max_plies ? ply + 1 : T.max_plies;
    if (keep) {                                                     // push_buffer: root planes (as boards) + policy
        size_t sidx = (size_t)g * T.max_plies + ply;
        for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) T.s_policy[sidx * A + k] = pol[r]; }
        if (lane < 6) {
            uint64_t w = 0;
            for (int i = 0; i < NC; ++i) { if (lane == i) w = root.p.c[i]; if (lane == 3 + i) w = root.o.c[i]; }
            T.s_boards[sidx * 6 + lane] = w;
        }
    }
    // ---- move choice (:518-524)
    int c = -1;
    if (ply < T.tau_plies) {
        // sample(lp, Weights(pol[lp])): t = u * sum(w), first index whose running sum >= t (source order)
        float total = 0.0f; bool st = false; uint64_t nzm[NR]; float run[NR];   // run: the running sum up to and including the lane's action
        for (int r = 0; r < NR; ++r) {
            nzm[r] = wballot(64 * r + lane < A && pol[r] != 0.0f);
            run[r] = chain64(pol[r], nzm[r], total, false, 0.0f, st);
        }
        const float u = ufirst(uniform_move(T.seed, gid, (uint32_t)ply));
        const float tt = u * total;
        // duel (:606): sample(1:maxActions, Weights(policy)) walks ALL actions, zero weights included — the same index as the
        // nonzero-list walk of self-play (:519-520) whenever tt > 0, and u is never 0 (uniform_move): a zero-weight action is never
        // chosen
        // (the walk compares the SAME running sums the total came from: one ordered pass over the row, not two)
        int last = -1;
        for (int r = 0; r < NR; ++r) {
            if (!nzm[r]) continue;
            last = 64 * r + 63 - __builtin_clzll(nzm[r]);
            if (c >= 0) continue;
            uint64_t ge = wballot(((nzm[r] >> lane) & 1ull) && !(run[r] < tt));
            if (ge) c = 64 * r + __builtin_ctzll(ge);
        }
        if (c < 0) c = last;
    } else {
        // argmax(pol): first maximum
        float best = -__builtin_inff();
        for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; best = (k < A && pol[r] > best) ? pol[r] : best; }
        best = ufirst(wave_max(best));
        for (int r = 0; r < NR && c < 0; ++r) {
            uint64_t eq = wballot(64 * r + lane < A && pol[r] == best);
            if (eq) c = 64 * r + __builtin_ctzll(eq);
        }
        if (c < 0) c = 0;
    }
    bool fault = c < 0 || ply >= 254;                               // (no game of this library lasts 254 plies: a loop that would not end is a fault, not a hang)
    if (!fault) {
        bool ok = wballot(lane == 0 && G::canPlay(P, root, c)) != 0;   // "faute" guard (:526-529)
        fault = !ok;
    }
    if (keep && lane == 0) T.s_net[(size_t)g * T.max_plies + ply] = (uint8_t)T.net_tag;   // which network searched this ply (agz_set_network_tag)
    if (fault) {
        // the game is abandoned (the call reports AGZ_ERR_ILLEGAL_MOVE at its end); its slot takes the next game like that of a finished one,
        // so that the batch the host has queued plies for (run-ahead) keeps its size
        if (lane == 0) { atomicAdd(&T.stats[4], 1ull); if (T.ring) atomicAdd(&T.stats[kg < T.k_cur_end ? 8 : 9], 1ull); if (keep) T.s_move[(size_t)g * T.max_plies + ply] = (int16_t)c; if (in_range) { T.g_nplies[g] = np_end; T.g_result[g] = 0; T.g_final[g] = pack(root); } }
        const uint32_t a = take_game<INPLACE>(T, slot, order);
        if (lane == 0) { if (!a) atomicAdd(&T.stats[7], 1ull); T.alive[slot] = a & 1u; }
        return a;
    }
    WPos<NC> np = G::play(P, root, c);
    int res; const bool f = G::isOver(P, np, res);
    uint32_t alive = f ? 0u : 1u;
    if (lane == 0) {
        if (keep) T.s_move[(size_t)g * T.max_plies + ply] = (int16_t)c;
        if (f) {
            if (in_range) { T.g_nplies[g] = np_end; T.g_result[g] = (int8_t)res; T.g_final[g] = pack(np); }
            atomicAdd(&T.stats[res == 1 ? 0 : (res == 0 ? 1 : 2)], 1ull);     // :541-547
            atomicAdd(&T.stats[3], (unsigned long long)ply);                   // tot_length += round (:535)
            if (T.ring) atomicAdd(&T.stats[kg < T.k_cur_end ? 8 : 9], 1ull);
        } else {
            if (INPLACE) T.states[(size_t)slot * T.V] = pack(np); else T.newpos[slot] = pack(np);
            T.slot_ply[slot] = (uint32_t)ply + 1u;
        }
    }
    if (f) {                                                                   // (wave-uniform) the slot takes a game that waits, if there is one
        alive = take_game<INPLACE>(T, slot, order);                            // (1 | the ply of the game taken << 8)
        if (!alive && lane == 0) {
            atomicAdd(&T.stats[7], 1ull);
            if (!INPLACE) T.newpos[slot] = pack(np);
            T.slot_ply[slot] = (uint32_t)ply + 1u;
        }
    }
    if (lane == 0) T.alive[slot] = alive & 1u;
    return f ? alive : (1u | (((uint32_t)ply + 1u) << 8));
}

}  // namespace agz
