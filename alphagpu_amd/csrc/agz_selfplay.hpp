// agz_selfplay.hpp — the ply loop of mcts(actor,visits,ngames,buffer) (mcts_gpu.jl:494-561) on the device:
// sample capture (push_buffer, mainGobang.jl:54-68), move choice (:518-524), play / isOver (:530-531),
// order-preserving compaction of finished games (:550-553) and re_init (:557-561).  The reference does this
// in a single-threaded host loop with a 32 MB D2H + H2D per ply; here only one 4-byte count crosses PCIe.
#pragma once
#include "agz_device.hpp"
#include "agz_plystep.hpp"

namespace agz {

// one wavefront per slot
template <int FAM, int NR, int NC>
__global__ __launch_bounds__(256) void k_advance(const PlyPar T) {
    const int slot = ufirst((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (slot >= T.L) return;
    (void)advance_slot<FAM, NR, NC, false>(T, slot);
}

// policy_final rows that a search with node rows by the root's legal RANK (agz_tree_eager.hpp KPR_) left in rank order -> action order, in
// place: entry k of the result is the rank(k)-th entry of the row if action k is legal at the root, else 0.  One wavefront per slot;
// every load of a row comes before its first store.
template <int FAM, int NR, int NC>
__global__ __launch_bounds__(256) void k_spread_policy(const PlyPar T) {
    using G = Game<FAM, NC>;
    const GamePar& P = T.G;
    const int lane = lane_id();
    const int slot = ufirst((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (slot >= T.L) return;
    const int A = P.A;
    const WPos<NC> root = load_pos<NC>(T.states + (size_t)slot * T.V);
    float* const row = const_cast<float*>(T.policy_final) + (size_t)slot * A;
    float out[NR]; int base = 0;
    for (int r = 0; r < NR; ++r) {
        const int k = 64 * r + lane;
        const bool legal = k < A && G::canPlay(P, root, k);
        const uint64_t m = wballot(legal);
        const int rank = base + (int)__popcll(m & ((1ull << lane) - 1ull));
        out[r] = legal ? row[rank] : 0.0f;
        base += (int)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int r = 0; r < NR; ++r) { const int k = 64 * r + lane; if (k < A) row[k] = out[r]; }
}

// sums of the per-slot descent counters of the last search, added to two device accumulators (roofline bookkeeping without a
// device-to-host copy per ply)
__global__ __launch_bounds__(256) void k_fold_counters(const uint32_t* cnt_p, const uint32_t* cnt_new, int L, unsigned long long* acc) {
    unsigned long long a = 0, b = 0;
    for (int i = (int)(blockIdx.x * 256 + threadIdx.x); i < L; i += (int)gridDim.x * 256) { a += cnt_p[i]; b += cnt_new[i]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&acc[0], a); atomicAdd(&acc[1], b); }
}

// exclusive scan of the 0/1 flags alive[0..L) by one workgroup of 1024 threads -> newslot[], total -> *count.  Wave w owns a
// contiguous segment and walks it 64 flags at a time (coalesced loads, counts by ballot): one pass for the segment totals, a
// 16-entry scan of them, one pass for the slots.
// hostflag (may be null): a 64-bit word in host-visible memory that receives (seq << 32 | total) — the host polls it instead of
// waiting for a copy and a stream synchronisation (the ply loop's only round trip to the host)
// hostflag[1] receives the number of games started so far (*next_game; written before hostflag[0]).
// dead (may be null): the number of slots this ply's k_advance left without a game.  0 — every game goes on or its slot was refilled, the
// usual case while games still start — makes the compaction the identity: the scan is skipped, count[1] = 1 tells k_compact so.
__global__ __launch_bounds__(1024) void k_scan_alive(const uint32_t* alive, uint32_t* newslot, int L, uint32_t* count,
                                                     unsigned long long* hostflag, uint32_t seq, const unsigned long long* next_game = nullptr,
                                                     unsigned long long* dead = nullptr, const unsigned long long* finished = nullptr) {
    __shared__ uint32_t part[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (dead && ufirst((uint32_t)(*dead == 0ull)) != 0u) {        // (the word is stable: k_advance has finished; only this kernel resets it)
        if (t == 0) {
            *count = (uint32_t)L; count[1] = 1u;
            if (hostflag) {
                if (next_game) __hip_atomic_store(hostflag + 1, *next_game, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (finished) __hip_atomic_store(hostflag + 2, *finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(hostflag, ((unsigned long long)seq << 32) | (unsigned long long)(uint32_t)L, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    const int seg = ((L + 1023) / 1024) * 64;                    // flags per wave (a multiple of 64)
    const int b = w * seg, e = (b + seg < L) ? b + seg : L;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t s = 0;
    for (int i = b; i < e; i += 256) {                            // four chunks per turn: their loads are in flight together
        uint32_t a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = i + 64 * j + lane; a[j] = k < e ? alive[k] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) s += (uint32_t)__popcll(wballot(a[j] != 0u));
    }
    if (lane == 0) part[w] = s;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const uint32_t v = part[j]; base += j < w ? v : 0u; total += v; }
    for (int i = b; i < e; i += 256) {
        uint32_t a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = i + 64 * j + lane; a[j] = k < e ? alive[k] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t m = wballot(a[j] != 0u);
            const int k = i + 64 * j + lane;
            if (k < e) newslot[k] = base + (uint32_t)__popcll(m & below);
            base += (uint32_t)__popcll(m);
        }
    }
    if (t == 0) {
        *count = total; count[1] = 0u;
        if (dead) *dead = 0ull;
        if (hostflag) {
            if (next_game) __hip_atomic_store(hostflag + 1, *next_game, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (finished) __hip_atomic_store(hostflag + 2, *finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(hostflag, ((unsigned long long)seq << 32) | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// re_init (:359-373): move surviving games to their compacted slots
__global__ void k_compact(const PlyPar T, const uint32_t* newslot, const uint32_t* gid_in, uint32_t* gid_out, uint32_t* ply_out) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= T.L || !T.alive[slot]) return;
    uint32_t ns = T.identity[1] ? (uint32_t)slot : newslot[slot];    // (k_scan_alive: nothing to compact this ply)
    T.states[(size_t)ns * T.V] = T.newpos[slot];
    gid_out[ns] = gid_in[slot];
    ply_out[ns] = T.slot_ply[slot];
}

// ---------------------------------------------------------------------------------------------------
// packed sample records (Sample, mainGobang.jl:34-43 + update_buffer :70-80 + decode mcts_gpu.jl:464-474)
//   {u32 game_id, i32 ply, i32 move, f32 value, i8 player, i8 pad[3], f32 policy[A], i8 state[2VS], i8 fstate[FS]}
// ---------------------------------------------------------------------------------------------------
// PoolSample order of the generation's samples (mcts_gpu.jl:513-516: one push per ply and per game still running, ply-major, games
// in slot = game-id order) built on the device: block p takes ply p — its first record is the number of samples of earlier plies,
// sum over games of min(nplies, p), then the games with nplies > p in order (ballot ranks).  order[s] = game << 8 | ply.
// ring0 / cap: game g of the call sits at entry (ring0 + g) mod cap of the per-game arrays (chained calls; 0 / anything >= G otherwise)
__global__ __launch_bounds__(1024) void k_sample_order(const int32_t* nplies, int G, uint32_t* order, unsigned long long* total, uint32_t ring0 = 0,
                                                       uint32_t cap = 0xffffffffu) {
    __shared__ unsigned long long red[16];
    __shared__ uint32_t part[16];
    const int p = (int)blockIdx.x, t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned long long b = 0;
    for (int g = t; g < G; g += 1024) { const int n = nplies[(ring0 + (uint32_t)g) % cap]; b += (unsigned long long)(n < p ? n : p); }
    for (int o = 32; o > 0; o >>= 1) b += __shfl_down(b, o, 64);
    if (lane == 0) red[w] = b;
    __syncthreads();
    unsigned long long base = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) base += red[j];
    const uint64_t below = (1ull << lane) - 1ull;
    for (int g0 = 0; g0 < G; g0 += 1024) {
        const int g = g0 + t;
        const bool in = g < G && nplies[(ring0 + (uint32_t)g) % cap] > p;
        const uint64_t m = wballot(in);
        __syncthreads();                                           // (part of the previous chunk has been read)
        if (lane == 0) part[w] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { const uint32_t v = part[j]; before += j < w ? v : 0u; all += v; }
        if (in) order[base + before + (uint32_t)__popcll(m & below)] = ((uint32_t)g << 8) | (uint32_t)p;
        base += all;
    }
    if (total && p == (int)gridDim.x - 1 && t == 0) *total = base;   // (the last ply's block ends at the number of samples)
}

struct PackPar {
    int32_t A, VS, FS, max_plies, rec_bytes;
    uint32_t game_id_base;
    const uint64_t* s_boards; const float* s_policy; const int16_t* s_move; const uint8_t* s_net;
    const int32_t* g_nplies; const int8_t* g_result; const Pos* g_final;
    const uint32_t* order;   // [n] (g << 8 | ply)  PoolSample order
    uint32_t ring0, cap, k0; // chained calls: game g of the call is entry (ring0 + g) mod cap of the per-game arrays and game k0 + g of the chain
    int64_t n;
    uint8_t* out;
};
// One WAVE per record, a grid-stride loop over the records (a workgroup per record — 25 million of them for a call of 20 generations — was
// bound by the rate at which workgroups are dispatched: 27.7 ms for 5.1 M records of 592 bytes).
__global__ __launch_bounds__(256) void k_pack_samples(const PackPar T) {
    const int lane = (int)(threadIdx.x & 63);
    const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); s < T.n; s += nw) {
        uint32_t key = T.order[s];
        const int gl = (int)(key >> 8), ply = (int)(key & 0xff);
        const int g = (int)((T.ring0 + (uint32_t)gl) % T.cap);
        uint8_t* rec = T.out + (size_t)s * T.rec_bytes;
        const size_t sidx = (size_t)g * T.max_plies + ply;
        const int player = (ply & 1) ? -1 : 1;                  // Position().player == 1 and play() flips it
        const int res = T.g_result[g];
        if (lane == 0) {
            reinterpret_cast<uint32_t*>(rec)[0] = T.game_id_base + T.k0 + (uint32_t)gl;
            reinterpret_cast<int32_t*>(rec)[1] = ply;
            reinterpret_cast<int32_t*>(rec)[2] = T.s_move[sidx];
            reinterpret_cast<float*>(rec)[3] = (float)((1 + res * player) / 2.0);      // mainGobang.jl:76
            reinterpret_cast<int8_t*>(rec)[16] = (int8_t)player;
            rec[17] = T.s_net[sidx];                                                    // the network that searched this ply (agz_set_network_tag)
            rec[18] = rec[19] = 0;
        }
        float* pol = reinterpret_cast<float*>(rec + 20);
        int8_t* st = reinterpret_cast<int8_t*>(rec + 20 + 4 * T.A);
        int8_t* fs = st + 2 * T.VS;
        const Pos fin = T.g_final[g];
        for (int k = lane; k < T.A; k += 64) pol[k] = T.s_policy[sidx * T.A + k];
        for (int j = lane; j < 2 * T.VS; j += 64) {
            int b = j < T.VS ? j : j - T.VS;
            uint64_t w = T.s_boards[sidx * 6 + (j < T.VS ? 0 : 3) + (b >> 6)];
            st[j] = (int8_t)((w >> (b & 63)) & 1);
        }
        for (int j = lane; j < T.FS; j += 64) {
            int bit = (int)((fin.p[j >> 6] >> (j & 63)) & 1);
            int v = bit ? fin.player : -fin.player;                                    // decode :464-474
            fs[j] = (int8_t)(v * player);                                              // fstate * player :77
        }
        for (int j = 20 + 4 * T.A + 2 * T.VS + T.FS + lane; j < T.rec_bytes; j += 64) rec[j] = 0;
    }
}

// ---- known-answer test of the game plugins ON THE DEVICE (agz_perft): one level of a breadth-first perft ------------------------
// thread i takes position i of the level: a finished game counts in term[] (result +1 / 0 / -1 in absolute colours) and is not
// extended (Game::isOver); at the last level every position counts as a node; otherwise every legal action (Game::canPlay) is played
// (Game::play) — into the next level, or, one ply above the last level (count_only), straight into the counters.
// cnt: [0] nodes at the final depth, [1..3] finished games by result, [4] positions written to `out`, [5] positions dropped (out full)
template <int FAM, int NC>
__global__ __launch_bounds__(256) void k_perft_level(const GamePar P, const Pos* in, const unsigned long long n_in, Pos* out,
                                                     const unsigned long long cap_out, unsigned long long* cnt, const int remaining) {
    using G = Game<FAM, NC>;
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const WPos<NC> s = unpack<NC>(in[i]);
    int r;
    if (G::isOver(P, s, r)) {
        atomicAdd(&cnt[r == 1 ? 1 : (r == 0 ? 2 : 3)], 1ull);            // (isOver reports the winner in absolute colours)
        if (remaining == 0) atomicAdd(&cnt[0], 1ull);
        return;
    }
    if (remaining == 0) { atomicAdd(&cnt[0], 1ull); return; }
    unsigned long long nodes = 0, t[3] = {0, 0, 0};
    for (int a = 0; a < P.A; ++a) {
        if (!G::canPlay(P, s, a)) continue;
        const WPos<NC> c = G::play(P, s, a);
        if (remaining == 1) {                                             // the last level is counted, not stored
            int rc;
            if (G::isOver(P, c, rc)) ++t[rc == 1 ? 0 : (rc == 0 ? 1 : 2)];
            ++nodes;
        } else {
            const unsigned long long k = atomicAdd(&cnt[4], 1ull);
            if (k < cap_out) out[k] = pack(c); else atomicAdd(&cnt[5], 1ull);
        }
    }
    if (nodes) atomicAdd(&cnt[0], nodes);
    for (int j = 0; j < 3; ++j) if (t[j]) atomicAdd(&cnt[1 + j], t[j]);
}

}  // namespace agz
