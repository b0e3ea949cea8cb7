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

namespace agz {

struct PersistPar {
    SmallPar S;               // the search (S.T at offset 0: rollout_eager_body reads its TreePar from the start of the argument segment);
                              // S.T.L = S.F.L = P.L = the slots of the launch, S.T.game_id / slot_ply / states = P's
    PlyPar P;                 // the ply step (refill_total = games that may be started, ring / k_cur_end for chained calls)
    uint32_t ngames_cur;      // chained call: stop once stats[8] (finished games of the running call) reaches this; 0: run until no slot holds a game
    int32_t flag_off;         // two LDS words of the workgroup: "some wave still has a game" / "stop"
    unsigned long long* acc;  // [0] expanded nodes traversed, [1] nodes created (roofline bookkeeping), [2] searches of a game (x V = rollouts), [3] slots with a game at the end
};

// The same template parameters as k_search_small; only the shapes whose every wave is a full tree wave (TW = 4 or 8, ROLE_ALL) are built.
template <int FAM, int NC, int KPL, int H, int TW, int WV, int G = 8>
__global__ __launch_bounds__(64 * (TW == 8 ? 8 : NW_WAVES), WV) void k_selfplay_small(const PersistPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_small[];
    static_assert(offsetof(PersistPar, S) == 0 && offsetof(SmallPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    static_assert(TW == 4 || TW == 8, "every wave of the workgroup is a tree wave");
    typedef const PersistPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto par = [&]() -> const PersistPar& { KArg p = karg; asm volatile("" : "+s"(p)); return *(const PersistPar*)p; };
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    constexpr int NWV = TW == 8 ? 8 : NW_WAVES;
    constexpr int NG = 64 / G;                                    // games of a tree wave
    constexpr int NR = (G * KPL + 63) / 64;                       // 64-action rows of the ply step (rows past the game's actions are empty)
    static_assert(TW == NWV, "tree waves == waves");
    const int lane = lane_id();
    uint8_t* const tree_lds = lds_small + (size_t)wave * par().S.tree_lds;
    uint8_t* const nn_lds = lds_small;
    uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_small + par().S.wl_off + (size_t)wave * par().S.wl_bytes);
    uint32_t* const flag = reinterpret_cast<uint32_t*>(lds_small + par().flag_off);
    // ---- the games of this wave: slots slot0 .. slot0 + NG - 1; a slot without a game takes one that waits (a chain's next call, or slots
    // the last call left empty when its pool ran dry)
    uint32_t amask = 0u;                                          // bit g: slot slot0 + g holds a game (wave-uniform)
    {
        const PersistPar& Q = par();
        const int slot0 = ((int)blockIdx.x * TW + wave) * NG;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int slot = slot0 + g;
            if (slot >= Q.P.L) break;
            uint32_t a = ufirst(Q.P.alive[slot]);
            if (!a && Q.P.refill_total) {
                if (lane == 0) {
                    Pos next;
                    a = start_next_game(Q.P, slot, next);
                    if (a) { Q.P.states[(size_t)slot * Q.P.V] = next; Q.P.slot_ply[slot] = 0u; Q.P.alive[slot] = 1u; }
                }
                a = ufirst(a);
            }
            amask |= a << g;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    if (threadIdx.x == 0) { flag[0] = 0u; flag[1] = 0u; }
#ifdef AGZ_PSTAMPS
    unsigned long long ps_t = __builtin_amdgcn_s_memtime(), ps_flag = 0, ps_search = 0, ps_cnt = 0, ps_adv = 0;
#define PSTAMP(x) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); x += n_ - ps_t; ps_t = n_; } while (0)
#else
#define PSTAMP(x) do { } while (0)
#endif
#pragma unroll 1
    for (uint32_t it = 0;; ++it) {
        // ---- does the workgroup go on?  Some wave of it still has a game, and the call's own games are not all over
        {
            const PersistPar& Q = par();
            uint32_t* const fw = flag + (it & 1u);
            __syncthreads();                                      // (the word was cleared a whole ply ago / at the entry)
            if (lane == 0) {
                uint32_t f = amask ? 1u : 0u;
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
        }
        PSTAMP(ps_flag);
        // ---- mcts_single (:376-462) for the games of this workgroup: the loop of k_search_small
        EagerCarry C = {1u, 0u, 0u, 0u, 0u, 0u, 0u};
        uint32_t wcount = 0;
        const int V_ = par().S.V;
#pragma unroll 1
        for (int k = 0; k <= V_; ++k) {
            int bx = (int)blockIdx.x;
            asm volatile("" : "+s"(bx));                          // opaque once per rollout (see k_search_small)
            const SmallPar& S = par().S;
            uint8_t* const io_blk = lds_small + S.io_off + (size_t)wave * S.io_bw;
            const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
            rollout_eager_body<FAM, NC, KPL, true, ((G < 8 || WV < 3 || (WV == 3 && KPL <= 16) || KPL <= 4) ? 2 : AGZ_PFM_LOW), true, ROLE_ALL, 0, G>(
                SF, tree_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount, io_blk, S.io_prowb, S.io_lgs, nullptr, ~amask);
            if (k < S.V) {
                __builtin_amdgcn_s_setprio(3);
                const SmallPar& S = par().S;
                mlp_wave_body<H, TW * NG / 16, 2, true, true, (WV < 4), (WV < 3), NWV>(S.F, nn_lds, bx, lds_small + S.io_off, S.io_bw, S.io_lgs);
                __syncthreads();
                __builtin_amdgcn_s_setprio(0);
            }
        }
        // (policy_final of this wave's games was written by lanes of this wave: the root's work item of the last-but-one rollout)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        PSTAMP(ps_search);
        // ---- roofline bookkeeping: descent counters of the search (the lead lane of a game's lane-group holds them)
        {
            const PersistPar& Q = par();
            if (Q.acc) {
                uint32_t ap = (lane % G) == 0 ? C.add_p : 0u, an = (lane % G) == 0 ? C.add_new : 0u;
                for (int o = 32; o > 0; o >>= 1) { ap += (uint32_t)__shfl_xor((int)ap, o, 64); an += (uint32_t)__shfl_xor((int)an, o, 64); }
                if (lane == 0) {
                    atomicAdd(Q.acc + 0, (unsigned long long)ap); atomicAdd(Q.acc + 1, (unsigned long long)an);
                    atomicAdd(Q.acc + 2, (unsigned long long)__builtin_popcount(amask));
                }
            }
        }
        PSTAMP(ps_cnt);
        // ---- the ply step of each game (:513-561), one after the other, the whole wave on one game
        {
            const PersistPar& Q = par();
            const int slot0 = ((int)blockIdx.x * TW + wave) * NG;
            uint32_t next_mask = 0u;
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                if (!((amask >> g) & 1u)) continue;
                next_mask |= advance_slot<FAM, NR, NC, true>(Q.P, slot0 + g) << g;
            }
            amask = next_mask;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the new roots are read by the other lanes of this wave
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        PSTAMP(ps_adv);
    }
#ifdef AGZ_PSTAMPS
    if (lane == 0 && par().acc) { unsigned long long* a = par().acc; atomicAdd(a + 4, ps_flag); atomicAdd(a + 5, ps_search); atomicAdd(a + 6, ps_cnt); atomicAdd(a + 7, ps_adv); }
#endif
    {
        const PersistPar& Q = par();
        if (lane == 0 && Q.acc && amask) atomicAdd(Q.acc + 3, (unsigned long long)__builtin_popcount(amask));
    }
}

// the shapes of AGZ_SMALL_SHAPES in their full-batch form (64-game workgroups of eight tree waves, four waves per SIMD) and the narrow
// form of the few-action games (Connect4: 4 lanes per tree, 16 trees per wave, workgroups of four waves, two waves per SIMD)
#define AGZ_PERSIST_VARIANTS(F, C, K, KW) KW template __global__ void k_selfplay_small<F, C, K, 128, 8, 4>(const PersistPar);
#define AGZ_PERSIST_NARROW_VARIANTS(F, C, K, GG, KW) KW template __global__ void k_selfplay_small<F, C, K, 128, 4, 2, GG>(const PersistPar);
#define AGZ_PERSIST_NARROW_SHAPES(X) X(F_C4, 1, 4, 4)

}  // namespace agz
