// agz_selfplay_big.hpp — the persistent self-play kernel (agz_selfplay_small.hpp: one launch per agz_selfplay / agz_selfplay_chain call, a
// workgroup loops over the plies of its own 64 games: search -> root policy -> move choice -> play / isOver -> sample capture -> next game)
// for the 512-wide trunks of BASELINE configs 3-5: the search is the rollout loop of k_search_big with 64-game workgroups (eight tree
// waves, the network pass on four leaf tiles: agz_search_big.hpp TWB = 8), one or two workgroups per CU.
#pragma once
#include "agz_search_big.hpp"
#include "agz_selfplay_small.hpp"

namespace agz {

struct PersistBigPar { BigSearchPar S; PersistTail X; };

// WG = workgroups per CU the register budget is cut for (1: up to 64 slots per CU, 2: up to 128)
template <int FAM, int NC, int KPL, int H, int WG>
__global__ __launch_bounds__(NB_THREADS, 2 * WG) void k_selfplay_big(const PersistBigPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_bigs[];
    static_assert(offsetof(PersistBigPar, S) == 0 && offsetof(BigSearchPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    static_assert(NB_THREADS == 512, "eight waves, every one a tree wave");
    constexpr int TW = 8;
    typedef const PersistBigPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto tail = [=]() -> const PersistTail& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistBigPar*)p)->X; };
    persist_loop<FAM, NC, KPL, 8, TW, false, false>(lds_bigs, tail, [&](const uint32_t amask, const bool, EagerCarry& C) {
        const auto spar = [=]() -> const BigSearchPar& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistBigPar*)p)->S; };
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        constexpr int PF_ = WG < 2 ? 2 : 1;
        uint32_t wcount = 0;
        const int V_ = spar().V;
#pragma unroll 1
        for (int k = 0; k <= V_; ++k) {
            int bx = (int)blockIdx.x;
            asm volatile("" : "+s"(bx));                          // (see k_search_small)
            {
                const BigSearchPar& S = spar();
                uint8_t* const own_lds = lds_bigs + (size_t)wave * S.tree_lds;
                uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_bigs + S.wl_off + (size_t)wave * S.wl_bytes);
                uint16_t* const nxw = S.nxw_off ? reinterpret_cast<uint16_t*>(lds_bigs + S.nxw_off) + (size_t)wave * (size_t)(8 * S.V) : nullptr;
                const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
                rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_ALL, 0>(SF, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                         nullptr, 0, 0, nullptr, ~amask, 0u, nxw);
            }
            if (k < V_) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();                                  // the planes of the 64 leaves are written
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const BigSearchPar& S = spar();
                const int L = S.T.L;
                mlp_big_body<H, TW / 2, (WG < 2)>(S.B, lds_bigs, [&](int row) { return bx * 64 + row < L ? bx * 64 + row : L; });
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();                                  // logits and values are visible to the tree waves
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
    });
}

// ONE 128-game workgroup per CU (round 5): eight waves of SIXTEEN trees, 4 lanes per tree (agz_tree_eager.hpp G_ = 4: KPL4 = 2 KPL actions per
// lane — the node records keep their layout, 4 KPL4 = 8 KPL entries per row), 256 registers.  What it buys the wide trunks: a layer's
// weight fragments stream from L2 ONCE PER 128 LEAVES (mlp_big_body<H, 8>: 69 -> 34 KB of L2 reads per leaf — the two 64-game workgroups
// of a CU pull 2 x 4.4 MB per rollout through an L2 path that delivers ~66-73 GB/s per CU) and FOUR k-rows of fragments in flight
// instead of the two that 128 registers hold (the 64-leaf pass of the two-workgroup build is bound by bytes in flight: 64 KB per
// workgroup against ~1.5 us of L2 latency under load = 43 GB/s, what it measures).  The price: nothing overlaps the tree step any more —
// which the 4-lane tree step makes a third shorter in vector instructions (the per-round fixed work is shared by twice the items).
template <int FAM, int NC, int KPL4, int H>
__global__ __launch_bounds__(NB_THREADS, 2) void k_selfplay_big4(const PersistBigPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_bigs[];
    static_assert(offsetof(PersistBigPar, S) == 0 && offsetof(BigSearchPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    constexpr int TW = 8, G = 4, NG = 64 / G;
    static_assert(TW * NG == NB_M, "the workgroup's games are the 128 rows of the activation tile");
    typedef const PersistBigPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto tail = [=]() -> const PersistTail& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistBigPar*)p)->X; };
    persist_loop<FAM, NC, KPL4, G, TW, false, true>(lds_bigs, tail, [&](const uint32_t amask, const bool, EagerCarry& C) {
        const auto spar = [=]() -> const BigSearchPar& { KArg p = karg; asm volatile("" : "+s"(p)); return ((const PersistBigPar*)p)->S; };
        const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        uint32_t wcount = 0;
        const int V_ = spar().V;
#ifdef AGZ_BIG4STAMPS
        unsigned long long st_[4] = {0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define B4STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define B4STAMP(i) do { } while (0)
#endif
#pragma unroll 1
        for (int k = 0; k <= V_; ++k) {
            int bx = (int)blockIdx.x;
            asm volatile("" : "+s"(bx));                          // (see k_search_small)
            {
                const BigSearchPar& S = spar();
                uint8_t* const own_lds = lds_bigs + (size_t)wave * S.tree_lds;
                uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_bigs + S.wl_off + (size_t)wave * S.wl_bytes);
                uint16_t* const nxw = S.nxw_off ? reinterpret_cast<uint16_t*>(lds_bigs + S.nxw_off) + (size_t)wave * (size_t)(NG * S.V) : nullptr;
                const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
                rollout_eager_body<FAM, NC, KPL4, true, 2, false, ROLE_ALL, 0, G>(SF, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                          nullptr, 0, 0, nullptr, ~amask, 0u, nxw);
            }
            B4STAMP(0);
            if (k < V_) {
                // (the barrier that publishes the planes of the 128 leaves is taken inside the pass, behind its first weight requests: PREB)
                B4STAMP(1);
                const BigSearchPar& S = spar();
                const int L = S.T.L;
#ifdef AGZ_BIG4STAMPS
                mlp_big_body<H, NB_M / 16, true, true>(S.B, lds_bigs, [&](int row) { return bx * NB_M + row < L ? bx * NB_M + row : L; }, tail().acc ? tail().acc + 6 : nullptr);
#else
                mlp_big_body<H, NB_M / 16, true, true>(S.B, lds_bigs, [&](int row) { return bx * NB_M + row < L ? bx * NB_M + row : L; });
#endif
                B4STAMP(2);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();                                  // logits and values are visible to the tree waves
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                B4STAMP(3);
            }
        }
#ifdef AGZ_BIG4STAMPS
        // cycles of every wave: tree step, barrier in front of the pass, network pass, barrier behind it (summed into acc[12..15])
        if ((threadIdx.x & 63) == 0 && tail().acc) for (int i = 0; i < 4; ++i) atomicAdd(tail().acc + 12 + i, st_[i]);
#endif
#undef B4STAMP
    });
}
// (family, chunks, actions per lane of the 4-lane form = twice the 8-lane block): the shapes of BASELINE configs 3-5, then every other built-in shape of up to
// 96 actions (Connect4, Reversi 6x6, the smaller Gobang and Hex boards)
#define AGZ_PERSIST_BIG4_SHAPES(X) X(F_LINE, 2, 24) X(F_HEX, 2, 24) X(F_REV, 1, 24) AGZ_PERSIST_BIG4_SHAPES_MORE(X)
#define AGZ_PERSIST_BIG4_SHAPES_MORE(X) X(F_LINE, 1, 8) X(F_LINE, 1, 16) X(F_C4, 1, 8) X(F_HEX, 1, 8) X(F_HEX, 1, 16) X(F_HEX, 2, 16) X(F_REV, 1, 16)
#define AGZ_PERSIST_BIG4_VARIANTS(F, C, K4, KW) KW template __global__ void k_selfplay_big4<F, C, K4, 512>(const PersistBigPar);

#define AGZ_PERSIST_BIG_VARIANTS(F, C, K, KW)                                \
    KW template __global__ void k_selfplay_big<F, C, K, 512, 1>(const PersistBigPar); \
    KW template __global__ void k_selfplay_big<F, C, K, 512, 2>(const PersistBigPar);

}  // namespace agz
