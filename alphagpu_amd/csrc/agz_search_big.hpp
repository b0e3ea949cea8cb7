// agz_search_big.hpp — a whole mcts_single (mcts_gpu.jl:376-462) in ONE launch for the WIDE trunks the reference ships
// (ressimplesf(..., 512, 4|6|8), main*.jl:123-128), for batches that leave the chip latency-bound.
//
// With a 512-wide network the two-kernel form (one tree launch + one k_mlp_big launch per rollout) costs ~100 us per rollout
// whatever the batch below ~8192 games: two kernel boundaries and a global barrier per rollout, 37 of the 81 plies of a
// generation.  Here an 8-wave workgroup owns 32 games for the whole search: waves 0-3 run the tree step (rollout_eager_body,
// 8 games each), a workgroup barrier hands the 32 leaves to all eight waves for the forward (mlp_big_body<H, 2>: the 32 x H
// activation tile resident in LDS, weights streamed from L2), a second barrier hands logits and values back.  The tree waves'
// LDS windows lie over the activation tile (the phases alternate); work lists and the per-game carry survive the forward.
// The two bodies are the functions the stand-alone kernels run: same bits (tested).
#pragma once
#include "agz_tree_eager.hpp"
#include "agz_nn_big.hpp"

namespace agz {

struct BigSearchPar {
    TreePar T;                // slot0 = 0, L = number of games
    BigPar B;                 // k_mlp_big's parameters, L = number of games
    int V;                    // rollouts
    int tree_lds;             // bytes of LDS of one tree wave
    int wl_off, wl_bytes;     // the tree waves' work lists live past the window the two phases share: offset, bytes per wave
    int xch_off;              // per tree wave: the carry it publishes for its helper wave (agz_tree_eager.hpp ROLE_*), 144 bytes
    int nxw_off;              // the tree waves' next-word tables (agz_tree_eager.hpp nxw), games per wave x V x 2 bytes each; 0: none
};

// WG = workgroups per CU the register budget is cut for (1: 256 registers, 2: 128)
// KPR: node rows by the root's legal rank (agz_tree_eager.hpp KPR_), 0 = by action
// TWB: tree waves (4: 32 games, the other four waves run the network only and, in the one-workgroup build, take the backup items;
// 8: 64 games, every wave a tree wave, the network body works on four leaf tiles: a layer's weights stream once for 64 games)
template <int FAM, int NC, int KPL, int H, int WG, int KPR = 0, int TWB = 4>
__global__ __launch_bounds__(NB_THREADS, 2 * WG) void k_search_big(const BigSearchPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_bigs[];
    // the parameters are read from the kernel-argument segment per phase (see tree_par(), agz_tree_eager.hpp)
    static_assert(offsetof(BigSearchPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    typedef const BigSearchPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto par = [&]() -> const BigSearchPar& { KArg p = karg; asm volatile("" : "+s"(p)); return *(const BigSearchPar*)p; };
    const BigSearchPar& S = par();
    constexpr int TW = TWB;                                       // tree waves: 8 TW games = the TW / 2 leaf tiles of mlp_big_body<H, TW / 2>
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    uint8_t* const own_lds = lds_bigs + (size_t)wave * S.tree_lds;   // (tree waves 0-3 and their helper waves 4-7 have tables of their own)
    uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_bigs + S.wl_off + (size_t)(wave % TW) * S.wl_bytes);
    uint32_t* const xch = reinterpret_cast<uint32_t*>(lds_bigs + S.xch_off + (size_t)(wave % TW) * 144);
    EagerCarry C = {1u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t wcount = 0;
    // wave priorities: the tree waves (which also take part in the network phase) run above the four network-only waves from their
    // first descent on (rollout_eager_body raises the priority there and nothing lowers it here): 9.36 vs 9.72 ms per ply at 16384 games
    // The network-only waves 4-7 take the BACKUP WORK ITEMS of the tree waves 0-3 while those expand the leaf (the two halves of a
    // rollout's first phase are independent: agz_tree_eager.hpp ROLE_*); a workgroup barrier joins them before the descent.
    // (one workgroup per CU only: with two the helpers compete with the other workgroup's waves, 8.43 vs 8.23 ms per ply at 16384 games)
    constexpr int PF_ = WG < 2 ? 2 : 1;
    constexpr bool SPLIT = WG < 2 && TW == 4;
    const int V_ = S.V;
#ifdef AGZ_BIGSTAMPS
    unsigned long long st_tree = 0, st_wait = 0, st_net = 0, st_join = 0;   // cycles of wave 0: tree step, barrier in front of the pass, network pass, barrier behind it
#endif
    for (int k = 0; k <= V_; ++k) {
#ifdef AGZ_BIGSTAMPS
        const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#endif
        int bx = (int)blockIdx.x;
        asm volatile("" : "+s"(bx));                              // (see k_search_small)
        const BigSearchPar& S = par();
        uint16_t* const nxw = S.nxw_off ? reinterpret_cast<uint16_t*>(lds_bigs + S.nxw_off) + (size_t)(wave % TW) * (size_t)(8 * S.V) : nullptr;   // (a helper wave: its tree wave's)
        if constexpr (!SPLIT) {
            const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
            if (wave < TW) rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_ALL, KPR>(SF, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                                            nullptr, 0, 0, nullptr, 0u, 0u, nxw);
        }
        if (SPLIT && k > 0) {
            const StepFlags SE = {(uint32_t)k, 0, 1, 0, k == S.V - 1, k == S.V};
            if (wave < TW) rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_EXPAND, KPR>(SE, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                                        nullptr, 0, 0, xch, 0u, 0u, nxw);
            else rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_ITEMS, KPR>(SE, own_lds, bx * TW + wave % TW, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                              nullptr, 0, 0, xch, 0u, 0u, nxw);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                                      // the leaf is expanded, the path's rows and next words are rebuilt
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (k < S.V) {
            if (SPLIT && wave < TW) {
                const StepFlags SS = {(uint32_t)k, k == 0, 0, 1, k == S.V - 1, 0};
                rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_EXPAND, KPR>(SS, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                             nullptr, 0, 0, xch, 0u, 0u, nxw);
            }
#ifdef AGZ_BIGSTAMPS
            const unsigned long long c1 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                                      // the planes of the 32 leaves are written
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AGZ_BIGSTAMPS
            const unsigned long long c2 = __builtin_amdgcn_s_memtime();
#endif
            const BigSearchPar& S = par();
            const int gpw = S.T.gpw, L = S.T.L;
            mlp_big_body<H, TW / 2, (WG < 2)>(S.B, lds_bigs, [&](int row) { return (row & 7) < gpw ? (bx * TW + (row >> 3)) * gpw + (row & 7) : L; });
#ifdef AGZ_BIGSTAMPS
            const unsigned long long c3 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                                      // logits and values are visible to the tree waves
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef AGZ_BIGSTAMPS
            st_tree += c1 - c0; st_wait += c2 - c1; st_net += c3 - c2; st_join += __builtin_amdgcn_s_memtime() - c3;
#endif
        }
    }
#ifdef AGZ_BIGSTAMPS
    if (threadIdx.x == 0 && S.T.dbg) {
        unsigned long long* d = S.T.dbg + (size_t)(32768 + (blockIdx.x & 32767)) * 16;
        d[0] += st_tree; d[1] += st_wait; d[2] += st_net; d[3] += st_join; d[4] += 1;
    }
#endif
}

// ONE 128-game workgroup per CU (round 5; the search form of k_selfplay_big4, agz_selfplay_big.hpp): eight waves of sixteen trees on 4 lanes
// each (KPL4 = twice the 8-lane block: the records keep their layout), the network pass on 128 leaves — a layer's weights stream from L2 once
// per 128 leaves, four k-rows of fragments in flight — 256 registers.  Above 64 games per CU, where the 8-lane form runs two 64-game
// workgroups per CU at 128 registers.  KPR4: node rows by the root's legal rank (twice the 8-lane KPR), 0 = by action.
template <int FAM, int NC, int KPL4, int H, int KPR4 = 0>
__global__ __launch_bounds__(NB_THREADS, 2) void k_search_big4(const BigSearchPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_bigs[];
    static_assert(offsetof(BigSearchPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    constexpr int TW = 8, G = 4, NG = 64 / G;
    static_assert(TW * NG == NB_M, "the workgroup's games are the 128 rows of the activation tile");
    typedef const BigSearchPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto par = [&]() -> const BigSearchPar& { KArg p = karg; asm volatile("" : "+s"(p)); return *(const BigSearchPar*)p; };
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    EagerCarry C = {1u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t wcount = 0;
    const int V_ = par().V;
#pragma unroll 1
    for (int k = 0; k <= V_; ++k) {
        int bx = (int)blockIdx.x;
        asm volatile("" : "+s"(bx));                              // (see k_search_small)
        {
            const BigSearchPar& S = par();
            uint8_t* const own_lds = lds_bigs + (size_t)wave * S.tree_lds;
            uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_bigs + S.wl_off + (size_t)wave * S.wl_bytes);
            uint16_t* const nxw = S.nxw_off ? reinterpret_cast<uint16_t*>(lds_bigs + S.nxw_off) + (size_t)wave * (size_t)(NG * S.V) : nullptr;
            const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
            rollout_eager_body<FAM, NC, KPL4, true, 2, false, ROLE_ALL, KPR4, G>(SF, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                         nullptr, 0, 0, nullptr, 0u, 0u, nxw);
        }
        if (k < V_) {
            // (the barrier that publishes the planes of the 128 leaves is taken inside the pass, behind its first weight requests: PREB)
            const BigSearchPar& S = par();
            const int L = S.T.L;
            mlp_big_body<H, NB_M / 16, true, true>(S.B, lds_bigs, [&](int row) { return bx * NB_M + row < L ? bx * NB_M + row : L; });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();                                      // logits and values are visible to the tree waves
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    }
}
// (family, chunks, actions per lane of the 4-lane form, rows per lane by legal rank): the shapes of BASELINE configs 3-5
#define AGZ_BIG4_SHAPES(X) X(F_LINE, 2, 24, 0) X(F_LINE, 2, 24, 16) X(F_LINE, 2, 24, 8) X(F_HEX, 2, 24, 0) X(F_HEX, 2, 24, 16) X(F_HEX, 2, 24, 8) X(F_REV, 1, 24, 0) AGZ_BIG4_SHAPES_MORE(X)
// ... and the other built-in shapes of up to 96 actions, rows by action
#define AGZ_BIG4_SHAPES_MORE(X) X(F_LINE, 1, 8, 0) X(F_LINE, 1, 16, 0) X(F_C4, 1, 8, 0) X(F_HEX, 1, 8, 0) X(F_HEX, 1, 16, 0) X(F_HEX, 2, 16, 0) X(F_REV, 1, 16, 0)
#define AGZ_BIG4_VARIANTS(F, C, K4, R4, KW) KW template __global__ void k_search_big4<F, C, K4, 512, R4>(const BigSearchPar);

#define AGZ_BIG_VARIANTS(F, C, K, KW)                                        \
    KW template __global__ void k_search_big<F, C, K, 512, 1>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 2>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 1, 0, 8>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 2, 0, 8>(const BigSearchPar);
#define AGZ_BIG_CMP_VARIANTS(F, C, K, R, KW)                                 \
    KW template __global__ void k_search_big<F, C, K, 512, 1, R>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 2, R>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 1, R, 8>(const BigSearchPar); \
    KW template __global__ void k_search_big<F, C, K, 512, 2, R, 8>(const BigSearchPar);

}  // namespace agz
