// agz_search_small.hpp — a whole mcts_single (mcts_gpu.jl:376-462: V x {select, network, expand, backup}) in ONE launch, for
// 128-wide trunks and every batch that fits the chip at once (up to 128 games per CU).
//
// One tree kernel and one network kernel per rollout put a global barrier after every rollout and two kernel boundaries on the
// critical path of every game.  Here a 4-wave workgroup owns its games for the whole search: TW of its waves run the eager-policy
// tree step (rollout_eager_body, up to 8 games each), a workgroup barrier hands the leaves to all four waves for the network
// forward (mlp_wave_body; planes and logits change hands through LDS), a second barrier hands logits and values back.
// TW = 2 (16 games per workgroup; its two network-only waves take the backup work items) up to 8192 games, TW = 4 (32 games) above; up to 64 games per CU the waves are sparse (the
// fewest games per wave that keep every workgroup resident; lane-groups without a game take work items of the wave's games).
// The two bodies are the very functions the stand-alone kernels run — same arithmetic, same bits (tested).
#pragma once
#include "agz_tree_eager.hpp"
#include "agz_nn_wave.hpp"

#ifndef AGZ_PERSIST_NXL
#define AGZ_PERSIST_NXL 1    // 1: the descent of the whole-search kernels follows next words kept in LDS (agz_tree_eager.hpp nxw) wherever they fit; 0: the records' (A/B)
#endif
#ifndef AGZ_PFM_LOW
#define AGZ_PFM_LOW 1       // item prefetch of the builds without register room (agz_tree_eager.hpp PFM): 1 = touch only, 3 = first part into registers
#endif

namespace agz {

struct SmallPar {
    TreePar T;                // slot0 = 0, L = number of games; rollout / do_* / last are set per rollout in the kernel
    Fused3Par F;              // the uniform weight tiling of agz_nn_wave.hpp, L = number of games
    int V;                    // rollouts
    int tree_lds;             // bytes of LDS of one tree wave
    int wl_off, wl_bytes;     // the tree waves' work lists live past the window the two phases share: offset, bytes per wave
    int io_off, io_bw;        // hand-over window (planes to the network, logits back): offset, bytes per tree wave (one row per game of a full wave)
    int io_prowb, io_lgs;     // ... bytes / floats of a row (the same row carries the leaf's planes to the network and its logits back)
    int xch_off;              // 16-game workgroups: per tree wave the carry it publishes for its helper wave (4 NG + 1 words in 16 NG + 16 bytes)
    int nxw_off;              // the tree waves' next-word tables (agz_tree_eager.hpp nxw), NG x V x 2 bytes per wave; 0: none (the descent reads the records)
};

// TW = tree waves per workgroup (2 or 4): the workgroup owns 8*TW games and runs the network with TW/2 leaf tiles.
// WV = waves per SIMD the register budget is cut for (2: 172 VGPRs, no spills; 3, 4: more workgroups per CU so that 24576 /
// 32768 games are resident at once with 32 games per workgroup).
// KPR: entries per lane of the node rows when they are indexed by the root's legal rank (agz_tree_eager.hpp KPR_), 0 = by action
// G: lanes per game tree (agz_tree_eager.hpp G_): a tree wave walks NG = 64 / G trees, the workgroup owns TW * NG games and the network
// body runs TW * NG / 16 leaf tiles.  G = 4 (16 trees per wave, 24 actions per lane on a 9x9 board): two 64-game workgroups of four
// waves per CU hold 32768 games with TWO waves per SIMD and 256 registers each — half the wave-instructions of the item loop's
// per-round fixed work per game.  G = 2: Connect4's 7 actions in 2 x 4 slots, 32 trees per wave.
// GPW_ (round 6): games per tree wave when that is fewer than its 64 / G lane-groups BY CONSTRUCTION (0: all of them; fewer at run time — T.gpw — is
// the sparse-wave dispatch of the small batches): a few-action game on 4-lane groups, eight games per wave of sixteen groups, four waves per SIMD
// (agz_selfplay_small.hpp GPW_); the network then works on TW x GPW_ rows
template <int FAM, int NC, int KPL, int H, int TW, int WV, int KPR = 0, int G = 8, int GPW_ = 0>
__global__ __launch_bounds__(64 * (TW == 8 ? 8 : NW_WAVES), WV) void k_search_small(const SmallPar) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_small[];
    // The parameters are READ FROM THE KERNEL-ARGUMENT SEGMENT where they are needed (scalar loads), through a pointer made opaque
    // once per phase: held in scalar registers from the entry on they (some 90 words) push the lane masks and addresses of the
    // tree step out into spill lanes, and every v_readlane / v_writelane of a spill is a 4-cycle vector instruction of a kernel
    // that is bound by vector issue.
    static_assert(offsetof(SmallPar, T) == 0, "rollout_eager_body reads its TreePar from the start of the argument segment");
    typedef const SmallPar __attribute__((address_space(4)))* KArg;
    const KArg karg = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    const auto par = [&]() -> const SmallPar& { KArg p = karg; asm volatile("" : "+s"(p)); return *(const SmallPar*)p; };
    const SmallPar& S = par();
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // (scalar: the tree step's addresses are scalar base + offset)
    static_assert(TW == 2 || TW == 4 || TW == 8, "tree waves per workgroup");
    constexpr int NWV = TW == 8 ? 8 : NW_WAVES;                   // waves of the workgroup (TW = 8: 64 games, every wave a tree wave)
    uint8_t* const tree_lds = lds_small + (size_t)(wave % TW) * S.tree_lds;
    uint8_t* const nn_lds = lds_small;                            // the two phases never overlap and the tree step keeps nothing
                                                                  // in this window from one rollout to the next: same memory
    uint32_t* const wl_lds = reinterpret_cast<uint32_t*>(lds_small + S.wl_off + (size_t)(wave % TW) * S.wl_bytes);
    EagerCarry C = {1u, 0u, 0u, 0u, 0u, 0u, 0u};                      // what a game carries from rollout to rollout: in registers
    uint32_t wcount = 0;
    // 16-game workgroups (TW = 2: the small batches of a generation's tail, where a rollout's dependent chain is what a ply
    // costs): waves 2 and 3 would idle through the tree phase, so they take the BACKUP WORK ITEMS of the tree waves 0 and 1 while
    // those expand the leaf — the two halves of a rollout's first phase are independent (agz_tree_eager.hpp ROLE_*); a workgroup
    // barrier joins them before the descent.
    // (item prefetch: into registers wherever they are free — up to 3 waves per SIMD, and rows of 4 actions per lane in the 128-register
    //  build: Connect4 85.8 -> 85.0 ms per generation; rows of 8 gain nothing or spill; rows of 24 actions spill 60 registers with it
    //  at 3 waves per SIMD and 4 without: Gobang 13x13 at 24576 games 9.2 -> 7.7 ms per ply)
    constexpr int NG = 64 / G, GPW = GPW_ ? GPW_ : NG;
    constexpr bool SPLIT = TW == 2 && G == 8;
    uint8_t* const own_lds = lds_small + (size_t)(SPLIT ? wave : wave % TW) * S.tree_lds;   // (a helper wave has tables of its own)
    uint32_t* const xch = reinterpret_cast<uint32_t*>(lds_small + S.xch_off + (size_t)(wave % TW) * (16 * NG + 16));
    const int V_ = S.V;
#ifdef AGZ_WGTIME
    const unsigned long long wg_t0 = wall_clock64();
#endif
    for (int k = 0; k <= V_; ++k) {
        // the workgroup index is made opaque once per rollout: otherwise every per-game address of both bodies is hoisted out
        // of this loop and kept alive across them (hundreds of registers, spills)
        int bx = (int)blockIdx.x;
        asm volatile("" : "+s"(bx));
        const SmallPar& S = par();
        uint8_t* const io_blk = lds_small + S.io_off + (size_t)(wave % TW) * S.io_bw;
        uint16_t* const nxw = S.nxw_off ? reinterpret_cast<uint16_t*>(lds_small + S.nxw_off) + (size_t)(wave % TW) * (size_t)(NG * S.V) : nullptr;   // (a helper wave: its tree wave's)
        if constexpr (SPLIT) {
            if (k > 0) {
                const StepFlags SE = {(uint32_t)k, 0, 1, 0, k == S.V - 1, k == S.V};
                if (wave < TW) rollout_eager_body<FAM, NC, KPL, true, 2, true, ROLE_EXPAND, KPR>(SE, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                                         io_blk, S.io_prowb, S.io_lgs, xch, 0u, 0u, nxw);
                else rollout_eager_body<FAM, NC, KPL, true, 2, true, ROLE_ITEMS, KPR>(SE, own_lds, bx * TW + wave % TW, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                               io_blk, S.io_prowb, S.io_lgs, xch, 0u, 0u, nxw);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();                                  // the leaf is expanded, the path's rows and next words are rebuilt
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            if (k < S.V && wave < TW) {
                const StepFlags SS = {(uint32_t)k, k == 0, 0, 1, k == S.V - 1, 0};
                rollout_eager_body<FAM, NC, KPL, true, 2, true, ROLE_EXPAND, KPR>(SS, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                            io_blk, S.io_prowb, S.io_lgs, xch, 0u, 0u, nxw);
            }
        } else {
            const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
            if (wave < TW) rollout_eager_body<FAM, NC, KPL, true, ((G < 8 || WV < 3 || (WV == 3 && KPL <= 16) || KPL <= 4) ? 2 : AGZ_PFM_LOW), true, ROLE_ALL, KPR, G>(SF, tree_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                                   io_blk, S.io_prowb, S.io_lgs, nullptr, 0u, 0u, nxw);
        }
#ifdef AGZ_STAMPS
        const unsigned long long t_nn0 = __builtin_amdgcn_s_memtime();
#endif
        if (k < S.V) {
            // (the barrier that publishes the planes of the leaves sits inside, after the first weight fragments are requested)
            // wave priorities: the phases that wait on latencies with little arithmetic (network: barriers, LDS, MFMA chain; descent
            // and child creation in the tree step: dependent loads) go first, so that the waves get back to the arithmetic of the
            // work items sooner (measured -1.6 % per generation)
            __builtin_amdgcn_s_setprio(3);
            const SmallPar& S = par();
#ifdef AGZ_STAMPS
            mlp_wave_body<H, TW * GPW / 16, 2, true, true, (WV < 4), (WV < 3), NWV>(S.F, nn_lds, bx, lds_small + S.io_off, S.io_bw, S.io_lgs,
                                                              S.T.dbg ? S.T.dbg + (size_t)(32768 + bx * NWV + wave) * 16 : nullptr);
#else
            mlp_wave_body<H, TW * GPW / 16, 2, true, true, (WV < 4), (WV < 3), NWV>(S.F, nn_lds, bx, lds_small + S.io_off, S.io_bw, S.io_lgs);
#endif
            __syncthreads();                                      // logits and values are visible to the tree waves
            __builtin_amdgcn_s_setprio(0);
        }
#ifdef AGZ_STAMPS
        if ((threadIdx.x & 63) == 0 && wave < TW && S.T.dbg) S.T.dbg[(size_t)(bx * TW + wave) * 16 + 15] += __builtin_amdgcn_s_memtime() - t_nn0;
#endif
    }
#ifdef AGZ_WGTIME
    // diagnostic: when did this workgroup start and end (100 MHz clock), per launch (step & 255) and workgroup
    if (threadIdx.x == 0 && par().T.dbg && blockIdx.x < 512) {
        unsigned long long* d = par().T.dbg + ((size_t)(par().T.step & 255u) * 512 + blockIdx.x) * 2;
        d[0] = wg_t0; d[1] = wall_clock64();
    }
#endif
}

// The instantiated shapes (game family, bitboard chunks, actions per lane): agz_small_inst.hip compiles them in four
// parts in parallel, agz_engine.hip only declares them.
#define AGZ_SMALL_SHAPES_0(X) X(F_LINE, 1, 4) X(F_LINE, 1, 8) X(F_LINE, 2, 12) X(F_LINE, 2, 16)
#define AGZ_SMALL_SHAPES_1(X) X(F_LINE, 3, 24) X(F_C4, 1, 4) X(F_HEX, 1, 4) X(F_HEX, 1, 8)
#define AGZ_SMALL_SHAPES_2(X) X(F_HEX, 2, 8) X(F_HEX, 2, 12) X(F_HEX, 2, 16) X(F_HEX, 3, 16)
#define AGZ_SMALL_SHAPES_3(X) X(F_HEX, 3, 24) X(F_REV, 1, 12) X(F_REV, 1, 8) AGZ_EXTRA_SHAPES(X)   // (+ the shapes of a plugged-in game, agz_games.hpp K_EXTRA)
#define AGZ_SMALL_SHAPES(X) AGZ_SMALL_SHAPES_0(X) AGZ_SMALL_SHAPES_1(X) AGZ_SMALL_SHAPES_2(X) AGZ_SMALL_SHAPES_3(X)
#define AGZ_SMALL_VARIANTS(F, C, K, KW)                                      \
    KW template __global__ void k_search_small<F, C, K, 128, 2, 2>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 2>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 3>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 4>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 8, 4>(const SmallPar);
// rows by the root's legal rank (12 -> 8 entries per lane: 9x9 boards from ply 17 on, -> 4 from ply 49 on); parts 4-6 of agz_small_inst.hip
#define AGZ_SMALL_CMP_SHAPES_4(X) X(F_LINE, 2, 12, 8) X(F_HEX, 2, 12, 8) X(F_LINE, 2, 12, 4) X(F_HEX, 2, 12, 4)
// 11x11 (16 actions per lane -> 12 / 8 / 4) and 13x13 (24 -> 16 / 8 / 4): parts 5 and 6
#define AGZ_SMALL_CMP_SHAPES_5(X) X(F_LINE, 2, 16, 12) X(F_LINE, 2, 16, 8) X(F_LINE, 2, 16, 4) X(F_LINE, 3, 24, 16) X(F_LINE, 3, 24, 8) X(F_LINE, 3, 24, 4)
#define AGZ_SMALL_CMP_SHAPES_6(X) X(F_HEX, 2, 16, 12) X(F_HEX, 2, 16, 8) X(F_HEX, 2, 16, 4) X(F_HEX, 3, 16, 12) X(F_HEX, 3, 16, 8) X(F_HEX, 3, 16, 4)
#define AGZ_SMALL_CMP_SHAPES(X) AGZ_SMALL_CMP_SHAPES_4(X) AGZ_SMALL_CMP_SHAPES_5(X) AGZ_SMALL_CMP_SHAPES_6(X)
#define AGZ_SMALL_CMP_VARIANTS(F, C, K, R, KW)                               \
    KW template __global__ void k_search_small<F, C, K, 128, 2, 2, R>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 2, R>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 3, R>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 4, R>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 8, 4, R>(const SmallPar);

// narrow lane-groups (G = 4: 16 trees per wave, G = 2: 32): (family, chunks, actions per lane, rows per lane by legal rank or 0, lanes per tree);
// part 7 of agz_small_inst.hip.  Workgroups of four tree waves; register budgets for 2 and 1 waves per SIMD.
#define AGZ_SMALL_NARROW_SHAPES(X) X(F_LINE, 2, 24, 0, 4) X(F_LINE, 2, 24, 16, 4) X(F_LINE, 2, 24, 8, 4) X(F_C4, 1, 4, 0, 4) X(F_C4, 1, 4, 0, 2)
#define AGZ_SMALL_NARROW_VARIANTS(F, C, K, R, GG, KW)                        \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 2, R, GG>(const SmallPar); \
    KW template __global__ void k_search_small<F, C, K, 128, 4, 1, R, GG>(const SmallPar);
// ... and the few-action shapes with SPARSE waves (half of the lane-groups hold a game, four waves per SIMD in 128 registers)
#define AGZ_SMALL_NARROW_SPARSE_SHAPES(X) X(F_C4, 1, 4, 0, 4)
#define AGZ_SMALL_NARROW_SPARSE_VARIANTS(F, C, K, R, GG, KW) KW template __global__ void k_search_small<F, C, K, 128, 4, 4, R, GG, 32 / GG>(const SmallPar);

}  // namespace agz
