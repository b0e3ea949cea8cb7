// agz_search_small.hpp — a whole mcts_single (mcts_gpu.jl:376-462: V x {select, network, expand, backup}) in ONE launch,
// for the long tail of a generation where few games are alive.
//
// With < ~2000 games every launch is bound by its own dependency chain, and a rollout pays two kernel boundaries
// (dispatch, ramp-up, drain, the round trip of planes / logits through L2).  Here a 4-wave workgroup owns 16 games for the
// whole search: waves 0-1 run the register-row tree step (rollout_reg_body, 8 games each), a workgroup barrier hands the
// 16 leaves to all four waves for the network forward (mlp_wave_body), a second barrier hands logits and values back.
// The two bodies are the very functions the stand-alone kernels run — same arithmetic, same bits (tested).
#pragma once
#include "agz_tree_reg.hpp"
#include "agz_nn_wave.hpp"

namespace agz {

struct SmallPar {
    TreePar T;                // slot0 = 0, L = number of games; rollout / do_* / last are set per rollout in the kernel
    Fused3Par F;              // the uniform weight tiling of agz_nn_wave.hpp, L = number of games
    int V;                    // rollouts
    int tree_lds;             // bytes of LDS of one tree wave
};

#ifndef AGZ_SMALL_WAVES
#define AGZ_SMALL_WAVES 2
#endif
// TW = tree waves per workgroup (2 or 4): the workgroup owns 8*TW games and runs the network with TW/2 leaf tiles.
template <int FAM, int NC, int KPL, int H, int TW>
__global__ __launch_bounds__(64 * NW_WAVES, AGZ_SMALL_WAVES) void k_search_small(const SmallPar S) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_small[];
    const int wave = (int)threadIdx.x >> 6;
    static_assert(TW == 2 || TW == 4, "tree waves per workgroup");
    uint8_t* const tree_lds = lds_small + (size_t)(wave % TW) * S.tree_lds;
    uint8_t* const nn_lds = lds_small + (size_t)TW * S.tree_lds;
    for (int k = 0; k <= S.V; ++k) {
        const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1};
        // the workgroup index is made opaque once per rollout: otherwise every per-game address of both bodies is hoisted out
        // of this loop and kept alive across them (hundreds of registers, spills)
        int bx = (int)blockIdx.x;
        asm volatile("" : "+s"(bx));
        if (wave < TW) rollout_reg_body<FAM, NC, 8, KPL>(S.T, SF, tree_lds, bx * TW + wave);
        if (k < S.V) {
            // (the barrier that publishes the planes of the leaves sits inside, after the first weight fragments are requested)
            mlp_wave_body<H, TW / 2, 2, true>(S.F, nn_lds, bx);
            __syncthreads();                                      // logits and values are visible to the tree waves
        }
    }
}

}  // namespace agz
