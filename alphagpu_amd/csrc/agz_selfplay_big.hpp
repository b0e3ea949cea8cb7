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
                const StepFlags SF = {(uint32_t)k, k == 0, k > 0, k < S.V, k == S.V - 1, k == S.V};
                rollout_eager_body<FAM, NC, KPL, true, PF_, false, ROLE_ALL, 0>(SF, own_lds, bx * TW + wave, C, wl_lds, (uint32_t)S.wl_bytes >> 2, wcount,
                                                                         nullptr, 0, 0, nullptr, ~amask);
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

#define AGZ_PERSIST_BIG_VARIANTS(F, C, K, KW)                                \
    KW template __global__ void k_selfplay_big<F, C, K, 512, 1>(const PersistBigPar); \
    KW template __global__ void k_selfplay_big<F, C, K, 512, 2>(const PersistBigPar);

}  // namespace agz
