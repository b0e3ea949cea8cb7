// agz_nn_wave.hpp — the snetwork2 forward (DenseNet.jl:294-304), latency-first: one 4-wave workgroup per 16 leaves.
//
// A layout that stages the weights through LDS (tried first, removed) spends its time waiting: every layer is "weights chunk
// arrives -> commit to LDS -> barrier -> 32 MFMAs -> epilogue -> barrier", ~7.6 K cycles per layer for ~0.5 K cycles of matrix
// work, and a search with few games left (the long tail of a self-play generation) pays that latency on every rollout.
// Here the pre-tiled weight fragments (1 KiB per 16x32 tile, lane l's 16 bytes at +16 l) go from L2 straight into the MFMA
// A operand.  Wave w of the workgroup owns the neuron tiles {w*TPW .. w*TPW+TPW-1} of every layer, so it streams only a
// quarter of the weights and can keep NW_DEPTH = 4 whole layers of its fragments in flight in registers (weights do not
// depend on activations: the prefetch runs across layer boundaries and the matrix core never waits for memory).
// The activations of the 16 leaves ping-pong between two 4 KiB LDS strips, one workgroup barrier per layer.
// Same MFMA instruction, operand order and k order as the per-layer kernels (agz_nn.hpp) -> bit-identical logits and values.
//
// P.w16 is the UNIFORM tiling built by agz_set_network: the network is a sequence of GROUPS of KTH = H/32 k-rows of
// NTH = H/16 tiles: layer 0 = G0 = ceil((INP/32)/KTH) groups (zero rows pad it), the T residual layers one group each,
// zero-weight residual groups (the identity on post-ReLU activations: relu(b + relu(0)) = b) up to a multiple of
// NW_DEPTH, the head (zero tiles pad it to NTH), then NW_DEPTH groups of slack so that the prefetch needs no bounds test.
#pragma once
#include "agz_nn.hpp"

namespace agz {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define AGZ_GLB __attribute__((address_space(1)))

// arguments of the fused forward kernels (k_mlp_wave, and the network phase of k_search_small)
struct Fused3Par {
    const uint16_t* planes; int INP;      // [L][INP] bf16, INP % 32 == 0
    const uint16_t* w16;                  // [layer 0 | T residual layers | head] fragments (16x16x32 tiling)
    const float* bias_head;
    float* logits; int LGS; float* vout;
    int L, T, A, AOP;
    int gpw, tw;                          // k_search_small with sparse waves: row r of a workgroup's tile is game slot
                                          // (bidx*tw + r/rb)*gpw + r%rb if r%rb < gpw (gpw = 0: rows are consecutive leaves)
    int rb;                               // ... rb = games of a full tree wave = rows of its block of the hand-over window (8, 16 or 32)
};

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __attribute__((noinline)) float sigmoid_ool(float x) { return sigmoid_spec(x); }
// relu of an MFMA accumulator element (x > 0 ? x : 0 for every non-NaN x, -0 included) as ONE integer instruction: the bits of a
// non-negative float order like the integer, those of a negative one are a negative integer.  (A NaN with a clear sign bit passes
// through, where the comparison form returns 0: the bit-identity with the per-layer kernels of agz_nn.hpp, which keep the
// comparison, and the dropped second relu of the residual sum hold for FINITE weights and activations — what a network is.)
__device__ __forceinline__ float relu_bits(float x) { const int b = __float_as_int(x); return __int_as_float(b > 0 ? b : 0); }

#ifndef AGZ_NN_SWP
#define AGZ_NN_SWP 0        // A/B switch (round 6, measured: no gain — at four waves per SIMD the other workgroups' waves cover these latencies): bit 0 the 4-wave network body reads its next operand tile ahead of the running MFMAs; bit 1 the residual's old value of the next tile ahead of the store (36 spilled registers in the headline kernel)
#endif
constexpr int NW_WAVES = 4;               // waves per workgroup (= per 16-leaf tile)
constexpr int NW_DEPTH = 4;               // hidden groups are padded to a multiple of this (the deepest prefetch)

__host__ __device__ inline int nw_hidden_groups(int INP, int H, int T) {
    const int kth = H / 32, g0 = (INP / 32 + kth - 1) / kth;
    return (g0 + T + NW_DEPTH - 1) / NW_DEPTH * NW_DEPTH;
}

// LT = 16-leaf tiles per workgroup: every weight fragment fetched from L2 feeds LT MFMAs (LT = 1 for the lowest latency,
// larger LT when the batch is big enough for the L2 weight stream to become the bound).
// DEPTH = groups (layers) of weight fragments in flight per wave: 4 hides the whole L2 latency behind one workgroup's own
// work, 2 halves the registers so that twice as many workgroups share a CU (throughput mode for big batches).
// ZC: a group that opens a layer accumulates from the constant 0 instead of cleared registers (16 moves less per layer; the
// duplicated first k-step costs registers: not in the 128-register build of the whole-search kernel, where it spills)
template <int H, int LT, int DEPTH, bool PRE_BARRIER = false, bool IO = false, bool ZC = false, bool BP = false, int NWV = 4>
__device__ __forceinline__ void mlp_wave_body(const Fused3Par& P, uint8_t* const smem, const int bidx, uint8_t* const io = nullptr,
                                              const int io_bw = 0, const int io_lgs = 0, unsigned long long* const nn_dbg = nullptr);

template <int H, int LT, int DEPTH>
__global__ __launch_bounds__(64 * NW_WAVES) void k_mlp_wave(const Fused3Par P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_wave[];
    mlp_wave_body<H, LT, DEPTH>(P, smem_wave, (int)blockIdx.x);
}

// The 4-wave workgroup's forward for the leaves [bidx*16*LT, +16*LT) (also called from k_search_small); contains
// workgroup barriers: every wave of the workgroup must call it.  PRE_BARRIER: the input planes are being written by other
// waves of this workgroup; the barrier that publishes them is taken AFTER the first weight fragments have been requested.
// IO (whole-search kernel): planes and logits are handed over through LDS instead of a round trip through L2 — tile row r of the
// workgroup (game r % rb of tree wave r / rb) is row r of `io`: rows of io_lgs floats (= the row stride, >= the padded plane row),
// io_bw = rb rows per tree wave; the tree wave has left the leaf's planes there (zero padded to whole k-rows) and the head leaves
// the logits (and the value in column A) in the same row; the global arrays are not written (agz_get_logits reads what the
// stepwise API's network launch left).
// NWV: waves of the workgroup (4, or 8: 64-game workgroups of k_search_small — every wave then owns one tile of neurons instead of two)
template <int H, int LT, int DEPTH, bool PRE_BARRIER, bool IO, bool ZC, bool BP, int NWV>
__device__ __forceinline__ void mlp_wave_body(const Fused3Par& P, uint8_t* const smem, const int bidx, uint8_t* const io, const int io_bw,
                                              const int io_lgs, unsigned long long* const nn_dbg) {
#ifdef AGZ_STAMPS
    unsigned long long st_[6] = {0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
#define NN_STAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); st_[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define NN_STAMP(i) do {} while (0)
#endif
    constexpr int NTH = H / 16, KTH = H / 32, TPW = NTH / NWV;
    constexpr int ROWB = H * 2 + 16;
    static_assert(TPW >= 1, "at least one neuron tile per wave");
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_));                            // opaque per call (see rollout_reg_body)
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6) & (NWV - 1);
    constexpr int ML = 16 * LT;                                   // leaves per workgroup
    const int leaf0 = bidx * ML;
    // leaf (game slot) of tile row `row`, or a value >= P.L for an unused row
    auto leaf_of = [&](int row) -> int {
        if (P.gpw == 0) return leaf0 + row;
        const int rb = P.rb;
        return (row & (rb - 1)) < P.gpw ? (bidx * P.tw + row / rb) * P.gpw + (row & (rb - 1)) : P.L;
    };
    const int G0 = (P.INP / 32 + KTH - 1) / KTH;                 // groups of layer 0
    const int KR0 = P.INP / 32;                                  // k-rows of layer 0 that hold planes: the rows past them in its last group (zero weights x zero
                                                                 // padding) are SKIPPED — acc + 0 x 0 is acc, bit for bit — so that a row of the hand-over window
                                                                 // ends with the planes (round 6: the 128 bytes per game this frees hold the tree's next words)
    const int NGH = nw_hidden_groups(P.INP, H, P.T);             // groups before the head
    const int PROWB = G0 * KTH * 64 + 16;
    uint8_t* const act0 = smem;                                   // [ML][ROWB] x 2 (ping-pong), then [ML][PROWB] input planes
    uint8_t* const pl = smem + 2 * ML * ROWB;
    const int lrow = lane & 15, q4 = lane >> 4;
    // this wave's fragments of the running group: a scalar base that advances one group at a time + the lane's 16 bytes
    const AGZ_GLB v4u* wsrc = (const AGZ_GLB v4u*)P.w16 + (size_t)wave * TPW * 64 + lane;

    static_assert(DEPTH == 2 || DEPTH == 4, "the group loop below is unrolled by hand");
    bf16x8 A[DEPTH][KTH][TPW];
#define NW_LOADGROUP(d)                                                                                 \
    do {                                                                                                \
        _Pragma("unroll") for (int k = 0; k < KTH; ++k)                                                 \
            _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                           \
                const v4u w_ = wsrc[(k * NTH + t) * 64]; A[d][k][t] = *reinterpret_cast<const bf16x8*>(&w_); \
            }                                                                                           \
        wsrc += KTH * NTH * 64;                                                                         \
    } while (0)
    NW_LOADGROUP(0); NW_LOADGROUP(1);
    if constexpr (DEPTH == 4) { NW_LOADGROUP(2); NW_LOADGROUP(3); }
    NN_STAMP(0);
    if constexpr (PRE_BARRIER) __syncthreads();
    NN_STAMP(1);

    if constexpr (!IO) {   // the ML rows of input planes -> LDS (coalesced 16-B loads), zero beyond INP
        const int segs = G0 * KTH * 4, isegs = P.INP / 8;
        const AGZ_GLB uint16_t* gp = (const AGZ_GLB uint16_t*)P.planes;
        constexpr int TPR = 64 * NWV / ML;                   // threads per tile row (a power of two): no division by segs
        const int row = (tid_ & (64 * NWV - 1)) / TPR, mm = leaf_of(row);
        for (int seg = tid_ & (TPR - 1); seg < segs; seg += TPR) {
            v4u v = {0u, 0u, 0u, 0u};
            if (mm < P.L && seg < isegs) v = *(const AGZ_GLB v4u*)(gp + (size_t)mm * P.INP + seg * 8);
            *reinterpret_cast<v4u*>(pl + (size_t)row * PROWB + seg * 16) = v;
        }
        __syncthreads();
    }
    // tile row lrow of tile 0 of the input planes; the next 16-leaf tile is pstride bytes further
    const int io_rs = io_lgs * 4;                                 // (IO) bytes of a row of the hand-over window
    const uint8_t* const prow0 = IO ? io + (size_t)lrow * io_rs : pl + (size_t)lrow * PROWB;
    const int pstride = IO ? 16 * io_rs : 16 * PROWB;

    f32x4 acc[LT][TPW];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int lt = 0; lt < LT; ++lt)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[lt][t] = zero4;
    int cur = 0;                                                  // activation strip holding the current layer's input
    // one group: KTH k-rows of this wave's TPW tiles; closes a layer unless it is an inner group of layer 0
#define NW_GROUP(d, g)                                                                                  \
    do {                                                                                                \
        const int g_ = (g);                                                                             \
        const uint8_t* const brow_ = g_ < G0 ? prow0 + (size_t)g_ * KTH * 64                             \
                                             : act0 + (size_t)cur * ML * ROWB + (size_t)lrow * ROWB;    \
        const int bstride_ = g_ < G0 ? pstride : 16 * ROWB;       /* next 16-leaf tile */               \
        /* Builds with register room (BP: up to 2 waves per SIMD) issue ALL operand reads of the group, and the residual's old */ \
        /* activations, before the first MFMA: left to the compiler each ds_read_b128 sits right in front of its two MFMAs behind an */ \
        /* lgkmcnt(0) - eight exposed LDS latencies per layer, and four more in the epilogue (read old, wait, write new, four times). */ \
        /* The denser builds have no room for it (168 registers: 6 - 26 spilled, 3.39 -> 3.50 ms per ply at 24576 games; 128: 36 - 77) */ \
        /* and hide the latencies behind their other waves anyway; the sparse ones gain 1 - 2 % per ply. */ \
        const bool res_ = g_ >= G0;                                                                     \
        const int kmax_ = g_ < G0 ? KR0 - g_ * KTH : KTH;         /* (wave-uniform; >= 1) */           \
        bf16x8 bq_[BP ? KTH : 1][LT];                                                                   \
        uint2 old_[LT][TPW];                                                                            \
        if constexpr (BP) {                                                                             \
            /* (the reads are unconditional — a skipped k-row reads what lies behind the planes, inside the workgroup's LDS, and is not used: */ \
            /*  reads under the condition leave the operand registers partly defined and the 256-register builds spill 50 of them) */ \
            _Pragma("unroll") for (int k = 0; k < KTH; ++k)                                             \
                _Pragma("unroll") for (int lt = 0; lt < LT; ++lt)                                       \
                    bq_[k][lt] = *reinterpret_cast<const bf16x8*>(brow_ + (size_t)lt * bstride_ + k * 64 + q4 * 16); \
            if (res_) {                                                                                 \
                _Pragma("unroll") for (int lt = 0; lt < LT; ++lt)                                       \
                    _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                     \
                        old_[lt][t] = *reinterpret_cast<const uint2*>(act0 + (size_t)cur * ML * ROWB + (size_t)(lt * 16 + lrow) * ROWB \
                                                                      + (16 * (wave * TPW + t) + 4 * q4) * 2); \
            }                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                          \
            _Pragma("unroll") for (int k = 0; k < KTH; ++k) {                                           \
                /* a group that opens a layer accumulates from the constant 0 (no clearing of 16 registers per layer) */ \
                if (ZC && k == 0 && (g_ == 0 || g_ >= G0)) {                                            \
                    _Pragma("unroll") for (int lt = 0; lt < LT; ++lt)                                   \
                        _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                 \
                            acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][0][t], bq_[0][lt], (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0); \
                } else if (k == 0 || k < kmax_) {                                                       \
                    _Pragma("unroll") for (int lt = 0; lt < LT; ++lt)                                   \
                        _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                 \
                            acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][k][t], bq_[k][lt], acc[lt][t], 0, 0, 0); \
                }                                                                                       \
            }                                                                                           \
        } else if constexpr ((AGZ_NN_SWP & 1) != 0 && !ZC && NWV == 4 && LT <= 2) {                                                  \
            /* (round 6) the dense builds, software-pipelined by ONE operand tile: the ds_read_b128 of tile (k, lt) + 1 is issued in front of the */ \
            /* MFMAs of tile (k, lt) — left to the compiler every read sat right in front of its two MFMAs behind an lgkmcnt(0): eight exposed LDS */ \
            /* round trips per layer on a rollout's dependent chain.  Four more registers (the network phase has them: its 64 registers of weight */ \
            /* fragments + 16 accumulators leave room; it is the TREE step that fills the 128). */         \
            bf16x8 bring_[2];                                                                           \
            bring_[0] = *reinterpret_cast<const bf16x8*>(brow_ + q4 * 16);                              \
            _Pragma("unroll") for (int i_ = 0; i_ < KTH * LT; ++i_) {                                   \
                const int k = i_ / LT, lt = i_ % LT, kn = (i_ + 1) / LT, ltn = (i_ + 1) % LT;           \
                if (k == 0 || k < kmax_) {                                                              \
                    if (i_ + 1 < KTH * LT && (kn == 0 || kn < kmax_))                                   \
                        bring_[(i_ + 1) & 1] = *reinterpret_cast<const bf16x8*>(brow_ + (size_t)ltn * bstride_ + kn * 64 + q4 * 16); \
                    __builtin_amdgcn_sched_barrier(0);                                                  \
                    _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                     \
                        acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][k][t], bring_[i_ & 1], acc[lt][t], 0, 0, 0); \
                    __builtin_amdgcn_sched_barrier(0);                                                  \
                }                                                                                       \
            }                                                                                           \
        } else {                                                                                        \
            _Pragma("unroll") for (int k = 0; k < KTH; ++k)                                             \
                if (k == 0 || k < kmax_)                                                                \
                _Pragma("unroll") for (int lt = 0; lt < LT; ++lt) {                                     \
                    const bf16x8 b_ = *reinterpret_cast<const bf16x8*>(brow_ + (size_t)lt * bstride_ + k * 64 + q4 * 16); \
                    if (ZC && k == 0 && (g_ == 0 || g_ >= G0)) {                                        \
                        _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                 \
                            acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][0][t], b_, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0); \
                    } else {                                                                            \
                        _Pragma("unroll") for (int t = 0; t < TPW; ++t)                                 \
                            acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[d][k][t], b_, acc[lt][t], 0, 0, 0); \
                    }                                                                                   \
                }                                                                                       \
        }                                                                                               \
        NN_STAMP(2);                                                                                    \
        NW_LOADGROUP(d);                                          /* group g + DEPTH */                 \
        if (g_ >= G0 - 1) {                                                                             \
            /* (round 6, dense builds) the residual's old value of tile i + 1 is requested before tile i is stored: the stores go to the other strip, */ \
            /* but the compiler cannot know and kept every read behind the store of the tile before — four LDS round trips in a row per layer */ \
            uint2 oring_[2];                                                                            \
            auto old_at_ = [&](const int i) -> uint2 {                                                  \
                return *reinterpret_cast<const uint2*>(act0 + (size_t)cur * ML * ROWB + (size_t)((i / TPW) * 16 + lrow) * ROWB + (16 * (wave * TPW + (i % TPW)) + 4 * q4) * 2); \
            };                                                                                          \
            if constexpr (!BP && (AGZ_NN_SWP & 2) != 0) { if (res_) oring_[0] = old_at_(0); }           \
            _Pragma("unroll") for (int lt = 0; lt < LT; ++lt) {                                         \
                uint8_t* const new_ = act0 + (size_t)(cur ^ 1) * ML * ROWB + (size_t)(lt * 16 + lrow) * ROWB; \
                _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                       \
                    const int n = 16 * (wave * TPW + t) + 4 * q4;  /* acc[lt][t][r] = out[neuron n + r][leaf 16 lt + lrow] */ \
                    float x0 = relu_bits(acc[lt][t][0]), x1 = relu_bits(acc[lt][t][1]);                 \
                    float x2 = relu_bits(acc[lt][t][2]), x3 = relu_bits(acc[lt][t][3]);                 \
                    if (res_) {                                    /* b = relu(b + relu(W b)) */        \
                        uint2 o;                                                                        \
                        if constexpr (BP) o = old_[lt][t];                                              \
                        else if constexpr ((AGZ_NN_SWP & 2) != 0) {                                     \
                            const int i_ = lt * TPW + t;                                                \
                            o = oring_[i_ & 1];                                                         \
                            if (i_ + 1 < LT * TPW) oring_[(i_ + 1) & 1] = old_at_(i_ + 1);              \
                            __builtin_amdgcn_sched_barrier(0);                                          \
                        } else o = *reinterpret_cast<const uint2*>(act0 + (size_t)cur * ML * ROWB + (size_t)(lt * 16 + lrow) * ROWB + n * 2); \
                        x0 += __uint_as_float(o.x << 16); x1 += __uint_as_float(o.x & 0xffff0000u);     \
                        x2 += __uint_as_float(o.y << 16); x3 += __uint_as_float(o.y & 0xffff0000u);     \
                        /* (no second relu: b >= 0 and relu(W b) >= 0, so the sum is its own relu, bit for bit) */ \
                    }                                                                                   \
                    *reinterpret_cast<uint2*>(new_ + n * 2) = make_uint2(pk_bf16(x0, x1), pk_bf16(x2, x3)); \
                    if constexpr (!ZC) acc[lt][t] = zero4;                                              \
                }                                                                                       \
            }                                                                                           \
            cur ^= 1;                                                                                   \
            NN_STAMP(3);                                                                                \
            __syncthreads();                                                                            \
            NN_STAMP(4);                                                                                \
        }                                                                                               \
    } while (0)
#pragma unroll 1
    for (int g = 0; g < NGH; g += DEPTH) {
        NW_GROUP(0, g); NW_GROUP(1, g + 1);
        if constexpr (DEPTH == 4) { NW_GROUP(2, g + 2); NW_GROUP(3, g + 3); }
    }
#undef NW_GROUP
#undef NW_LOADGROUP

    {   // head: D = X * W^T (logits leave row-major); its fragments sit in buffer 0 (NGH is a multiple of DEPTH).  A head wider than
        // the trunk (A + 1 > H, up to 2 H outputs: Gobang 13x13 on a 128-wide trunk) has a second group of tiles, which the group
        // loop above has already brought into buffer 1
        const uint8_t* const brow = act0 + (size_t)cur * ML * ROWB + (size_t)lrow * ROWB;
        const int NT = P.AOP / 16;
        int mrow[LT][4];                                            // game slot of tile row 16 lt + 4 q4 + r (once, not per tile)
#pragma unroll
        for (int lt = 0; lt < LT; ++lt) {                          // rows 16 lt + 4 q4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) mrow[lt][r] = leaf_of(16 * lt + 4 * q4 + r);
        }
#define NW_HEAD(buf, tile0)                                                                             \
        if ((tile0) + wave * TPW < NT) {                                                                \
            _Pragma("unroll") for (int k = 0; k < KTH; ++k)                                             \
                _Pragma("unroll") for (int lt = 0; lt < LT; ++lt) {                                     \
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + (size_t)lt * 16 * ROWB + k * 64 + q4 * 16); \
                    _Pragma("unroll") for (int t = 0; t < TPW; ++t) acc[lt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, A[buf][k][t], k == 0 ? zero4 : acc[lt][t], 0, 0, 0); \
                }                                                                                       \
            /* acc[lt][t][r] = out[leaf = leaf0 + 16 lt + 4 q4 + r][n = 16 tile + (lane & 15)] */        \
            _Pragma("unroll") for (int t = 0; t < TPW; ++t) {                                           \
                const int tile = (tile0) + wave * TPW + t;                                              \
                if (tile < NT) {                                                                        \
                    const int n = 16 * tile + (lane & 15);                                              \
                    const float bias = P.bias_head[n];                                                  \
                    _Pragma("unroll") for (int lt = 0; lt < LT; ++lt) {                                 \
                        /* (IO) row 16 lt + 4 q4 + r of the hand-over window */                        \
                        float* const lrow_ = IO ? reinterpret_cast<float*>(io) + (size_t)(16 * lt + 4 * q4) * io_lgs + n : nullptr; \
                        if (n < P.A) {                                                                  \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                             \
                                const int m = mrow[lt][r]; const float o = acc[lt][t][r] + bias;        \
                                if (!IO && m < P.L) P.logits[(size_t)m * P.LGS + n] = o;                \
                                if constexpr (IO) lrow_[(size_t)r * io_lgs] = o;                        \
                            }                                                                           \
                        } else if (n == P.A) {                                                          \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                             \
                                const int m = mrow[lt][r]; const float o = sigmoid_ool(acc[lt][t][r] + bias); \
                                if (!IO && m < P.L) P.vout[m] = o;                                      \
                                if constexpr (IO) lrow_[(size_t)r * io_lgs] = o;                        \
                            }                                                                           \
                        }                                                                               \
                    }                                                                                   \
                }                                                                                       \
            }                                                                                           \
        }
        NW_HEAD(0, 0)
        if (NT > NTH) { NW_HEAD(1, NTH) }                           // (wave-uniform, rare)
#undef NW_HEAD
    }
    NN_STAMP(5);
#ifdef AGZ_STAMPS
    // [0] weight requests, [1] first barrier (wait for the tree waves), [2] B reads + MFMA issue (waits for the weights),
    // [3] epilogue (waits for the MFMAs), [4] layer barriers, [5] head
    if (nn_dbg && lane == 0) for (int i = 0; i < 6; ++i) nn_dbg[i] += st_[i];
#endif
#undef NN_STAMP
}

}  // namespace agz
