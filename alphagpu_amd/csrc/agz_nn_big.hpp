// agz_nn_big.hpp — the snetwork2 forward (DenseNet.jl:294-304) for WIDE trunks (H = 256, 512; BASELINE configs 3-5 use
// 512x8) in one launch: 8 waves, 128 leaves per workgroup, activations resident in LDS, weights streamed from L2.
//
// With H = 512 a layer is 128 x 512 x 512: the 133 KiB activation tile [128][H] bf16 stays in LDS for the whole network
// and is updated IN PLACE — every wave first accumulates its whole share of the layer in registers (wave w owns the
// neurons [w*H/8, (w+1)*H/8) of all 128 leaves: 8 leaf tiles x NTW neuron tiles of v_mfma_f32_16x16x32_bf16, 128 fp32
// accumulator registers at H = 512), a barrier ends all reads of the layer's input, then the bf16 epilogue overwrites it.
// Weights: the pre-tiled 1 KiB fragments (tile (kt, nt) of a layer at (kt*NT + nt) KiB) go from L2 straight into the
// MFMA A operand; a wave reads only its own neuron tiles (no redundancy inside the workgroup) and keeps two k-rows in
// flight ahead of the matrix core, across layer boundaries.  One workgroup per CU: a 32768-leaf batch is one round.
// 2 * 128 * (INP*H + T*H*H + AOP*H) flop per workgroup; 512x8 on Gobang 9x9: 4.44 MFLOP / leaf, 146 GFLOP / launch.
#pragma once
#include "agz_nn_wave.hpp"

#ifndef AGZ_BIG_PIPE
#define AGZ_BIG_PIPE 1      // 0: the two-rows-in-flight k-loop everywhere (A/B)
#endif

namespace agz {

constexpr int NB_M = 128;                 // leaves per workgroup (MT = 8 leaf tiles; MT = 2 -> 32 leaves for small batches)
constexpr int NB_THREADS = 512;

struct BigPar {
    const uint16_t* planes; int INP;      // [L][INP] bf16
    const uint16_t* wh;                   // hidden k-rows: layer 0 (padded with zero rows to an even count), T residual layers, 2 rows of slack
    const uint16_t* whead;                // head k-rows, AOP/16 tiles each
    const float* bias_head;
    float* logits; int LGS; float* vout;
    int L, T, A, AOP, K0R;                // K0R = k-rows of layer 0 after padding
    int ROWB;                             // bytes per activation row in LDS: 2 * max(H, 32*K0R) rounded up to 256 (XOR-swizzled, below)
};

// MT = 16-leaf tiles per workgroup: 8 (128 leaves) when the batch fills the chip, 2 (32 leaves) below ~8192 leaves, where
// the launch is bound by one workgroup's own chain of layers and a quarter of the work per workgroup is ~4x faster.
// mlp_big_body: the 8-wave workgroup's forward for the leaves of tile rows 0 .. 16 MT - 1, slot_of(row) = leaf (game slot) of a
// row or a value >= P.L (also called from k_search_big, agz_search_big.hpp); contains workgroup barriers.
// PIPE (builds with register room: one workgroup per CU): FOUR k-rows of weight fragments in flight — a fragment is requested three k-steps
// (~1.5 K cycles of matrix work) before the matrix core needs it; with two rows the request of a row went out when the row before
// it had just been used up, and its L2 latency stood in front of every second k-step — and the B operand of leaf tile i + 2 is read from
// LDS while the MFMAs of tile i issue (explicit schedule groups).  Same MFMA instruction, same k order per accumulator: same bits.
template <int H, int MT, bool PIPE = false, typename SlotOf>
__device__ __forceinline__ void mlp_big_body(const BigPar& P, uint8_t* const act, SlotOf slot_of) {
    constexpr int NT = H / 16, KTH = H / 32, NTW = NT / 8;       // neuron tiles per layer / k-rows per layer / neuron tiles per wave
    constexpr int MB = 16 * MT;                                  // leaves per workgroup
    const int ROWB = P.ROWB;
    static_assert(NTW >= 1 && KTH % 2 == 0, "H must be a multiple of 128");
    int tid_ = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_));                               // opaque per call (a caller may loop over rollouts)
    const int tid = tid_ & (NB_THREADS - 1), lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, q4 = lane >> 4;
    const AGZ_GLB v4u* wsrc = (const AGZ_GLB v4u*)P.wh + (size_t)wave * NTW * 64 + lane;   // this wave's tiles of k-row 0

    bf16x8 A0[NTW], A1[NTW], A2[PIPE ? NTW : 1], A3[PIPE ? NTW : 1];
#define NB_LOADROW(buf)                                                                                 \
    do {                                                                                                \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) { const v4u w_ = wsrc[t * 64]; buf[t] = *reinterpret_cast<const bf16x8*>(&w_); } \
        wsrc += NT * 64;                                                                                \
    } while (0)
    NB_LOADROW(A0); NB_LOADROW(A1);
    if constexpr (PIPE) { NB_LOADROW(A2); NB_LOADROW(A3); }

    {   // input planes -> columns [0, 32*K0R) of the activation tile, zero beyond INP
        const int segs = P.K0R * 4, isegs = P.INP / 8;
        const AGZ_GLB uint16_t* gp = (const AGZ_GLB uint16_t*)P.planes;
        for (int c = tid; c < MB * segs; c += NB_THREADS) {
            const int row = c / segs, seg = c - row * segs, mm = slot_of(row);
            v4u v = {0u, 0u, 0u, 0u};
            if (mm < P.L && seg < isegs) v = *(const AGZ_GLB v4u*)(gp + (size_t)mm * P.INP + seg * 8);
            *reinterpret_cast<v4u*>(act + (size_t)row * ROWB + ((seg * 16) ^ ((row & 15) << 4))) = v;
        }
    }
    __syncthreads();

    // LDS layout of the activation tile: row r at r * ROWB (a multiple of 256 bytes = one bank row), byte b of the row at b ^ ((r & 15) << 4):
    // the 16-byte slot of a k-chunk is XORed with the row's low bits.  ds_read_b128 is serviced in four groups of 16 lanes
    // ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) whose lanes read 16 different rows at one or two neighbouring k-chunks —
    // with the XOR every group touches all 16 slots of the bank row once (a padded stride of 256 n + 16 bytes left one slot of every
    // group busy twice: half the read rate of the B operand, which all eight waves read in full for every layer).
    const int swz = lrow << 4;
    const uint8_t* const brow = act + (size_t)lrow * ROWB;               // + 16*mt*ROWB + ((64*kt + 16*q4) ^ swz)
    const int cx = (q4 * 16) ^ swz;                                      // (64 kt and 16 q4 share no bit: (64 kt + 16 q4) ^ swz = 64 kt ^ cx)
    f32x4 acc[MT][NTW];
#define NB_STEP(buf, kt)                                                                                \
    do {                                                                                                \
        bf16x8 b_[MT];                                                                                  \
        const int off_ = ((kt) * 64) ^ cx;                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) b_[mt] = *reinterpret_cast<const bf16x8*>(brow + (size_t)mt * 16 * ROWB + off_); \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                           \
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[t], b_[mt], acc[mt][t], 0, 0, 0); \
        NB_LOADROW(buf);                                          /* k-row + 2 (next layer's after the last two) */ \
    } while (0)

#pragma unroll 1
    for (int l = 0; l <= P.T; ++l) {                             // input + residual layers: D = W * X^T
        const int KTl = l == 0 ? P.K0R : KTH;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NTW; ++t) { acc[mt][t][0] = 0.0f; acc[mt][t][1] = 0.0f; acc[mt][t][2] = 0.0f; acc[mt][t][3] = 0.0f; }
        if constexpr (PIPE) {
#define NB_STEP4(buf, kt)                                                                               \
    do {                                                                                                \
        bf16x8 b_[MT];                                                                                  \
        const int off_ = ((kt) * 64) ^ cx;                                                              \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) b_[mt] = *reinterpret_cast<const bf16x8*>(brow + (size_t)mt * 16 * ROWB + off_); \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
            _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                             \
                acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[t], b_[mt], acc[mt][t], 0, 0, 0); \
        NB_LOADROW(buf);                                          /* k-row + 4 */                       \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        /* the B operands of two leaf tiles up front ... */ \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                \
            __builtin_amdgcn_sched_group_barrier(0x008, NTW, 0);  /* ... then NTW MFMAs per tile, */    \
            if (i + 2 < MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* each followed by the read for tile i + 2 */ \
        }                                                                                               \
        __builtin_amdgcn_sched_group_barrier(0x020, NTW, 0);      /* the weight fragments of row + 4 */ \
    } while (0)
            // (at the loop's back edge the compiler waits for ALL outstanding weight requests — vmcnt(0) — instead of the four oldest:
            //  once per four k-steps; unrolling a layer's loop in full removes that wait and spills ~390 registers)
#pragma unroll 1
            for (int kt = 0; kt < KTl; kt += 4) { NB_STEP4(A0, kt); NB_STEP4(A1, kt + 1); NB_STEP4(A2, kt + 2); NB_STEP4(A3, kt + 3); }
#undef NB_STEP4
        } else {
#pragma unroll 1
            for (int kt = 0; kt < KTl; kt += 2) { NB_STEP(A0, kt); NB_STEP(A1, kt + 1); }
        }
        __syncthreads();                                          // every wave has read the layer's input
        const bool res = l > 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            uint8_t* const orow = act + (size_t)(mt * 16 + lrow) * ROWB;   // (this lane's row: its bytes at b ^ swz)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int n = 16 * (wave * NTW + t) + 4 * q4;     // acc[mt][t][r] = out[neuron n + r][leaf 16 mt + lrow]
                float x0 = acc[mt][t][0] > 0.0f ? acc[mt][t][0] : 0.0f, x1 = acc[mt][t][1] > 0.0f ? acc[mt][t][1] : 0.0f;
                float x2 = acc[mt][t][2] > 0.0f ? acc[mt][t][2] : 0.0f, x3 = acc[mt][t][3] > 0.0f ? acc[mt][t][3] : 0.0f;
                uint2* dst = reinterpret_cast<uint2*>(orow + ((n * 2) ^ swz));
                if (res) {                                        // b = relu(b + relu(W b))
                    const uint2 o = *dst;
                    x0 += __uint_as_float(o.x << 16); x1 += __uint_as_float(o.x & 0xffff0000u);
                    x2 += __uint_as_float(o.y << 16); x3 += __uint_as_float(o.y & 0xffff0000u);
                    // (no second relu: b >= 0 and relu(W b) >= 0, so the sum is its own relu, bit for bit)
                }
                *dst = make_uint2(pk_bf16(x0, x1), pk_bf16(x2, x3));
            }
        }
        __syncthreads();
    }
#undef NB_STEP
#undef NB_LOADROW

    {   // head: D = X * W^T (logits leave row-major); wave w takes neuron tile w of the AOP/16 head tiles
        const int NTH = P.AOP / 16;
        for (int tile = wave; tile < NTH; tile += 8) {
            f32x4 hacc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { hacc[mt][0] = 0.0f; hacc[mt][1] = 0.0f; hacc[mt][2] = 0.0f; hacc[mt][3] = 0.0f; }
            const AGZ_GLB v4u* hw = (const AGZ_GLB v4u*)P.whead + (size_t)tile * 64 + lane;
#pragma unroll 4
            for (int kt = 0; kt < KTH; ++kt) {
                const v4u w_ = hw[(size_t)kt * NTH * 64];
                const bf16x8 w = *reinterpret_cast<const bf16x8*>(&w_);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 x = *reinterpret_cast<const bf16x8*>(brow + (size_t)mt * 16 * ROWB + ((kt * 64 + q4 * 16) ^ swz));
                    hacc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, w, hacc[mt], 0, 0, 0);
                }
            }
            // hacc[mt][r] = out[leaf of tile row 16 mt + 4 q4 + r][n = 16 tile + (lane & 15)]
            const int n = 16 * tile + (lane & 15);
            const float bias = P.bias_head[n];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int mw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) mw[r] = slot_of(16 * mt + 4 * q4 + r);
                if (n < P.A) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (mw[r] < P.L) P.logits[(size_t)mw[r] * P.LGS + n] = hacc[mt][r] + bias;
                } else if (n == P.A) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (mw[r] < P.L) P.vout[mw[r]] = sigmoid_ool(hacc[mt][r] + bias);
                }
            }
        }
    }
}

template <int H, int MT>
__global__ __launch_bounds__(NB_THREADS, 1) void k_mlp_big(const BigPar P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t act_big[];   // [16 MT][ROWB]
    const int leaf0 = (int)blockIdx.x * 16 * MT;
    mlp_big_body<H, MT, (AGZ_BIG_PIPE != 0)>(P, act_big, [&](int row) { return leaf0 + row; });
}

}  // namespace agz
