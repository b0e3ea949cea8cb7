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
// PREB: the caller's leaves are still being written (planes in global memory, by other waves of this workgroup): the barrier that publishes
// them is taken HERE, behind the first requests for weight fragments — a wave that arrives early has its first k-rows on the way while it waits
template <int H, int MT, bool PIPE = false, bool PREB = false, typename SlotOf>
__device__ __forceinline__ void mlp_big_body(const BigPar& P, uint8_t* const act, SlotOf slot_of, unsigned long long* const nn_dbg = nullptr) {
#ifdef AGZ_BIG4STAMPS
    unsigned long long nst_[6] = {0, 0, 0, 0, 0, 0}, nst_t = __builtin_amdgcn_s_memtime();
#define NBSTAMP(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); nst_[i] += n_ - nst_t; nst_t = n_; } while (0)
#else
#define NBSTAMP(i) do { } while (0)
#endif
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
    if constexpr (PREB) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();                                          // the planes of the workgroup's leaves are written
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }

    {   // input planes -> columns [0, 32*K0R) of the activation tile, zero beyond INP: NB_THREADS / MB threads per row, each takes every
        // (NB_THREADS / MB)-th 16-byte segment of its row; six loads are requested before the first is stored (the loop used to pay one memory
        // round trip and an integer division per segment: 4 us per pass)
        const int segs = P.K0R * 4, isegs = P.INP / 8;
        const AGZ_GLB uint16_t* gp = (const AGZ_GLB uint16_t*)P.planes;
        constexpr int TPR = NB_THREADS / MB, NB6 = 6;
        const int row = tid / TPR, s0 = tid % TPR, mm = slot_of(row);
        const AGZ_GLB uint16_t* const src = gp + (size_t)(mm < P.L ? mm : 0) * P.INP;
        uint8_t* const drow = act + (size_t)row * ROWB;
        const int rsw = (row & 15) << 4;
        for (int sb = s0; sb < segs; sb += NB6 * TPR) {
            v4u v[NB6];
#pragma unroll
            for (int j = 0; j < NB6; ++j) {
                const int seg = sb + j * TPR;
                v[j] = (v4u){0u, 0u, 0u, 0u};
                if (mm < P.L && seg < isegs) v[j] = *(const AGZ_GLB v4u*)(src + seg * 8);
            }
#pragma unroll
            for (int j = 0; j < NB6; ++j) {
                const int seg = sb + j * TPR;
                if (seg < segs) *reinterpret_cast<v4u*>(drow + ((seg * 16) ^ rsw)) = v[j];
            }
        }
    }
    __syncthreads();
    NBSTAMP(0);

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
        NBSTAMP(1);
        const bool res = l > 0;
        if constexpr (PIPE) {
            // builds with register room: the epilogue's ARITHMETIC runs in front of the barrier — relu, the residual's old value (this lane's own
            // bytes of the tile: nobody else writes them), the sum, the bf16 pair — while the slower waves still read the layer's input; behind
            // the barrier only the stores are left (measured on the 128-leaf pass: 2.6 us of epilogue + 1.4 us at the barrier per layer before)
            uint2 pk[MT][NTW];
            // (the old values are requested four tiles ahead of their use: left alone the compiler waits for every single read; all of them at once
            //  — 64 registers beside 128 accumulators and four k-rows of fragments in flight — spill.  The reads are UNCONDITIONAL — layer 0 reads
            //  its own input bytes and masks them — so that the block is straight-line code: behind a branch per tile the ring of registers
            //  became moves that waited for the read just issued)
            constexpr int NTILE = MT * NTW, PD = NTILE >= 4 ? 4 : NTILE;
            uint2 oldv[PD];
            auto old_at = [&](const int i) -> uint2 {
                return *reinterpret_cast<const uint2*>(act + (size_t)((i / NTW) * 16 + lrow) * ROWB + (((16 * (wave * NTW + (i % NTW)) + 4 * q4) * 2) ^ swz));
            };
#pragma unroll
            for (int i = 0; i < PD; ++i) oldv[i] = old_at(i);
#pragma unroll
            for (int i = 0; i < NTILE; ++i) {
                const int mt = i / NTW, t = i % NTW;
                // (relu as one integer max per element, see relu_bits: same bits for finite accumulators)
                float x0 = relu_bits(acc[mt][t][0]), x1 = relu_bits(acc[mt][t][1]), x2 = relu_bits(acc[mt][t][2]), x3 = relu_bits(acc[mt][t][3]);
                uint2 o = oldv[i % PD];
                if (i + PD < NTILE) oldv[i % PD] = old_at(i + PD);
                o.x = res ? o.x : 0u; o.y = res ? o.y : 0u;       // (x + 0 = x for the non-negative x of a relu: layer 0 is unchanged)
                x0 += __uint_as_float(o.x << 16); x1 += __uint_as_float(o.x & 0xffff0000u);
                x2 += __uint_as_float(o.y << 16); x3 += __uint_as_float(o.y & 0xffff0000u);
                pk[mt][t] = make_uint2(pk_bf16(x0, x1), pk_bf16(x2, x3));
                __builtin_amdgcn_sched_barrier(0);
            }
            NBSTAMP(3);
            __syncthreads();                                      // every wave has read the layer's input
            NBSTAMP(2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                uint8_t* const orow = act + (size_t)(mt * 16 + lrow) * ROWB;
#pragma unroll
                for (int t = 0; t < NTW; ++t) *reinterpret_cast<uint2*>(orow + (((16 * (wave * NTW + t) + 4 * q4) * 2) ^ swz)) = pk[mt][t];
            }
            NBSTAMP(3);
            __syncthreads();
            NBSTAMP(4);
            continue;
        }
        __syncthreads();                                          // every wave has read the layer's input
        NBSTAMP(2);
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
        NBSTAMP(3);
        __syncthreads();
        NBSTAMP(4);
    }
#undef NB_STEP
#undef NB_LOADROW

    {   // head: D = X * W^T (logits leave row-major); wave w takes neuron tile w of the AOP/16 head tiles
        const int NTH = P.AOP / 16;
        if constexpr (PIPE && MT >= 4) {
            // builds with register room: a head tile's leaves in four parts, NTH x 4 (tile, part) units dealt to the waves in runs of consecutive
            // units — a head of six tiles (Gobang 9x9: 82 outputs) keeps all eight waves busy instead of six — so that a wave needs the
            // fragments of two tiles at most: both requested at once, up front
            constexpr int HQ = 4, MQ = MT / HQ;
            const int NU = NTH * HQ, UPW = (NU + 7) / 8, u0 = wave * UPW, u1 = u0 + UPW < NU ? u0 + UPW : NU;
            auto head_load = [&](bf16x8 (&dst)[KTH], const int tile) {
                const AGZ_GLB v4u* hw = (const AGZ_GLB v4u*)P.whead + (size_t)tile * 64 + lane;
#pragma unroll
                for (int kt = 0; kt < KTH; ++kt) { const v4u w_ = hw[(size_t)kt * NTH * 64]; dst[kt] = *reinterpret_cast<const bf16x8*>(&w_); }
            };
            auto head_unit = [&](const bf16x8 (&w)[KTH], const int u) {
                const int tile = u / HQ, m0 = (u % HQ) * MQ;
                f32x4 hacc[MQ];
#pragma unroll
                for (int m = 0; m < MQ; ++m) { hacc[m][0] = 0.0f; hacc[m][1] = 0.0f; hacc[m][2] = 0.0f; hacc[m][3] = 0.0f; }
#pragma unroll
                for (int kt = 0; kt < KTH; ++kt) {
#pragma unroll
                    for (int m = 0; m < MQ; ++m) {
                        const bf16x8 x = *reinterpret_cast<const bf16x8*>(brow + (size_t)(m0 + m) * 16 * ROWB + ((kt * 64 + q4 * 16) ^ swz));
                        hacc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, w[kt], hacc[m], 0, 0, 0);
                    }
                }
                // hacc[m][r] = out[leaf of tile row 16 (m0 + m) + 4 q4 + r][n = 16 tile + (lane & 15)]
                const int n = 16 * tile + (lane & 15);
                const float bias = P.bias_head[n];
#pragma unroll
                for (int m = 0; m < MQ; ++m) {
                    int mw[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) mw[r] = slot_of(16 * (m0 + m) + 4 * q4 + r);
                    if (n < P.A) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (mw[r] < P.L) P.logits[(size_t)mw[r] * P.LGS + n] = hacc[m][r] + bias;
                    } else if (n == P.A) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (mw[r] < P.L) P.vout[mw[r]] = sigmoid_ool(hacc[m][r] + bias);
                    }
                }
            };
            if (u0 < NU) {
                bf16x8 wa[KTH], wb[KTH];
                const int ta = u0 / HQ, tb = (u1 - 1) / HQ;
                head_load(wa, ta);
                if (tb != ta) head_load(wb, ta + 1);
                int u = u0;
#pragma unroll 1
                for (; u < u1 && u / HQ == ta; ++u) head_unit(wa, u);
                int cur = ta + 1;
#pragma unroll 1
                for (; u < u1; ++u) {
                    if (u / HQ != cur) { cur = u / HQ; head_load(wb, cur); }   // (a wide head: more than two tiles per wave)
                    head_unit(wb, u);
                }
            }
        } else
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
    NBSTAMP(5);
#ifdef AGZ_BIG4STAMPS
    // [0] first weight requests + planes -> LDS + barrier, [1] k-loops, [2] barrier behind them, [3] epilogues, [4] barrier behind them, [5] head
    if (nn_dbg && lane == 0) for (int i = 0; i < 6; ++i) atomicAdd(nn_dbg + i, nst_[i]);
#endif
#undef NBSTAMP
}

template <int H, int MT>
__global__ __launch_bounds__(NB_THREADS, 1) void k_mlp_big(const BigPar P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t act_big[];   // [16 MT][ROWB]
    const int leaf0 = (int)blockIdx.x * 16 * MT;
    mlp_big_body<H, MT, (AGZ_BIG_PIPE != 0)>(P, act_big, [&](int row) { return leaf0 + row; });
}

}  // namespace agz
