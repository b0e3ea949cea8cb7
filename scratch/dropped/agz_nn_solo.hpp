// agz_nn_solo.hpp — the snetwork2 forward (DenseNet.jl:294-304) of ONE wavefront for its own 8 leaves: no workgroup barrier.
//
// Inside the whole-search kernel (agz_search_small.hpp) a tree wave owns 8 games.  The 4-wave forward of agz_nn_wave.hpp makes
// the four tree waves of a workgroup meet at a barrier twice per rollout and once per layer: at 128 games per CU that phase
// (waiting for the slowest tree wave + 9 barriers under 16-wave contention) was 30 % of a rollout.  Here every wave evaluates
// the whole network for its own leaves: rows 0..7 of the 16-row MFMA tile are the leaves (rows 8..15 mirror them and are
// discarded), the wave owns ALL neuron tiles, streams every weight fragment from L2 itself (the network is small: 278 KB at
// 128x6, L2-resident; 4x the L2 traffic of the shared form, none of its waiting) and keeps one k-row (NTH fragments) in flight
// ahead of the matrix core.  Activations ping-pong between two 8-row LDS strips private to the wave.
// Same MFMA instruction, operand order and k order as mlp_wave_body -> bit-identical logits and values.
#pragma once
#include "agz_nn_wave.hpp"

namespace agz {

// LDS of one wave: two activation strips [8][H*2+16]; the input planes [8][PROWB] lie over them (they are dead when layer 0's
// outputs are written)
__host__ __device__ inline int nn_solo_lds(int INP, int H) {
    const int kth = H / 32, g0 = (INP / 32 + kth - 1) / kth;
    const int a = 8 * 2 * (H * 2 + 16), b = 8 * (g0 * kth * 64 + 16);
    return ((a > b ? a : b) + 15) & ~15;
}

// P.w16 = the uniform tiling of agz_nn_wave.hpp (groups of KTH k-rows x NTH tiles); slot_of(row) = game slot of leaf row (>= P.L: none)
template <int H, typename SlotOf>
__device__ __forceinline__ void mlp_solo_body(const Fused3Par& P, uint8_t* const smem, SlotOf slot_of) {
    constexpr int NTH = H / 16, KTH = H / 32;
    constexpr int ROWB = H * 2 + 16;
    int lane_ = (int)threadIdx.x;
    asm volatile("" : "+v"(lane_));                               // opaque per call (see rollout_eager_body)
    const int lane = lane_ & 63;
    const int KT0 = P.INP / 32;                                   // k-rows of layer 0 that hold weights
    const int G0 = (KT0 + KTH - 1) / KTH;                         // groups of layer 0 in the tiling
    const int NGH = nw_hidden_groups(P.INP, H, P.T);              // groups before the head
    const int PROWB = G0 * KTH * 64 + 16;
    uint8_t* const act0 = smem;                                   // [8][ROWB] x 2 (ping-pong)
    uint8_t* const pl = smem;                                     // [8][PROWB] input planes, over the strips (dead before they are written)
    const int lrow = lane & 15, q4 = lane >> 4, r8 = lrow & 7;    // tile row (leaf), k quarter; rows 8..15 mirror rows 0..7
    const AGZ_GLB v4u* const wbase = (const AGZ_GLB v4u*)P.w16 + lane;

    bf16x8 A0[NTH], A1[NTH];
    // k-row `row` of the tiling (row = group * KTH + k), tiles 0..NTH-1
#define NS_LOAD(buf, row)                                                                               \
    do {                                                                                                \
        const AGZ_GLB v4u* const s_ = wbase + (size_t)(row) * NTH * 64;                                 \
        _Pragma("unroll") for (int t = 0; t < NTH; ++t) { const v4u w_ = s_[t * 64]; buf[t] = *reinterpret_cast<const bf16x8*>(&w_); } \
    } while (0)
    const int row_head = NGH * KTH, row_l1 = P.T > 0 ? G0 * KTH : row_head;   // first k-row of the head / of the layer after layer 0
    NS_LOAD(A0, 0);
    {   // the 8 rows of input planes -> LDS (16-B pieces), zero beyond INP
        const int segs = G0 * KTH * 4, isegs = P.INP / 8;
        const AGZ_GLB uint16_t* gp = (const AGZ_GLB uint16_t*)P.planes;
        const int row = lane >> 3, mm = slot_of(row);
        for (int seg = lane & 7; seg < segs; seg += 8) {
            v4u v = {0u, 0u, 0u, 0u};
            if (mm < P.L && seg < isegs) v = *(const AGZ_GLB v4u*)(gp + (size_t)mm * P.INP + seg * 8);
            *reinterpret_cast<v4u*>(pl + (size_t)row * PROWB + seg * 16) = v;
        }
    }
    AGZ_WSYNC();

    f32x4 acc[NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) { acc[t][0] = 0.0f; acc[t][1] = 0.0f; acc[t][2] = 0.0f; acc[t][3] = 0.0f; }
    int cur = 0;
    // one k-row: acc[t] += W[tile t][k-row] * X[k-row]  (A = weights, B = activations: a lane holds 4 consecutive neurons of one leaf)
#define NS_STEP(buf, bptr)                                                                              \
    do {                                                                                                \
        const bf16x8 b_ = *reinterpret_cast<const bf16x8*>(bptr);                                       \
        _Pragma("unroll") for (int t = 0; t < NTH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[t], b_, acc[t], 0, 0, 0); \
    } while (0)
    // bf16 epilogue of a hidden layer: b = relu(W x) (layer 0) or relu(b + relu(W b)) (residual layers)
    auto close_layer = [&](const bool res) {
        const uint8_t* const old_ = act0 + (size_t)cur * 8 * ROWB + (size_t)r8 * ROWB;
        uint8_t* const new_ = act0 + (size_t)(cur ^ 1) * 8 * ROWB + (size_t)r8 * ROWB;
#pragma unroll
        for (int t = 0; t < NTH; ++t) {
            const int n = 16 * t + 4 * q4;                        // acc[t][r] = out[neuron n + r][leaf lrow]
            float x0 = acc[t][0] > 0.0f ? acc[t][0] : 0.0f, x1 = acc[t][1] > 0.0f ? acc[t][1] : 0.0f;
            float x2 = acc[t][2] > 0.0f ? acc[t][2] : 0.0f, x3 = acc[t][3] > 0.0f ? acc[t][3] : 0.0f;
            if (res) {
                const uint2 o = *reinterpret_cast<const uint2*>(old_ + n * 2);
                x0 += __uint_as_float(o.x << 16); x1 += __uint_as_float(o.x & 0xffff0000u);
                x2 += __uint_as_float(o.y << 16); x3 += __uint_as_float(o.y & 0xffff0000u);
                x0 = x0 > 0.0f ? x0 : 0.0f; x1 = x1 > 0.0f ? x1 : 0.0f; x2 = x2 > 0.0f ? x2 : 0.0f; x3 = x3 > 0.0f ? x3 : 0.0f;
            }
            if (lrow < 8) *reinterpret_cast<uint2*>(new_ + n * 2) = make_uint2(pk_bf16(x0, x1), pk_bf16(x2, x3));
            acc[t][0] = 0.0f; acc[t][1] = 0.0f; acc[t][2] = 0.0f; acc[t][3] = 0.0f;
        }
        cur ^= 1;
        AGZ_WSYNC();
    };

    // ---- layer 0: KT0 k-rows over the planes (two per turn of the loop: the fragment buffers alternate)
    {
        const uint8_t* const brow = pl + (size_t)r8 * PROWB + q4 * 16;
        int k = 0;
#pragma unroll 1
        for (; k + 1 < KT0; k += 2) {
            NS_LOAD(A1, k + 1);
            NS_STEP(A0, brow + k * 64);
            NS_LOAD(A0, k + 2 < KT0 ? k + 2 : row_l1);
            NS_STEP(A1, brow + (k + 1) * 64);
        }
        if (k < KT0) {                                            // odd count: one row left in A0
            NS_LOAD(A1, row_l1);
            NS_STEP(A0, brow + k * 64);
#pragma unroll
            for (int t = 0; t < NTH; ++t) A0[t] = A1[t];
        }
        close_layer(false);
    }
    // ---- residual layers: KTH k-rows each (KTH is even), A0 holds the layer's first row on entry
#pragma unroll 1
    for (int l = 0; l < P.T; ++l) {
        const int row0 = (G0 + l) * KTH;
        const int next0 = l + 1 < P.T ? row0 + KTH : row_head;
        const uint8_t* const brow = act0 + (size_t)cur * 8 * ROWB + (size_t)r8 * ROWB + q4 * 16;
#pragma unroll
        for (int k = 0; k < KTH; k += 2) {
            NS_LOAD(A1, row0 + k + 1);
            NS_STEP(A0, brow + k * 64);
            NS_LOAD(A0, k + 2 < KTH ? row0 + k + 2 : next0);
            NS_STEP(A1, brow + (k + 1) * 64);
        }
        close_layer(true);
    }
    // ---- head: D = X * W^T (logits leave row-major): rows 4 q4 + r of the tile are leaves, columns (lane & 15) neurons
    {
        const uint8_t* const brow = act0 + (size_t)cur * 8 * ROWB + (size_t)r8 * ROWB + q4 * 16;
        const int NT = P.AOP / 16;
#pragma unroll
        for (int k = 0; k < KTH; k += 2) {
            NS_LOAD(A1, row_head + k + 1);
            {   const bf16x8 b_ = *reinterpret_cast<const bf16x8*>(brow + k * 64);
#pragma unroll
                for (int t = 0; t < NTH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_, A0[t], acc[t], 0, 0, 0); }
            if (k + 2 < KTH) NS_LOAD(A0, row_head + k + 2);
            {   const bf16x8 b_ = *reinterpret_cast<const bf16x8*>(brow + (k + 1) * 64);
#pragma unroll
                for (int t = 0; t < NTH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b_, A1[t], acc[t], 0, 0, 0); }
        }
        // acc[t][r] = out[leaf row 4 q4 + r][n = 16 t + (lane & 15)]; rows 0..7 (q4 < 2) are the wave's leaves
        if (q4 < 2) {
            int mrow[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) mrow[r] = slot_of(4 * q4 + r);
#pragma unroll
            for (int t = 0; t < NTH; ++t) {
                if (t < NT) {
                    const int n = 16 * t + (lane & 15);
                    const float bias = P.bias_head[n];
                    if (n < P.A) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int m = mrow[r]; if (m < P.L) P.logits[(size_t)m * P.LGS + n] = acc[t][r] + bias; }
                    } else if (n == P.A) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int m = mrow[r]; if (m < P.L) P.vout[m] = sigmoid_ool(acc[t][r] + bias); }
                    }
                }
            }
        }
    }
#undef NS_STEP
#undef NS_LOAD
}

}  // namespace agz
