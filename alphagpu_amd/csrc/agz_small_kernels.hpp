// agz_small_kernels.hpp — stand-alone pieces of the stepwise API and the getters (one wavefront or workgroup per slot).
#pragma once
#include "agz_device.hpp"

namespace agz {

// softmax!(prior) (mcts_gpu.jl:417) as its own kernel (stepwise mode: agz_rollout_eval): the exponential and the source-order
// sum of the fused path (exact: exp_spec, bf16 mode: exp2_spec)
template <int NR>
static __global__ __launch_bounds__(256) void k_softmax(const float* logits, int LGS, float* prior_eval, int A, int L, int exact) {
    const int lane = lane_id();
    const int slot = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (slot >= L) return;
    float x[NR];
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; x[r] = k < A ? logits[(size_t)slot * LGS + k] : 0.0f; }
    float m = -__builtin_inff();
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; m = (k < A && x[r] > m) ? x[r] : m; }
    m = ufirst(wave_max(m));
    float carry = 0.0f; bool st = false;
    for (int r = 0; r < NR; ++r) {
        int k = 64 * r + lane;
        float e = exact ? exp_spec(x[r] - m) : exp2_spec(x[r] - m);
        x[r] = k < A ? e : 0.0f;
        int nr = A - 64 * r; uint64_t full = nr >= 64 ? ~0ull : ((1ull << nr) - 1ull);
        (void)chain64(x[r], full, carry, false, 0.0f, st);
    }
    const float s = carry;
    for (int r = 0; r < NR; ++r) { int k = 64 * r + lane; if (k < A) prior_eval[(size_t)slot * A + k] = x[r] / s; }
}

// decoder_roots (mcts_gpu.jl:225-246) / decoder for getters: fp32 planes of node `which` (0 = root, else leaf[slot])
static __global__ void k_planes(const Pos* states, const uint32_t* leaf, int use_leaf, int V, int VS, int L, float* out) {
    int slot = blockIdx.x;
    if (slot >= L) return;
    const Pos* s = states + (size_t)slot * V + (use_leaf ? leaf[slot] : 0u);
    for (int j = threadIdx.x; j < 2 * VS; j += blockDim.x) {
        int b = j < VS ? j : j - VS;
        const uint64_t* w = j < VS ? s->p : s->o;
        out[(size_t)slot * 2 * VS + j] = ((w[b >> 6] >> (b & 63)) & 1) ? 1.0f : 0.0f;
    }
}

// visits[:,1,:] and q[:,1,:] of the root as fp32 [L][A] (record layout of agz_tree_eager.hpp: the rank byte of an action leads to
// the edge's entry in the node's list)
static __global__ void k_root_stats(const uint8_t* recs, const uint32_t* meta, int V, uint32_t rec_bytes, uint32_t off_rk, uint32_t off_el,
                                    uint32_t off_vis, int A, int L, float* visits, float* q) {
    int slot = blockIdx.x;
    if (slot >= L) return;
    const uint8_t* rec = recs + (size_t)slot * V * rec_bytes;
    bool expanded = (meta[(size_t)slot * V] & M_EXPANDED) != 0;
    for (int k = threadIdx.x; k < A; k += blockDim.x) {
        float vv = 0.0f, qq = 0.0f;
        const uint32_t rk = expanded ? rec[off_rk + k] : 0u;
        if (rk != 0u) {
            vv = (float)rec[off_vis + rk - 1u];
            qq = reinterpret_cast<const float*>(rec + off_el)[2 * (rk - 1u)];
        }
        if (visits) visits[(size_t)slot * A + k] = vv;
        if (q) q[(size_t)slot * A + k] = qq;
    }
}

}  // namespace agz
