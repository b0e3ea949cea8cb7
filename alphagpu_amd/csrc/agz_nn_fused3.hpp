// agz_nn_fused3.hpp — the whole snetwork2 forward (DenseNet.jl:294-304) for 128 leaves per workgroup in ONE launch,
// third layout: v_mfma_f32_16x16x32_bf16, 8 waves per workgroup, each wave owns 16 leaves completely.
//
// Measured on an earlier 32x32x16 / 4-wave version (removed): one wave per SIMD, so every LDS->MFMA dependency and
// the whole bf16 epilogue (44 % of the kernel) sat exposed; 46 us for 8.5 GFLOP.  Here two waves share a SIMD and
// overlap each other's MFMA and epilogue phases, and a wave's epilogue is half as long (8 groups of 4 neurons).
//
// Orientation D = W * X^T: A operand = weight fragment (16 neurons x 32 k), B operand = activations (32 k x 16 leaves).
// C layout of 16x16: col = lane & 15 (leaf), row = 4 (lane >> 4) + reg (neuron) -> a lane holds 4 consecutive neurons
// of one leaf per tile: activations are updated in place in LDS, 8 bytes per access, no inter-wave hazard.
// The head layer swaps the operands (D = X * W^T) so that logits leave the accumulator row-major.
// Weight fragments are pre-tiled on the host: tile (kt, nt) at ((kt*NT + nt) * 512) elements, lane l element j =
// W[n = 16 nt + (l & 15)][k = 32 kt + 8 (l >> 4) + j]; staged through LDS in <= 32 KiB chunks, next chunk prefetched.
#pragma once
#include "agz_nn.hpp"

namespace agz {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define AGZ_GLB __attribute__((address_space(1)))

constexpr int F3_M = 128;                 // leaves per workgroup
constexpr int F3_THREADS = 512;
constexpr int F3_WCHUNK = 32 * 1024;

struct Fused3Par {
    const uint16_t* planes; int INP;      // [L][INP] bf16, INP % 32 == 0
    const uint16_t* w16;                  // [layer 0 | T residual layers | head] fragments (16x16x32 tiling)
    const float* bias_head;
    float* logits; int LGS; float* vout;
    int L, T, A, AOP;
    int gpw, tw;                          // k_search_small with sparse waves: row r of a workgroup's tile is game slot
                                          // (bidx*tw + r/8)*gpw + r%8 if r%8 < gpw (gpw = 0: rows are consecutive leaves)
};

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __attribute__((noinline)) float sigmoid_ool(float x) { return sigmoid_spec(x); }

template <int H>
__global__ __launch_bounds__(F3_THREADS, 2) void k_mlp_fused3(const Fused3Par P) {
    constexpr int NTH = H / 16;                                  // 16-neuron tiles of a hidden layer
    constexpr int ROWB = H * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const act = smem;                                   // [128][ROWB]
    uint8_t* const wl = smem + (size_t)F3_M * ROWB;              // weight chunk (32 KiB)
    uint8_t* const pl = wl + F3_WCHUNK;                          // input planes tile [128][PROWB]
    const int PROWB = P.INP * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = wave * 16 + (lane & 15);                    // this lane's leaf inside the tile
    const int q4 = lane >> 4;                                    // k-quarter (operands) / neuron quad (accumulator)
    uint8_t* const myrow = act + (size_t)lrow * ROWB;
    const uint8_t* const myprow = pl + (size_t)lrow * PROWB;
    const int nlayers = P.T + 2;

    f32x4 acc[NTH];
    v4u pf0, pf1, pf2, pf3;                                      // next weight chunk in flight (32 KiB / 512 threads)

#define F3_DIMS(ll, KT, NT, off)                                                                        \
    do {                                                                                                \
        if ((ll) == 0) { KT = P.INP / 32; NT = NTH; off = 0; }                                          \
        else if ((ll) <= P.T) { KT = H / 32; NT = NTH; off = (size_t)(P.INP / 32) * NTH * 512 + (size_t)((ll) - 1) * (H / 32) * NTH * 512; } \
        else { KT = H / 32; NT = P.AOP / 16; off = (size_t)(P.INP / 32) * NTH * 512 + (size_t)P.T * (H / 32) * NTH * 512; } \
    } while (0)
#define F3_PREFETCH(ll, k0)                                                                             \
    do {                                                                                                \
        int KT_, NT_; size_t off_; F3_DIMS(ll, KT_, NT_, off_);                                         \
        int kc_ = F3_WCHUNK / (NT_ * 1024); if ((k0) + kc_ > KT_) kc_ = KT_ - (k0);                     \
        const AGZ_GLB v4u* src_ = (const AGZ_GLB v4u*)(P.w16 + off_ + (size_t)(k0) * NT_ * 512);        \
        const int n16_ = kc_ * NT_ * 64;                                                                \
        const v4u z_ = {0u, 0u, 0u, 0u};                                                                \
        pf0 = tid < n16_ ? src_[tid] : z_;               pf1 = tid + 512 < n16_ ? src_[tid + 512] : z_; \
        pf2 = tid + 1024 < n16_ ? src_[tid + 1024] : z_; pf3 = tid + 1536 < n16_ ? src_[tid + 1536] : z_; \
    } while (0)
#define F3_COMMIT()                                                                                     \
    do {                                                                                                \
        v4u* d_ = reinterpret_cast<v4u*>(wl);                                                           \
        d_[tid] = pf0; d_[tid + 512] = pf1; d_[tid + 1024] = pf2; d_[tid + 1536] = pf3;                 \
    } while (0)

    {   // input planes of the tile -> LDS (coalesced 16-B loads)
        const int segs = P.INP / 8;
        const AGZ_GLB uint16_t* gp = (const AGZ_GLB uint16_t*)P.planes;
        for (int c = tid; c < F3_M * segs; c += F3_THREADS) {
            const int row = c / segs, seg = c - row * segs, mm = blockIdx.x * F3_M + row;
            v4u v = {0u, 0u, 0u, 0u};
            if (mm < P.L) v = *(const AGZ_GLB v4u*)(gp + (size_t)mm * P.INP + seg * 8);
            *reinterpret_cast<v4u*>(pl + (size_t)row * PROWB + seg * 16) = v;
        }
    }
    F3_PREFETCH(0, 0);
    F3_COMMIT();
    __syncthreads();

    int l = 0, kt0 = 0;
    while (l < nlayers) {
        int KT, NT; size_t offu; F3_DIMS(l, KT, NT, offu); (void)offu;
        const int kcmax = F3_WCHUNK / (NT * 1024);
        const int kc = (kt0 + kcmax > KT) ? KT - kt0 : kcmax;
        int nl = l, nk = kt0 + kc;
        if (nk >= KT) { nl = l + 1; nk = 0; }
        if (kt0 == 0) {
#pragma unroll
            for (int t = 0; t < NTH; ++t) { acc[t][0] = 0.0f; acc[t][1] = 0.0f; acc[t][2] = 0.0f; acc[t][3] = 0.0f; }
        }
        if (nl < nlayers) F3_PREFETCH(nl, nk);

        const uint8_t* wlane = wl + lane * 16;
        const uint8_t* const inrow = l == 0 ? myprow : myrow;
        if (l < nlayers - 1) {                                     // input / hidden layers: D = W * X^T
            bf16x8 b = *reinterpret_cast<const bf16x8*>(inrow + kt0 * 64 + q4 * 16);
#pragma unroll 1
            for (int c = 0; c < kc; ++c) {
                bf16x8 a[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) a[t] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(c * NTH + t) * 1024);
                bf16x8 bn = b;
                if (c + 1 < kc) bn = *reinterpret_cast<const bf16x8*>(inrow + (kt0 + c + 1) * 64 + q4 * 16);
#pragma unroll
                for (int t = 0; t < NTH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b, acc[t], 0, 0, 0);
                b = bn;
            }
        } else {                                                   // head: D = X * W^T (logits leave row-major)
#pragma unroll 1
            for (int c = 0; c < kc; ++c) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(myrow + (kt0 + c) * 64 + q4 * 16);
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    if (t < NT) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(c * NT + t) * 1024);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (kt0 + kc >= KT) {                                      // epilogue of the layer (only this wave's own rows)
            if (l < nlayers - 1) {
                const bool res = l > 0;
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    const int n = 16 * t + 4 * q4;                  // acc[t][r] = out[neuron n + r][leaf lrow]
                    float x0 = acc[t][0] > 0.0f ? acc[t][0] : 0.0f, x1 = acc[t][1] > 0.0f ? acc[t][1] : 0.0f;
                    float x2 = acc[t][2] > 0.0f ? acc[t][2] : 0.0f, x3 = acc[t][3] > 0.0f ? acc[t][3] : 0.0f;
                    uint2* dst = reinterpret_cast<uint2*>(myrow + n * 2);
                    if (res) {                                     // b = relu(b + relu(W b))
                        const uint2 o = *dst;
                        x0 += __uint_as_float(o.x << 16); x1 += __uint_as_float(o.x & 0xffff0000u);
                        x2 += __uint_as_float(o.y << 16); x3 += __uint_as_float(o.y & 0xffff0000u);
                        x0 = x0 > 0.0f ? x0 : 0.0f; x1 = x1 > 0.0f ? x1 : 0.0f; x2 = x2 > 0.0f ? x2 : 0.0f; x3 = x3 > 0.0f ? x3 : 0.0f;
                    }
                    *dst = make_uint2(pk_bf16(x0, x1), pk_bf16(x2, x3));
                }
            } else {
                // head: acc[t][r] = out[leaf = 16 wave + 4 q4 + r][n = 16 t + (lane & 15)]
                const int mw = blockIdx.x * F3_M + wave * 16 + 4 * q4;
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    if (t < NT) {
                        const int n = 16 * t + (lane & 15);
                        const float bias = P.bias_head[n];
                        if (n < P.A) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (mw + r < P.L) P.logits[(size_t)(mw + r) * P.LGS + n] = acc[t][r] + bias;
                        } else if (n == P.A) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) if (mw + r < P.L) P.vout[mw + r] = sigmoid_ool(acc[t][r] + bias);
                        }
                    }
                }
            }
        }
        __syncthreads();                                           // all waves finished reading wl
        if (nl < nlayers) F3_COMMIT();
        __syncthreads();
        l = nl; kt0 = nk;
    }
#undef F3_DIMS
#undef F3_PREFETCH
#undef F3_COMMIT
}

}  // namespace agz
