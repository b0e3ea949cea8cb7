// agz_nn_fused.hpp — the whole snetwork2 forward (DenseNet.jl:294-304) for 128 leaves per workgroup in ONE launch.
//
// Orientation: D = W * X^T on v_mfma_f32_32x32x16_bf16, i.e. the weight fragment is the A operand and the
// activations are the B operand.  The accumulator then holds, per lane, ONE leaf (col = lane & 31) and 16
// neurons in groups of 4 consecutive ones (row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)), so that
//   * a wave owns 32 leaves completely (all H neurons): activations are updated IN PLACE in LDS with no
//     inter-wave hazard and no activation barrier,
//   * the epilogue moves 4 bf16 (8 bytes) per LDS access instead of 16 scattered 2-byte ones.
// Weights are pre-tiled on the host ([kt][nt][lane][8] bf16, the same bytes serve as A fragment of W) and staged
// through LDS in chunks of <= 32 KiB shared by the 4 waves; the next chunk is prefetched into registers while
// the current one feeds the MFMAs.  The input layer reads the bf16 planes straight from global memory.
// LDS = 128 * (2H + 16) + 32 KiB  (66 KiB at H = 128 -> two workgroups per CU).
#pragma once
#include "agz_nn.hpp"

namespace agz {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define AGZ_GLOBAL __attribute__((address_space(1)))

constexpr int F2_M = 128;                 // leaves per workgroup
constexpr int F2_WCHUNK = 32 * 1024;      // bytes of weight fragments staged at a time

#ifdef AGZ_STAMPS
#define NSTAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); nacc[i] += t_ - nlast; nlast = t_; } while (0)
#else
#define NSTAMP(i) do { } while (0)
#endif

struct Fused2Par {
    const uint16_t* planes; int INP;      // [L][INP] bf16, INP % 32 == 0
    const uint16_t* t0; const uint16_t* tres; const uint16_t* thead;
    const float* bias_head;
    float* logits; int LGS; float* vout;
    int L, T, A, AOP;
    unsigned long long* dbg;
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float relu(float x) { return x > 0.0f ? x : 0.0f; }
__device__ __attribute__((noinline)) float sigmoid_call(float x) { return sigmoid_spec(x); }

template <int H>
__global__ __launch_bounds__(256, 2) void k_mlp_fused2(const Fused2Par P) {
    constexpr int NTH = H / 32;
    constexpr int ROWB = H * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const act = smem;                                   // [128][ROWB]
    uint8_t* const wl = smem + (size_t)F2_M * ROWB;              // weight chunk (32 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = wave * 32 + (lane & 31);                    // this lane's leaf inside the tile
    const int m = blockIdx.x * F2_M + lrow;
    const int half = lane >> 5;
    uint8_t* const myrow = act + (size_t)lrow * ROWB;
    const int nlayers = P.T + 2;

    f32x16 acc[NTH];
#ifdef AGZ_STAMPS
    unsigned long long nacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nlast = __builtin_amdgcn_s_memtime();
#endif
    u32x4 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7;               // next weight chunk in flight (32 KiB / 256 threads)

#define AGZ_LAYER_DIMS(ll, KT, NT, w)                                                       \
    do {                                                                                    \
        /* one allocation [layer 0 | residual layers | head]: w is always P.t0 + offset, i.e. a global pointer */ \
        size_t off_;                                                                        \
        if ((ll) == 0) { KT = P.INP / 16; NT = NTH; off_ = 0; }                             \
        else if ((ll) <= P.T) { KT = H / 16; NT = NTH; off_ = (size_t)(P.INP / 16) * NTH * 512 + (size_t)((ll) - 1) * (H / 16) * NTH * 512; } \
        else { KT = H / 16; NT = P.AOP / 32; off_ = (size_t)(P.INP / 16) * NTH * 512 + (size_t)P.T * (H / 16) * NTH * 512; } \
        w = P.t0 + off_;                                                                    \
    } while (0)
#define AGZ_PREFETCH(ll, k0)                                                                \
    do {                                                                                    \
        int KT_, NT_; const uint16_t* w_; AGZ_LAYER_DIMS(ll, KT_, NT_, w_);                 \
        int kc_ = F2_WCHUNK / (NT_ * 1024); if ((k0) + kc_ > KT_) kc_ = KT_ - (k0);         \
        const AGZ_GLOBAL u32x4* src_ = (const AGZ_GLOBAL u32x4*)(w_ + (size_t)(k0) * NT_ * 512);  \
        const int n16_ = kc_ * NT_ * 64;                                                    \
        const u32x4 z_ = {0u, 0u, 0u, 0u};                                                  \
        pf0 = tid < n16_ ? src_[tid] : z_;               pf1 = tid + 256 < n16_ ? src_[tid + 256] : z_;   \
        pf2 = tid + 512 < n16_ ? src_[tid + 512] : z_;   pf3 = tid + 768 < n16_ ? src_[tid + 768] : z_;   \
        pf4 = tid + 1024 < n16_ ? src_[tid + 1024] : z_; pf5 = tid + 1280 < n16_ ? src_[tid + 1280] : z_; \
        pf6 = tid + 1536 < n16_ ? src_[tid + 1536] : z_; pf7 = tid + 1792 < n16_ ? src_[tid + 1792] : z_; \
    } while (0)
#define AGZ_COMMIT()                                                                        \
    do {                                                                                    \
        u32x4* d_ = reinterpret_cast<u32x4*>(wl);                                           \
        d_[tid] = pf0; d_[tid + 256] = pf1; d_[tid + 512] = pf2; d_[tid + 768] = pf3;       \
        d_[tid + 1024] = pf4; d_[tid + 1280] = pf5; d_[tid + 1536] = pf6; d_[tid + 1792] = pf7; \
    } while (0)

    // the input planes of the tile go to LDS once (coalesced 16-B loads, all in flight together); layer 0 then reads its
    // B fragments from there exactly like the hidden layers read theirs
    uint8_t* const pl = wl + F2_WCHUNK;                         // [128][PROWB]
    const int PROWB = P.INP * 2 + 16;
    {
        const int segs = P.INP / 8;
        const AGZ_GLOBAL uint16_t* gp = (const AGZ_GLOBAL uint16_t*)P.planes;
        for (int c = tid; c < F2_M * segs; c += 256) {
            const int row = c / segs, seg = c - row * segs, mm = blockIdx.x * F2_M + row;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (mm < P.L) v = *(const AGZ_GLOBAL u32x4*)(gp + (size_t)mm * P.INP + seg * 8);
            *reinterpret_cast<u32x4*>(pl + (size_t)row * PROWB + seg * 16) = v;
        }
    }
    const uint8_t* const myprow = pl + (size_t)lrow * PROWB;
    AGZ_PREFETCH(0, 0);
    AGZ_COMMIT();
    __syncthreads();
    NSTAMP(0);

    int l = 0, kt0 = 0;
    while (l < nlayers) {
        int KT, NT; const uint16_t* wunused; AGZ_LAYER_DIMS(l, KT, NT, wunused); (void)wunused;
        const int kcmax = F2_WCHUNK / (NT * 1024);
        const int kc = (kt0 + kcmax > KT) ? KT - kt0 : kcmax;
        int nl = l, nk = kt0 + kc;
        if (nk >= KT) { nl = l + 1; nk = 0; }
        if (kt0 == 0) {
#pragma unroll
            for (int t = 0; t < NTH; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        }
        if (nl < nlayers) AGZ_PREFETCH(nl, nk);
        NSTAMP(1);

        // ---- MFMAs of this chunk: acc[t] += Wfrag(c, t) * Xfrag(c); fragments of step c+1 are read while step c computes
        const uint8_t* wlane = wl + lane * 16;
        if (l == 0 || l < nlayers - 1) {                           // input / hidden layers: NT == NTH, D = W * X^T
            const uint8_t* const inrow = l == 0 ? myprow : myrow;
            bf16x8 b, a[NTH];
            b = *reinterpret_cast<const bf16x8*>(inrow + kt0 * 32 + half * 16);
#pragma unroll
            for (int t = 0; t < NTH; ++t) a[t] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)t * 1024);
#pragma unroll 1
            for (int c = 0; c < kc; ++c) {
                bf16x8 bn = b, an[NTH];
#pragma unroll
                for (int t = 0; t < NTH; ++t) an[t] = a[t];
                if (c + 1 < kc) {
                    bn = *reinterpret_cast<const bf16x8*>(inrow + (kt0 + c + 1) * 32 + half * 16);
#pragma unroll
                    for (int t = 0; t < NTH; ++t) an[t] = *reinterpret_cast<const bf16x8*>(wlane + (size_t)((c + 1) * NTH + t) * 1024);
                }
#pragma unroll
                for (int t = 0; t < NTH; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b, acc[t], 0, 0, 0);
                b = bn;
#pragma unroll
                for (int t = 0; t < NTH; ++t) a[t] = an[t];
            }
        } else {                                                   // head: operands swapped (D = X * W^T): logits leave row-major
#pragma unroll 1
            for (int c = 0; c < kc; ++c) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(myrow + (kt0 + c) * 32 + half * 16);
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    if (t < NT) {
                        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wlane + (size_t)(c * NT + t) * 1024);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        NSTAMP(2);
        // ---- epilogue when the layer's K range is complete (only this wave's own rows are touched)
        if (kt0 + kc >= KT) {
            if (l < nlayers - 1) {
                const bool res = l > 0;
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int n = 32 * t + 8 * g + 4 * half;
                        float x0 = relu(acc[t][4 * g]), x1 = relu(acc[t][4 * g + 1]), x2 = relu(acc[t][4 * g + 2]), x3 = relu(acc[t][4 * g + 3]);
                        uint2* dst = reinterpret_cast<uint2*>(myrow + n * 2);
                        if (res) {                                         // b = relu(b + relu(W b))
                            const uint2 o = *dst;
                            x0 = relu(x0 + __uint_as_float(o.x << 16)); x1 = relu(x1 + __uint_as_float(o.x & 0xffff0000u));
                            x2 = relu(x2 + __uint_as_float(o.y << 16)); x3 = relu(x3 + __uint_as_float(o.y & 0xffff0000u));
                        }
                        *dst = make_uint2(pack_bf16x2(x0, x1), pack_bf16x2(x2, x3));
                    }
                }
            } else {
                // head: acc[t][e] = out[leaf = mw + (e&3) + 8(e>>2) + 4 half][n = 32 t + (lane & 31)]
                const int mw = blockIdx.x * F2_M + wave * 32;
#pragma unroll
                for (int t = 0; t < NTH; ++t) {
                    if (t < NT) {
                        const int n = 32 * t + (lane & 31);
                        const float bias = P.bias_head[n];
                        if (n < P.A) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int mm = mw + (e & 3) + 8 * (e >> 2) + 4 * half;
                                if (mm < P.L) P.logits[(size_t)mm * P.LGS + n] = acc[t][e] + bias;
                            }
                        } else if (n == P.A) {
#pragma unroll
                            for (int e = 0; e < 16; ++e) {
                                const int mm = mw + (e & 3) + 8 * (e >> 2) + 4 * half;
                                if (mm < P.L) P.vout[mm] = sigmoid_call(acc[t][e] + bias);
                            }
                        }
                    }
                }
            }
        }
        NSTAMP(3);
        // ---- rotate the weight chunk
        __syncthreads();
        NSTAMP(4);                                                   // all waves finished reading wl
        if (nl < nlayers) AGZ_COMMIT();
        NSTAMP(5);
        __syncthreads();
        NSTAMP(6);
        l = nl; kt0 = nk;
    }
#ifdef AGZ_STAMPS
    if (tid == 0 && P.dbg) for (int i = 0; i < 8; ++i) P.dbg[(size_t)blockIdx.x * 8 + i] += nacc[i];
#endif
#undef AGZ_LAYER_DIMS
#undef AGZ_PREFETCH
#undef AGZ_COMMIT
}

}  // namespace agz
