// agz_nn.hpp — batched policy/value forward of snetwork2 (DenseNet.jl:294-304) over the leaf batch.
//
//   b = relu(W0 x);  T x { b = relu(b + relu(Wi b)) };  logits = Wp b + bp;  v = sigma(Wv b + bv)
//
// Two arithmetic modes:
//   * BF16  (throughput): bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate, activations kept in bf16.
//   * EXACT (parity)    : fp32 VALU fma chains in k order starting from 0 — the definition the CPU oracle uses,
//                         so the whole search is bit-identical to the oracle.
#pragma once
#include "agz_device.hpp"

namespace agz {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ uint16_t f2bf(float x) {       // round-to-nearest-even, inputs are finite
    uint32_t u = __float_as_uint(x);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

enum { EPI_RELU = 0, EPI_RES = 1, EPI_HEAD = 2 };

// ---------------------------------------------------------------------------------------------------
// bf16 MFMA layer:  Y[M][N] = epi( X[M][K] * W^T )
//   X   bf16 row-major, leading dimension ldx (elements), K multiple of 32 (zero padded)
//   Wt  bf16 B-operand fragments, pre-tiled on the host: fragment (kt, nt) at ((kt*NT + nt) * 512) elements,
//       lane l element j = W[n = 32 nt + (l & 31)][k = 16 kt + 8 (l >> 5) + j]
//   workgroup = 256 threads = 2 x 2 waves, each wave a 64 x 64 output tile (2 x 2 MFMA 32x32 tiles).
// ---------------------------------------------------------------------------------------------------
constexpr int GB_M = 128, GB_N = 128, GB_K = 32;
constexpr int XS_ROWB = GB_K * 2 + 16;      // 80-byte LDS rows: 16-B slots 5r mod 16 are distinct for 16 consecutive rows

template <int EPI>
__global__ __launch_bounds__(256) void k_layer_bf16(const uint16_t* __restrict__ X, int ldx, int K,
                                                    const uint16_t* __restrict__ Wt, int NT,
                                                    uint16_t* __restrict__ Y, int ldy, int N, int M,
                                                    const float* __restrict__ bias, float* __restrict__ logits, int LGS,
                                                    float* __restrict__ vout, int A) {
    __shared__ __attribute__((aligned(16))) uint8_t xs[GB_M * XS_ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * GB_M, n0 = blockIdx.y * GB_N;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int KT = K / 16;
    for (int k0 = 0; k0 < K; k0 += GB_K) {
        // stage X[m0..m0+128][k0..k0+32] -> LDS (each thread 2 x 16 B)
        __syncthreads();
        for (int c = tid; c < GB_M * 4; c += 256) {
            int row = c >> 2, seg = c & 3;
            int m = m0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < M) v = *reinterpret_cast<const uint4*>(X + (size_t)m * ldx + k0 + seg * 8);
            *reinterpret_cast<uint4*>(xs + row * XS_ROWB + seg * 16) = v;
        }
        __syncthreads();
        for (int s = 0; s < 2; ++s) {
            const int kt = (k0 >> 4) + s;
            if (kt >= KT) break;
            bf16x8 a[2], b[2];
            for (int i = 0; i < 2; ++i) {
                int row = wm * 64 + i * 32 + (lane & 31);
                a[i] = *reinterpret_cast<const bf16x8*>(xs + row * XS_ROWB + s * 32 + (lane >> 5) * 16);
            }
            for (int j = 0; j < 2; ++j) {
                int nt = (n0 >> 5) + wn * 2 + j;
                if (nt < NT) b[j] = *reinterpret_cast<const bf16x8*>(Wt + ((size_t)kt * NT + nt) * 512 + lane * 8);
                else for (int e = 0; e < 8; ++e) b[j][e] = 0;
            }
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // epilogue.  C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (m >= M) continue;
                float x = acc[i][j][e];
                if (EPI == EPI_RELU) {
                    if (n < N) Y[(size_t)m * ldy + n] = f2bf(x > 0.0f ? x : 0.0f);
                } else if (EPI == EPI_RES) {
                    if (n < N) {
                        float r = x > 0.0f ? x : 0.0f;
                        float s = bf2f(X[(size_t)m * ldx + n]) + r;
                        Y[(size_t)m * ldy + n] = f2bf(s > 0.0f ? s : 0.0f);
                    }
                } else {
                    if (n < A) logits[(size_t)m * LGS + n] = x + bias[n];
                    else if (n == A) vout[m] = sigmoid_spec(x + bias[n]);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// EXACT fp32 layer:  y[m][o] = epi( sum_i W[o + O i] * x[m][i] ), k-ordered fma chain from 0.
//   block = 64 x 4 threads: lane -> output o, threadIdx.y -> group of 8 rows; tile = 32 rows staged in LDS.
// ---------------------------------------------------------------------------------------------------
enum { EX_RELU = 0, EX_RES = 1, EX_POLICY = 2, EX_VALUE = 3 };
constexpr int EX_TL = 32;

template <int EPI>
__global__ __launch_bounds__(256) void k_layer_exact(const float* __restrict__ X, int ldx, int K,
                                                     const float* __restrict__ W, int O,
                                                     float* __restrict__ Y, int ldy, int M,
                                                     const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) float xs_dyn[];      // [EX_TL][K4]
    const int K4 = (K + 3) & ~3;
    const int lane = threadIdx.x, g = threadIdx.y, tid = g * 64 + lane;
    const int m0 = blockIdx.x * EX_TL;
    const int o = blockIdx.y * 64 + lane;
    for (int c = tid; c < EX_TL * K4; c += 256) {
        int row = c / K4, i = c - row * K4;
        int m = m0 + row;
        xs_dyn[c] = (m < M && i < K) ? X[(size_t)m * ldx + i] : 0.0f;
    }
    __syncthreads();
    float acc[8];
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const bool live = o < O;
    const float* xr = xs_dyn + (size_t)(g * 8) * K4;
    int i = 0;
    for (; i + 4 <= K; i += 4) {
        float w0 = live ? W[(size_t)o + (size_t)O * i] : 0.0f;
        float w1 = live ? W[(size_t)o + (size_t)O * (i + 1)] : 0.0f;
        float w2 = live ? W[(size_t)o + (size_t)O * (i + 2)] : 0.0f;
        float w3 = live ? W[(size_t)o + (size_t)O * (i + 3)] : 0.0f;
        for (int j = 0; j < 8; ++j) {
            float4 x = *reinterpret_cast<const float4*>(xr + (size_t)j * K4 + i);
            acc[j] = __fmaf_rn(w0, x.x, acc[j]);
            acc[j] = __fmaf_rn(w1, x.y, acc[j]);
            acc[j] = __fmaf_rn(w2, x.z, acc[j]);
            acc[j] = __fmaf_rn(w3, x.w, acc[j]);
        }
    }
    for (; i < K; ++i) {
        float w = live ? W[(size_t)o + (size_t)O * i] : 0.0f;
        for (int j = 0; j < 8; ++j) acc[j] = __fmaf_rn(w, xr[(size_t)j * K4 + i], acc[j]);
    }
    if (!live) return;
    for (int j = 0; j < 8; ++j) {
        int m = m0 + g * 8 + j;
        if (m >= M) continue;
        float x = acc[j];
        if (EPI == EX_RELU) Y[(size_t)m * ldy + o] = x > 0.0f ? x : 0.0f;
        else if (EPI == EX_RES) {
            float r = x > 0.0f ? x : 0.0f;
            float s = X[(size_t)m * ldx + o] + r;
            Y[(size_t)m * ldy + o] = s > 0.0f ? s : 0.0f;
        } else if (EPI == EX_POLICY) Y[(size_t)m * ldy + o] = x + bias[o];
        else Y[(size_t)m * ldy + o] = sigmoid_spec(x + bias[o]);
    }
}

}  // namespace agz
