// generic_blocks.h -- workgroup-wide fp32 building blocks of the any-shape (VALU) kernels:
// dense forward, backward-to-input, weight-gradient + TF-Adam.  One workgroup of kThreads threads calls each
// block with the same arguments; callers place __syncthreads() between dependent blocks.
#pragma once
#include "rlc_common.h"

namespace gen {

constexpr int kThreads = 256;
constexpr int kRows = 4;   // batch rows per thread-item in the dense loops

// Y[b,n] = act( sum_k X[b,k] W[k,n] + sum_j E[b,j] W[K+j,n] + bias[n] );  act: 0 none, 1 relu, 2 tanh
// X: [B,K] row-major (ldx); E: optional extra input columns (the action, concatenated LAST:
// hydra_ddpg_network.py:128).  Lanes run over n (coalesced W rows); X/E reads are wave-uniform.
__device__ inline void blk_dense(const float* X, int ldx, int K, const float* E, int Ke, const float* W,
                          const float* bias, int N, float* Y, int ldy, int B, int act) {
    const int rb = (B + kRows - 1) / kRows;
    for (int it = threadIdx.x; it < rb * N; it += kThreads) {
        const int n = it % N;
        const int b0 = (it / N) * kRows;
        float acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; i++) acc[i] = 0.0f;
        for (int k = 0; k < K; k++) {
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < kRows; i++) {
                const int b = min(b0 + i, B - 1);
                acc[i] += X[(size_t)b * ldx + k] * w;
            }
        }
        for (int j = 0; j < Ke; j++) {
            const float w = W[(size_t)(K + j) * N + n];
#pragma unroll
            for (int i = 0; i < kRows; i++) {
                const int b = min(b0 + i, B - 1);
                acc[i] += E[b * Ke + j] * w;
            }
        }
        const float bs = bias[n];
#pragma unroll
        for (int i = 0; i < kRows; i++) {
            if (b0 + i < B) {
                float v = acc[i] + bs;
                if (act == 1) v = fmaxf(v, 0.0f);
                else if (act == 2) v = tanhf(v);
                Y[(size_t)(b0 + i) * ldy + n] = v;
            }
        }
    }
}

// dX[b,k] = (Hk[b,k] > 0) ? sum_n dY[b,n] W[k,n] : 0      (W[k][n] rows k < K only)
__device__ inline void blk_dense_bwd_input(const float* dY, int N, const float* W, const float* Hk, int K, float* dX,
                                    int B) {
    const int rb = (B + kRows - 1) / kRows;
    for (int it = threadIdx.x; it < rb * K; it += kThreads) {
        const int k = it % K;
        const int b0 = (it / K) * kRows;
        float acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; i++) acc[i] = 0.0f;
        for (int n = 0; n < N; n++) {
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < kRows; i++) {
                const int b = min(b0 + i, B - 1);
                acc[i] += dY[(size_t)b * N + n] * w;
            }
        }
#pragma unroll
        for (int i = 0; i < kRows; i++)
            if (b0 + i < B) dX[(size_t)(b0 + i) * K + k] = Hk[(size_t)(b0 + i) * K + k] > 0.0f ? acc[i] : 0.0f;
    }
}

struct AdamCtx {
    float* theta; float* m; float* v; float alpha; float* tap;   // tap may be null
};

__device__ __forceinline__ void adam_apply(const AdamCtx& c, int p, float g) {
    float m = c.m[p], v = c.v[p];
    const float nv = adam_step(c.theta[p], g, m, v, c.alpha);
    c.m[p] = m; c.v[p] = v; c.theta[p] = nv;
    if (c.tap) c.tap[p] = g;
}

// gradient of a dense layer's weights/bias + Adam, one parameter element per thread-item:
//   W[k,n] (k < K): sum_b X[b,k] dY[b,n];  W[K+j,n]: sum_b E[b,j] dY[b,n];  bias[n]: sum_b dY[b,n]
__device__ inline void blk_dense_grad_adam(const float* X, int ldx, int K, const float* E, int Ke, const float* dY,
                                    int N, int B, const AdamCtx& c, int oW, int ob) {
    const int rows = K + Ke + 1;   // last "row" is the bias
    for (int it = threadIdx.x; it < rows * N; it += kThreads) {
        const int n = it % N;
        const int k = it / N;
        float g = 0.0f;
        if (k < K) {
            for (int b = 0; b < B; b++) g += X[(size_t)b * ldx + k] * dY[(size_t)b * N + n];
            adam_apply(c, oW + k * N + n, g);
        } else if (k < K + Ke) {
            for (int b = 0; b < B; b++) g += E[b * Ke + (k - K)] * dY[(size_t)b * N + n];
            adam_apply(c, oW + k * N + n, g);
        } else {
            for (int b = 0; b < B; b++) g += dY[(size_t)b * N + n];
            adam_apply(c, ob + n, g);
        }
    }
}


// dX[b,k] (+)= sum_n dY[b,n] W[k,n], optionally masked by (Hk[b,k] > 0); accumulate = add into dX
__device__ inline void blk_dense_bwd_input_ex(const float* dY, int N, const float* W, const float* Hk, int K, float* dX,
                                              int B, bool accumulate) {
    const int rb = (B + kRows - 1) / kRows;
    for (int it = threadIdx.x; it < rb * K; it += kThreads) {
        const int k = it % K;
        const int b0 = (it / K) * kRows;
        float acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; i++) acc[i] = 0.0f;
        for (int n = 0; n < N; n++) {
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < kRows; i++) {
                const int b = min(b0 + i, B - 1);
                acc[i] += dY[(size_t)b * N + n] * w;
            }
        }
#pragma unroll
        for (int i = 0; i < kRows; i++)
            if (b0 + i < B) {
                const size_t p = (size_t)(b0 + i) * K + k;
                const float v = (Hk == nullptr || Hk[p] > 0.0f) ? acc[i] : 0.0f;
                dX[p] = accumulate ? dX[p] + v : v;
            }
    }
}

// ---- variants with an explicit leading dimension of dY (heads that write into a strided slot array) ----
__device__ inline void blk_dense_bwd_input_ld(const float* dY, int lddy, int N, const float* W, const float* Hk, int K,
                                              float* dX, int B, bool accumulate) {
    const int rb = (B + kRows - 1) / kRows;
    for (int it = threadIdx.x; it < rb * K; it += kThreads) {
        const int k = it % K;
        const int b0 = (it / K) * kRows;
        float acc[kRows];
#pragma unroll
        for (int i = 0; i < kRows; i++) acc[i] = 0.0f;
        for (int n = 0; n < N; n++) {
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int i = 0; i < kRows; i++) {
                const int b = min(b0 + i, B - 1);
                acc[i] += dY[(size_t)b * lddy + n] * w;
            }
        }
#pragma unroll
        for (int i = 0; i < kRows; i++)
            if (b0 + i < B) {
                const size_t p = (size_t)(b0 + i) * K + k;
                const float v = (Hk == nullptr || Hk[p] > 0.0f) ? acc[i] : 0.0f;
                dX[p] = accumulate ? dX[p] + v : v;
            }
    }
}

__device__ inline void blk_dense_grad_adam_ld(const float* X, int ldx, int K, const float* dY, int lddy, int N, int B,
                                              const AdamCtx& c, int oW, int ob) {
    const int rows = K + 1;   // last "row" is the bias
    for (int it = threadIdx.x; it < rows * N; it += kThreads) {
        const int n = it % N;
        const int k = it / N;
        float g = 0.0f;
        if (k < K) {
            for (int b = 0; b < B; b++) g += X[(size_t)b * ldx + k] * dY[(size_t)b * lddy + n];
            adam_apply(c, oW + k * N + n, g);
        } else {
            for (int b = 0; b < B; b++) g += dY[(size_t)b * lddy + n];
            adam_apply(c, ob + n, g);
        }
    }
}

}  // namespace gen
