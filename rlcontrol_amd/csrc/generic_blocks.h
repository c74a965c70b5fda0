// generic_blocks.h -- workgroup-wide fp32 building blocks of the any-shape (VALU) kernels:
// dense forward, backward-to-input, weight-gradient + TF-Adam.  One workgroup of kThreads threads calls each
// block with the same arguments; callers place __syncthreads() between dependent blocks.
#pragma once
#include "rlc_common.h"

namespace gen {

// workgroup size of the any-shape kernels; a translation unit may change it before including this header (each .hip
// is compiled and linked on its own, so the device code of two units never mixes).  512 threads = two waves per SIMD =
// 256 registers per lane: at 1024 threads (128 registers) the four update kernels spilled 90-190 VGPRs beside 290-590
// SGPR spills, the regime in which hipcc 7.2 miscompiled naf_generic.hip (profiles/r03_naf_spill_fault.md); at 512
// rlcontrol_amd/build.py's audit reports no whole-wave spill and no exec-0 restore copy in any of them.
#ifndef RLC_GEN_THREADS
#define RLC_GEN_THREADS 512
#endif
constexpr int kThreads = RLC_GEN_THREADS;
constexpr int kRows = 4;   // batch rows per thread-item in the dense loops

// A fresh, opaque pointer to the kernel-argument segment, typed as the by-value struct T that is the kernel's FIRST
// argument (offset 0).  Why the update kernels read their population view this way, re-made at the start of every phase:
// as a by-value argument every field is an invariant load that the compiler hoists to the kernel's entry and holds in
// SGPRs for the whole launch -- 650 to 900 SGPR spills into VGPR lanes, those VGPRs spilled in turn at the 128-VGPR
// budget of a 1024-thread workgroup -- and hipcc 7.2 miscompiled naf_generic.hip in that regime.
// Through an opaque pointer the scalar loads stay inside the phase that uses them.
template <class T>
__device__ __forceinline__ const T* kernarg_view() {
    unsigned long long k = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    typedef const T __attribute__((address_space(4)))* karg_ptr;
    return (const T*)(karg_ptr)k;
}

typedef float gf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// ---- 4 x 4 register-tile fast paths (taken when shapes are multiples of 4 and the operands 16-byte aligned;
// the element-wise summation order over the contraction index is the same as in the scalar loops below) ----

// Y[b0..+3][n0..+3]: per 4 k one 16-byte read per row of X (wave-uniform) and per row of W (coalesced over n)
__device__ inline void blk_dense_tile4(const float* X, int ldx, int K, const float* E, int Ke, const float* W,
                                       const float* bias, int N, float* Y, int ldy, int B, int act) {
    const int rb = (B + 3) >> 2, nb = N >> 2, K4 = K & ~3;
    for (int it = threadIdx.x; it < rb * nb; it += kThreads) {
        const int n0 = (it % nb) << 2;
        const int b0 = (it / nb) << 2;
        int br[4];
#pragma unroll
        for (int j = 0; j < 4; j++) br[j] = min(b0 + j, B - 1);
        gf4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] = gf4{0.f, 0.f, 0.f, 0.f};
        // operands of the next 4 k are loaded before the FMAs of the current ones (activations and weights come
        // from L2 / HBM and only four waves share a CU: the loads would otherwise sit on the critical path)
        gf4 x[4], w[4], xn[4], wn[4];
        auto load = [&](gf4 (&xx)[4], gf4 (&ww)[4], int k) {
#pragma unroll
            for (int j = 0; j < 4; j++) xx[j] = *reinterpret_cast<const gf4*>(&X[(size_t)br[j] * ldx + k]);
#pragma unroll
            for (int i = 0; i < 4; i++) ww[i] = *reinterpret_cast<const gf4*>(&W[(size_t)(k + i) * N + n0]);
        };
        if (K4 > 0) load(x, w, 0);
        for (int k = 0; k < K4; k += 4) {
            if (k + 4 < K4) load(xn, wn, k + 4);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] += x[j][i] * w[i];
#pragma unroll
            for (int j = 0; j < 4; j++) { x[j] = xn[j]; w[j] = wn[j]; }
        }
        for (int k = K4; k < K; k++) {
            const gf4 w = *reinterpret_cast<const gf4*>(&W[(size_t)k * N + n0]);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += X[(size_t)br[j] * ldx + k] * w;
        }
        for (int e = 0; e < Ke; e++) {
            const gf4 w = *reinterpret_cast<const gf4*>(&W[(size_t)(K + e) * N + n0]);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] += E[br[j] * Ke + e] * w;
        }
        const gf4 bs = *reinterpret_cast<const gf4*>(&bias[n0]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (b0 + j < B) {
                gf4 v = acc[j] + bs;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (act == 1) v[c] = fmaxf(v[c], 0.0f);
                    else if (act == 2) v[c] = tanhf(v[c]);
                    Y[(size_t)(b0 + j) * ldy + n0 + c] = v[c];
                }
            }
        }
    }
}

// dX[b0..+3][k0..+3] = sum_n dY[b][n] W[k][n]: per 4 n one 16-byte read per row of dY and per row of W
__device__ inline void blk_bwd_input_tile4(const float* dY, int lddy, int N, const float* W, const float* Hk, int K,
                                           float* dX, int B, bool accumulate) {
    const int rb = (B + 3) >> 2, kb = K >> 2;
    for (int it = threadIdx.x; it < rb * kb; it += kThreads) {
        const int k0 = (it % kb) << 2;
        const int b0 = (it / kb) << 2;
        int br[4];
#pragma unroll
        for (int j = 0; j < 4; j++) br[j] = min(b0 + j, B - 1);
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) acc[j][i] = 0.0f;
        gf4 d[4], w[4], dn[4], wn[4];
        auto load = [&](gf4 (&dd)[4], gf4 (&ww)[4], int n) {
#pragma unroll
            for (int j = 0; j < 4; j++) dd[j] = *reinterpret_cast<const gf4*>(&dY[(size_t)br[j] * lddy + n]);
#pragma unroll
            for (int i = 0; i < 4; i++) ww[i] = *reinterpret_cast<const gf4*>(&W[(size_t)(k0 + i) * N + n]);
        };
        load(d, w, 0);
        for (int n = 0; n < N; n += 4) {
            if (n + 4 < N) load(dn, wn, n + 4);
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) acc[j][i] += d[j][c] * w[i][c];
#pragma unroll
            for (int j = 0; j < 4; j++) { d[j] = dn[j]; w[j] = wn[j]; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (b0 + j < B)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const size_t p = (size_t)(b0 + j) * K + k0 + i;
                    const float v = (Hk == nullptr || Hk[p] > 0.0f) ? acc[j][i] : 0.0f;
                    dX[p] = accumulate ? dX[p] + v : v;
                }
    }
}

// Y[b,n] = act( sum_k X[b,k] W[k,n] + sum_j E[b,j] W[K+j,n] + bias[n] );  act: 0 none, 1 relu, 2 tanh
// X: [B,K] row-major (ldx); E: optional extra input columns (the action, concatenated LAST:
// hydra_ddpg_network.py:128).  Lanes run over n (coalesced W rows); X/E reads are wave-uniform.
__device__ inline void blk_dense(const float* X, int ldx, int K, const float* E, int Ke, const float* W,
                          const float* bias, int N, float* Y, int ldy, int B, int act) {
    if ((N & 3) == 0 && (ldx & 3) == 0 && K >= 4 && al16(X) && al16(W) && al16(bias)) {
        blk_dense_tile4(X, ldx, K, E, Ke, W, bias, N, Y, ldy, B, act);
        return;
    }
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

// dX[b,k] (+)= sum_n dY[b,n] W[k,n] (W[k][n] rows k < K only), masked by (Hk[b,k] > 0) unless Hk is null; lddy = leading
// dimension of dY (heads that write into a strided slot array); accumulate = add into dX.  One implementation for
// every caller: the 4 x 4 register-tile fast path when shapes and alignment allow, element-wise otherwise (same
// summation order over n).
__device__ inline void blk_dense_bwd_input_ld(const float* dY, int lddy, int N, const float* W, const float* Hk, int K,
                                              float* dX, int B, bool accumulate) {
    if ((N & 3) == 0 && (K & 3) == 0 && (lddy & 3) == 0 && al16(dY) && al16(W)) {
        blk_bwd_input_tile4(dY, lddy, N, W, Hk, K, dX, B, accumulate);
        return;
    }
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
// dense dY, always masked by Hk > 0, overwrite
__device__ inline void blk_dense_bwd_input(const float* dY, int N, const float* W, const float* Hk, int K, float* dX, int B) {
    blk_dense_bwd_input_ld(dY, N, N, W, Hk, K, dX, B, false);
}
// dense dY, optional mask, optional accumulation
__device__ inline void blk_dense_bwd_input_ex(const float* dY, int N, const float* W, const float* Hk, int K, float* dX,
                                              int B, bool accumulate) {
    blk_dense_bwd_input_ld(dY, N, N, W, Hk, K, dX, B, accumulate);
}

// ---- tf.contrib.layers.layer_norm (agents/network/base_network.py:53-56) on the rows of a [B, N] activation ----
// One wave per row, lane l holds features l, l+64, ...; row sums go through a fixed shuffle tree (deterministic).
#define RLC_LN_EPS 1e-12f
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Z (in: x.W + b, out: relu(gamma * nhat + beta)), nhat = (z - mean) * rsqrt(var + eps) and rstd kept for the
// backward pass when the pointers are non-null
__device__ inline void blk_layernorm_relu(float* Z, int N, int B, const float* beta, const float* gamma, float* nhat,
                                          float* rstd) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kThreads >> 6;
    for (int r = wave; r < B; r += nw) {
        float* z = Z + (size_t)r * N;
        float s = 0.0f;
        for (int n = lane; n < N; n += 64) s += z[n];
        const float mean = wave_sum64(s) / (float)N;
        float q = 0.0f;
        for (int n = lane; n < N; n += 64) { const float c = z[n] - mean; q += c * c; }
        const float rs = 1.0f / sqrtf(wave_sum64(q) / (float)N + RLC_LN_EPS);
        if (rstd && lane == 0) rstd[r] = rs;
        for (int n = lane; n < N; n += 64) {
            const float nh = (z[n] - mean) * rs;
            if (nhat) nhat[(size_t)r * N + n] = nh;
            z[n] = fmaxf(nh * gamma[n] + beta[n], 0.0f);
        }
    }
}
// dY [B, N] = gradient w.r.t. gamma * nhat + beta (relu mask applied).  Thread n < N returns the column sums that
// are the gradients of gamma[n] and beta[n] (call before blk_layernorm_bwd_rows overwrites dY).
__device__ inline void blk_layernorm_param_grads(const float* dY, const float* nhat, int N, int B, float& g_gamma,
                                                 float& g_beta) {
    g_gamma = 0.0f; g_beta = 0.0f;
    const int n = threadIdx.x;
    if (n < N)
        for (int b = 0; b < B; b++) {
            const float t = dY[(size_t)b * N + n];
            g_gamma += t * nhat[(size_t)b * N + n];
            g_beta += t;
        }
}
// dY -> gradient w.r.t. the layer's linear output, row by row: rstd * (g - mean(g) - nhat * mean(g * nhat)), g = dY * gamma
__device__ inline void blk_layernorm_bwd_rows(float* dY, const float* nhat, const float* rstd, const float* gamma, int N,
                                              int B) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = kThreads >> 6;
    for (int r = wave; r < B; r += nw) {
        float* dy = dY + (size_t)r * N;
        const float* nh = nhat + (size_t)r * N;
        float s1 = 0.0f, s2 = 0.0f;
        for (int n = lane; n < N; n += 64) { const float gq = dy[n] * gamma[n]; s1 += gq; s2 += gq * nh[n]; }
        const float m1 = wave_sum64(s1) / (float)N, m2 = wave_sum64(s2) / (float)N, rs = rstd[r];
        for (int n = lane; n < N; n += 64) dy[n] = rs * (dy[n] * gamma[n] - m1 - nh[n] * m2);
    }
}

struct AdamCtx {
    float* theta; float* m; float* v; float alpha; float* tap;   // tap may be null
    float eps = 1e-8f;   // TF's epsilon; torch's Adam (the KL agents) folds its bias correction into alpha and eps
};

// block-wide sum of v over threads (fixed order: deterministic); result broadcast to all threads
__device__ inline float blk_sum(float v, float* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < kThreads / 64; w++) s += red[w];
    __syncthreads();
    return s;
}

__device__ __forceinline__ void adam_apply(const AdamCtx& c, int p, float g) {
    float m = c.m[p], v = c.v[p];
    const float nv = adam_step_eps(c.theta[p], g, m, v, c.alpha, c.eps);
    c.m[p] = m; c.v[p] = v; c.theta[p] = nv;
    if (c.tap) c.tap[p] = g;
}

// W[k0..+3][n0..+3] gradient + Adam: per batch row one 16-byte read of X (wave-uniform) and of dY (coalesced)
__device__ inline void blk_grad_adam_tile4(const float* X, int ldx, int K, const float* dY, int lddy, int N, int B,
                                           const AdamCtx& c, int oW) {
    const int kb = K >> 2, nb = N >> 2;
    for (int it = threadIdx.x; it < kb * nb; it += kThreads) {
        const int n0 = (it % nb) << 2;
        const int k0 = (it / nb) << 2;
        gf4 g[4];
#pragma unroll
        for (int i = 0; i < 4; i++) g[i] = gf4{0.f, 0.f, 0.f, 0.f};
        // four batch rows in flight per thread (8 loads issued before their FMAs)
        int b = 0;
        for (; b + 4 <= B; b += 4) {
            gf4 x[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                x[q] = *reinterpret_cast<const gf4*>(&X[(size_t)(b + q) * ldx + k0]);
                d[q] = *reinterpret_cast<const gf4*>(&dY[(size_t)(b + q) * lddy + n0]);
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < 4; i++) g[i] += x[q][i] * d[q];
        }
        for (; b < B; b++) {
            const gf4 x = *reinterpret_cast<const gf4*>(&X[(size_t)b * ldx + k0]);
            const gf4 d = *reinterpret_cast<const gf4*>(&dY[(size_t)b * lddy + n0]);
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] += x[i] * d;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int p = oW + (k0 + i) * N + n0;
            gf4 th = *reinterpret_cast<const gf4*>(&c.theta[p]);
            gf4 m = *reinterpret_cast<const gf4*>(&c.m[p]);
            gf4 v = *reinterpret_cast<const gf4*>(&c.v[p]);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float mm = m[q], vv = v[q];
                th[q] = adam_step_eps(th[q], g[i][q], mm, vv, c.alpha, c.eps);
                m[q] = mm; v[q] = vv;
            }
            *reinterpret_cast<gf4*>(&c.m[p]) = m;
            *reinterpret_cast<gf4*>(&c.v[p]) = v;
            *reinterpret_cast<gf4*>(&c.theta[p]) = th;
            if (c.tap) *reinterpret_cast<gf4*>(&c.tap[p]) = g[i];
        }
    }
}

// gradient of a dense layer's weights/bias + Adam, one parameter element per thread-item:
//   W[k,n] (k < K): sum_b X[b,k] dY[b,n];  W[K+j,n]: sum_b E[b,j] dY[b,n];  bias[n]: sum_b dY[b,n]
__device__ inline void blk_dense_grad_adam(const float* X, int ldx, int K, const float* E, int Ke, const float* dY,
                                    int N, int B, const AdamCtx& c, int oW, int ob) {
    const int rows = K + Ke + 1;   // last "row" is the bias
    int first = 0;
    if ((N & 3) == 0 && (K & 3) == 0 && (ldx & 3) == 0 && (oW & 3) == 0 && al16(X) && al16(dY) && al16(c.theta) &&
        al16(c.m) && al16(c.v) && (c.tap == nullptr || al16(c.tap))) {
        blk_grad_adam_tile4(X, ldx, K, dY, N, N, B, c, oW);
        first = K * N;             // the action rows and the bias row stay on the element-wise path
    }
    for (int it = first + threadIdx.x; it < rows * N; it += kThreads) {
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


// ---- variant with an explicit leading dimension of dY (heads that write into a strided slot array) ----
__device__ inline void blk_dense_grad_adam_ld(const float* X, int ldx, int K, const float* dY, int lddy, int N, int B,
                                              const AdamCtx& c, int oW, int ob) {
    const int rows = K + 1;   // last "row" is the bias
    int first = 0;
    if ((N & 3) == 0 && (K & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && (oW & 3) == 0 && al16(X) && al16(dY) &&
        al16(c.theta) && al16(c.m) && al16(c.v) && (c.tap == nullptr || al16(c.tap))) {
        blk_grad_adam_tile4(X, ldx, K, dY, lddy, N, B, c, oW);
        first = K * N;
    }
    for (int it = first + threadIdx.x; it < rows * N; it += kThreads) {
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
