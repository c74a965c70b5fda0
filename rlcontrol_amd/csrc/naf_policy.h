// naf_policy.h -- B=1 forward of the NAF policy heads (greedy action mu and the columns of L) and the exploration
// sample of agents/network/naf_network.py:152-176, shared by the acting kernel, the on-device train step and the
// on-device evaluation (naf_generic.hip).
//
// Exploration: the reference draws rng.multivariate_normal(mu, noise_scale * pinv(L L^T)).  L is lower triangular
// with a strictly positive diagonal (exp of a clipped pre-activation), so L L^T is invertible and the draw is
//   mu + sqrt(noise_scale) * L^-T z,   z ~ N(0, I)
// (covariance noise_scale * L^-T L^-1 = noise_scale * (L L^T)^-1): one back-substitution with L^T instead of an SVD.
// numpy maps z to the sample through a different factor of the same covariance, and the device z comes from Philox:
// same distribution, different numbers (statistical parity, as for every device random stream).
#pragma once
#include "generic_blocks.h"
#include "naf_common.h"

#ifdef __HIPCC__

#define RLC_KEY_NAF_EPS 0x4E41465F4E4F4953ull

// LDS floats: x | h1 | ha | z[MAX_A] | dpre[MAX_A] | npre[NN] | out[MAX_A]
__host__ __device__ inline size_t naf_policy_lds_floats(const RlcNafDims& d) {
    return (size_t)((d.S + 3) & ~3) + ((d.L1 + 3) & ~3) + ((d.L2 + 3) & ~3) + 3 * RLC_NAF_MAX_A + (d.NN > 0 ? d.NN : 1) + 4;
}
struct NafPolicyLds { float *x, *h1, *ha, *z, *dpre, *npre, *out; };
__device__ inline NafPolicyLds naf_policy_carve(const RlcNafDims& d, float* base) {
    NafPolicyLds L;
    L.x = base;
    L.h1 = L.x + ((d.S + 3) & ~3);
    L.ha = L.h1 + ((d.L1 + 3) & ~3);
    L.z = L.ha + ((d.L2 + 3) & ~3);
    L.dpre = L.z + RLC_NAF_MAX_A;
    L.npre = L.dpre + RLC_NAF_MAX_A;
    L.out = L.npre + (d.NN > 0 ? d.NN : 1);
    return L;
}

// L.x = clipped state.  On return L.out[j] = mu_j = tanh(.) * a_max[j]; L.dpre / L.npre hold the pre-activations of
// the L columns.  One workgroup of gen::kThreads threads; trailing barrier included.
__device__ inline void naf_policy_forward(const RlcNafDims& d, const float* th, const NafPolicyLds& L, const float* amax) {
    using namespace gen;
    const int S = d.S, A = d.A, L1 = d.L1, L2 = d.L2, NN = d.NN;
    const int act = d.norm ? 0 : 1;     // norm_type 'layer': relu after the normalisation (never with the blocked layout)
    __syncthreads();
    blk_dense(L.x, S, S, nullptr, 0, th + d.W1, th + d.b1, L1, L.h1, L1, 1, act);
    __syncthreads();
    if (d.norm) {
        blk_layernorm_relu(L.h1, L1, 1, th + d.L1b, th + d.L1g, nullptr, nullptr);
        __syncthreads();
    }
    if (d.blocked) rlc_hidden_forward_row(th + d.Wa2, 1, th + d.ba2, L.h1, L1, L2, L.ha);
    else blk_dense(L.h1, L1, L1, nullptr, 0, th + d.Wa2, th + d.ba2, L2, L.ha, L2, 1, act);
    for (int c = 0; c < A; c++) blk_dense(L.h1, L1, L1, nullptr, 0, th + d.Wd[c], th + d.bd[c], 1, L.dpre + c, A, 1, 0);
    {
        int off = 0;
        for (int c = 0; c < A - 1; c++) {
            blk_dense(L.h1, L1, L1, nullptr, 0, th + d.Wn[c], th + d.bn[c], A - 1 - c, L.npre + off, NN, 1, 0);
            off += A - 1 - c;
        }
    }
    __syncthreads();
    if (d.norm) {
        blk_layernorm_relu(L.ha, L2, 1, th + d.La2b, th + d.La2g, nullptr, nullptr);
        __syncthreads();
    }
    blk_dense(L.ha, L2, L2, nullptr, 0, th + d.Wa3, th + d.ba3, A, L.z, A, 1, 0);
    __syncthreads();
    if ((int)threadIdx.x < A) L.out[threadIdx.x] = tanhf(L.z[threadIdx.x]) * amax[threadIdx.x];
    __syncthreads();
}

// element (row i, col c) of L, i >= c, from the head pre-activations
__device__ inline float naf_l_entry(const RlcNafDims& d, const NafPolicyLds& L, int i, int c) {
    if (i == c) return expf(fminf(fmaxf(L.dpre[c], -5.0f), 5.0f));
    int off = 0;
    for (int k = 0; k < c; k++) off += d.A - 1 - k;
    return L.npre[off + (i - c - 1)];
}

// thread 0: L.out <- clip(mu + sqrt(noise_scale) * L^-T z, a_min, a_max) (naf_network.py:176), z from the agent's Philox stream
__device__ inline void naf_explore(const RlcNafDims& d, const NafPolicyLds& L, const float* amin, const float* amax,
                                   float noise_scale, unsigned long long seed, unsigned long long ctr) {
    const int A = d.A;
    float y[RLC_NAF_MAX_A];
    const float sc = sqrtf(noise_scale);
    for (int j = 0; j < A; j++) {
        const Philox4 p = philox4x32_10(seed ^ RLC_KEY_NAF_EPS, ctr, (unsigned long long)(j >> 1));
        float n0, n1;
        philox_normal2(p, n0, n1);
        y[j] = sc * ((j & 1) ? n1 : n0);
    }
    for (int i = A - 1; i >= 0; i--) {           // solve L^T y' = y:  (L^T)[i][j] = L[j][i], j >= i
        float s = y[i];
        for (int j = i + 1; j < A; j++) s -= naf_l_entry(d, L, j, i) * y[j];
        y[i] = s / naf_l_entry(d, L, i, i);
    }
    for (int j = 0; j < A; j++) L.out[j] = fminf(fmaxf(L.out[j] + y[j], amin[j]), amax[j]);
}

#endif  // __HIPCC__
