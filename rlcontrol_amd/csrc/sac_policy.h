// sac_policy.h -- B=1 forward of the SAC-v1 policy (mean action or reparameterised sample), shared by the acting
// kernel (sac_generic.hip), the on-device train step (sac_rollout_device.h) and the on-device evaluation.
// Reference: agents/SoftActorCritic.py:55-112 (sample_action on B=1 / predict_action), sac_network.py:234-301,327-343.
#pragma once
#include "sac_common.h"

#ifdef __HIPCC__

#define RLC_KEY_SAC_EPS 0x9E3779B97F4A7C15ull

__device__ __forceinline__ float rlc_clip_scalar(float v, int on, float lo, float hi) {
    return on ? fminf(fmaxf(v, lo), hi) : v;
}

// LDS floats: x | h1 | h2 | out | eps (+ 4: a flag word of the train step) | red (layer-norm row reductions)
__host__ __device__ inline size_t sac_policy_lds_floats(const RlcSacDims& d) {
    return (size_t)((d.S + 3) & ~3) + ((d.L1A + 3) & ~3) + ((d.L2A + 3) & ~3) + 2 * ((d.A + 3) & ~3) + 4 + 40;
}
struct SacPolicyLds { float *x, *h1, *h2, *out, *eps, *red; };
__device__ inline SacPolicyLds sac_policy_carve(const RlcSacDims& d, float* base) {
    SacPolicyLds L;
    L.x = base;
    L.h1 = L.x + ((d.S + 3) & ~3);
    L.h2 = L.h1 + ((d.L1A + 3) & ~3);
    L.out = L.h2 + ((d.L2A + 3) & ~3);
    L.eps = L.out + ((d.A + 3) & ~3);
    L.red = L.eps + ((d.A + 3) & ~3) + 4;
    return L;
}

// the N(0,1) draw of action component j for the acting call number `nctr` of an agent (device stream)
__device__ inline float sac_act_eps(unsigned long long seed, unsigned long long nctr, int j) {
    const Philox4 p = philox4x32_10(seed ^ RLC_KEY_SAC_EPS, nctr, 0x4000000000000000ull + (unsigned long long)(j >> 1));
    float n0, n1;
    philox_normal2(p, n0, n1);
    return (j & 1) ? n1 : n0;
}

// L.x = clipped state (and L.eps[j] when sample != 0); on return L.out[j] = tanh(mu [+ eps*std]) * a_max.
// Every thread of the workgroup (a multiple of 64 threads) calls it; trailing barrier included.
// clamp_ls: 0 = SAC's log_std = -20 + 11*(tanh(.)+1); 1 = the KL agents' clamp(., -20, 2) (reversekl_network.py:318)
__device__ inline void sac_policy_forward(const RlcSacDims& d, const float* th, const SacPolicyLds& L, float amax0,
                                          int sample, int clamp_ls = 0) {
    const int S = d.S, A = d.A, L1A = d.L1A, L2A = d.L2A;
    const int tid = threadIdx.x, nthr = blockDim.x;
    __syncthreads();
    for (int k = tid; k < L1A; k += nthr) {
        float acc = 0.0f;
        for (int i = 0; i < S; i++) acc += L.x[i] * th[d.pW1 + i * L1A + k];
        acc += th[d.pb1 + k];
        L.h1[k] = d.norm ? acc : fmaxf(acc, 0.0f);
    }
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(L.h1, L1A, th + d.pL1b, th + d.pL1g, L.red);
    if (d.blocked) {
        rlc_hidden_forward_row(th + d.pW2, 1, th + d.pb2, L.h1, L1A, L2A, L.h2, !d.norm);
    } else {
        for (int n = tid; n < L2A; n += nthr) {
            float acc = 0.0f;
            for (int k = 0; k < L1A; k++) acc += L.h1[k] * th[d.pW2 + (size_t)k * L2A + n];
            acc += th[d.pb2 + n];
            L.h2[n] = d.norm ? acc : fmaxf(acc, 0.0f);
        }
    }
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(L.h2, L2A, th + d.pL2b, th + d.pL2g, L.red);
    const int wave = tid / 64, lane = tid % 64;
    for (int j = wave; j < A; j += nthr / 64) {
        float am = 0.0f, as = 0.0f;
        for (int n = lane; n < L2A; n += 64) {
            am += L.h2[n] * th[d.pWm + n * A + j];
            as += L.h2[n] * th[d.pWs + n * A + j];
        }
        for (int off = 32; off > 0; off >>= 1) { am += __shfl_down(am, off, 64); as += __shfl_down(as, off, 64); }
        if (lane == 0) {
            float u = am + th[d.pbm + j];
            if (sample) {
                const float raw = as + th[d.pbs + j];
                const float log_std = clamp_ls ? fminf(fmaxf(raw, -20.0f), 2.0f)
                                               : -20.0f + 0.5f * (2.0f - (-20.0f)) * (tanhf(raw) + 1.0f);
                // the KL agents above one action dimension sample MultivariateNormal(mean, diag_embed(std)): covariance
                // diag(std), i.e. a standard deviation of sqrt(std) (reversekl_network.py:383-389)
                u += L.eps[j] * ((clamp_ls && A > 1) ? sqrtf(expf(log_std)) : expf(log_std));
            }
            L.out[j] = tanhf(u) * amax0;
        }
    }
    __syncthreads();
}

#endif  // __HIPCC__
