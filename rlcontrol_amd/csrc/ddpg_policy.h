// ddpg_policy.h -- B=1 greedy forward of the hydra actor + the device OU process, shared by the acting
// kernel (ddpg_generic.hip), the on-device train step and the on-device evaluation (rollout_kernels.hip).
// Reference: agents/DDPG.py:34-72 (predict_action on one state), utils/exploration_policy.py:18-24.
#pragma once
#include "rlc_common.h"

#ifdef __HIPCC__

#define RLC_POLICY_THREADS 256

// LDS floats needed by ddpg_greedy_forward: x | h1 | h2 | act | red (layer-norm row reductions)
__host__ __device__ inline size_t ddpg_policy_lds_floats(const RlcDims& d) {
    return (size_t)((d.S + 3) & ~3) + ((d.H1 + 3) & ~3) + ((d.HA + 3) & ~3) + ((d.A + 3) & ~3) + 40;
}

struct DdpgPolicyLds {
    float *x, *h1, *h2, *act, *red;
};

__device__ inline DdpgPolicyLds ddpg_policy_carve(const RlcDims& d, float* base) {
    DdpgPolicyLds L;
    L.x = base;
    L.h1 = L.x + ((d.S + 3) & ~3);
    L.h2 = L.h1 + ((d.H1 + 3) & ~3);
    L.act = L.h2 + ((d.HA + 3) & ~3);
    L.red = L.act + ((d.A + 3) & ~3);
    return L;
}

// L.x holds the clipped state; on return (after the trailing barrier) L.act[j] = tanh(.)*a_max[j].
// Every thread of the workgroup (any size that is a multiple of 64) must call it.  One thread per output
// unit, k ascending in one accumulator (the summation order every caller shares); the weight column is
// fetched KC rows at a time so that KC loads are in flight per thread instead of one dependent chain.
__device__ inline void ddpg_greedy_forward(const RlcDims& d, const float* th, const DdpgPolicyLds& L,
                                           const float* amax) {
    const int S = d.S, A = d.A, H1 = d.H1, HA = d.HA;
    const int tid = threadIdx.x, nthr = blockDim.x;
    __syncthreads();
    for (int k = tid; k < H1; k += nthr) {
        float acc = 0.0f;
        for (int i = 0; i < S; i++) acc += L.x[i] * th[d.oW1 + i * H1 + k];
        acc += th[d.ob1 + k];
        L.h1[k] = d.norm ? acc : fmaxf(acc, 0.0f);
    }
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(L.h1, H1, th + d.oL1b, th + d.oL1g, L.red);
    rlc_hidden_forward_row(th + d.oWa2, d.blocked, th + d.oba2, L.h1, H1, HA, L.h2, !d.norm);
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(L.h2, HA, th + d.oL2b, th + d.oL2g, L.red);
    // one wave per output action: 64-lane shuffle reduction over HA
    const int wave = tid / RLC_WAVE, lane = tid % RLC_WAVE;
    for (int j = wave; j < A; j += nthr / RLC_WAVE) {
        float acc = 0.0f;
        for (int n = lane; n < HA; n += RLC_WAVE) acc += L.h2[n] * th[d.oWa3 + n * A + j];
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, RLC_WAVE);
        if (lane == 0) L.act[j] = tanhf(acc + th[d.oba3 + j]) * amax[j];
    }
    __syncthreads();
}

#define RLC_KEY_OU 0x5DEECE66Dull

// OU step for action component j of one agent: n <- n + N(mu, sigma) - theta*n ; returns clip(a + n)
// (utils/exploration_policy.py:18-21).  `ctr` = OU draws of this agent so far (one per acting call).
__device__ inline float ddpg_ou_explore(const RlcDev& dv, int agent, int j, float greedy, unsigned long long ctr) {
    const Philox4 p = philox4x32_10(dv.rep.seed[agent] ^ RLC_KEY_OU, ctr, (unsigned long long)(j / 2));
    float n0, n1;
    philox_normal2(p, n0, n1);
    const float z = (j & 1) ? n1 : n0;
    float noise = dv.ou_state[agent * dv.d.A + j];
    noise += (dv.ou_mu + dv.ou_sigma * z) - noise * dv.ou_theta;
    dv.ou_state[agent * dv.d.A + j] = noise;
    return fminf(fmaxf(greedy + noise, dv.amin[j]), dv.amax[j]);
}

#endif  // __HIPCC__
