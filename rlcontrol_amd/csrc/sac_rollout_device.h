// sac_rollout_device.h -- one training step of a SoftActorCritic agent inside the fused update kernel
// (sac_generic.hip), the SAC counterpart of ddpg_rollout_device.h: [episode reset] -> reparameterised action from
// the current policy (agents/SoftActorCritic.py:55-72: exploration_policy 'none', the sample IS the exploration)
// -> env.step -> BaseAgent.update's insert rule -> learn gate.  There is no noise state to reset (quirk Q8 does
// not apply).  The N(0,1) draws of acting and of the minibatch share the agent's Philox stream counter, as in the
// host-driven device path.  The ReverseKL / ForwardKL populations (kl_generic.hip, kl_mfma_kernel.h) take the same step:
// their handle is an RlcSacDev too (no state clip, clamped log_std).
#pragma once
#include "sac_policy.h"
#include "rollout_env.h"

#ifdef __HIPCC__

struct RlcSacRollout {
    RlcSacDev dv;
    RlcEnvDev env;
};

__device__ __noinline__ int rlc_sac_train_step_device(const RlcSacRollout* ro, int agent, float* scratch) {
    const RlcSacDev& dv = ro->dv;
    const RlcEnvDev& env = ro->env;
    const RlcSacDims& d = dv.d;
    const int S = d.S, A = d.A;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const SacPolicyLds L = sac_policy_carve(d, scratch);
    int* learn_flag = (int*)(L.eps + ((A + 3) & ~3));
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    double* obs = env.obs + (size_t)agent * S;

    __syncthreads();
    if (env.need_reset[agent]) {
        if (tid == 0) rlc_env_begin_episode(dv.rep, env, agent);
        __syncthreads();
    }
    for (int i = tid; i < S; i += nthr) L.x[i] = rlc_clip_scalar((float)obs[i], dv.clip_state, dv.smin0, dv.smax0);
    if (tid < A) L.eps[tid] = sac_act_eps(dv.rep.seed[agent], dv.noise_ctr[agent], tid);
    sac_policy_forward(d, th, L, dv.amax0, 1, dv.kl_kind != 0);       // the KL agents clamp log_std, SAC squashes it
    if (tid == 0) {
        dv.noise_ctr[agent] += 1;
        *learn_flag = rlc_env_advance_store(dv.rep, env, agent, L.out);
    }
    __syncthreads();
    const int learn = *learn_flag;
    __syncthreads();
    return learn;
}

#endif  // __HIPCC__
