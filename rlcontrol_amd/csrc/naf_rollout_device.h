// naf_rollout_device.h -- one training step of a NAF agent inside the fused update kernel (naf_generic.hip):
// [episode reset] -> exploration sample around the greedy action (naf_policy.h) -> env.step -> BaseAgent.update's
// insert rule -> learn gate; the NAF counterpart of ddpg_rollout_device.h / sac_rollout_device.h.
#pragma once
#include "naf_policy.h"
#include "rollout_env.h"

#ifdef __HIPCC__

struct RlcNafRollout {
    RlcNafDev dv;
    RlcEnvDev env;
    const float* noise_scale;            // [n_agents] (jsonfiles/agent/naf.json sweeps it)
    unsigned long long* noise_ctr;       // [n_agents] exploration draws so far
};

__device__ __noinline__ int rlc_naf_train_step_device(const RlcNafRollout* ro, int agent, float* scratch) {
    const RlcNafDev& dv = ro->dv;
    const RlcEnvDev& env = ro->env;
    const RlcNafDims& d = dv.d;
    const int S = d.S;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const NafPolicyLds L = naf_policy_carve(d, scratch);
    int* learn_flag = (int*)(L.out + RLC_NAF_MAX_A);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    double* obs = env.obs + (size_t)agent * S;

    __syncthreads();
    if (env.need_reset[agent]) {
        if (tid == 0) rlc_env_begin_episode(dv.rep, env, agent);
        __syncthreads();
    }
    for (int i = tid; i < S; i += nthr) L.x[i] = clip_state_val((float)obs[i], dv.clip_state, dv.smin[i], dv.smax[i]);
    naf_policy_forward(d, th, L, dv.amax);
    if (tid == 0) {
        naf_explore(d, L, dv.amin, dv.amax, ro->noise_scale[agent], dv.rep.seed[agent], ro->noise_ctr[agent]);
        ro->noise_ctr[agent] += 1;
        *learn_flag = rlc_env_advance_store(dv.rep, env, agent, L.out);
    }
    __syncthreads();
    const int learn = *learn_flag;
    __syncthreads();
    return learn;
}

#endif  // __HIPCC__
