// ddpg_rollout_device.h -- device side of the on-device experiment loop:
// the simulated environment and ONE training step of one agent, called at the top of every iteration of the
// fused update kernels (ddpg_mfma_kernel.h, ddpg_generic.hip) when a rollout is attached to the launch.
//
// Per training step the reference does (experiment.py:113-135, agents/base_agent.py:34-70):
//   env.step(action) -> agent.update(obs, obs_n, r, action, done, truncated)   [store unless truncated, learn()]
//   -> action = agent.step(obs_n) if not done -> eval() every eval_interval steps.
// Here the act that FOLLOWS update t opens iteration t+1 (same weights, same observation, same OU draw
// order): step(t+1) = [reset if the episode ended] act -> env.step -> replay put -> learn gate -> update.
// An evaluation between update t and step t+1 resets the OU state AFTER the pending action was drawn
// (quirk Q8): the launcher starts a new launch after every evaluation and flags its first step.
//
// Environment: Pendulum-v0 restated from the public gym 0.18.0 definition (third-party; see
// rlcontrol_amd/environments/pendulum.py), simulated in float64 like gym does.  Reset draws come from a
// Philox stream per agent, in the order a sequential run would draw them.
#pragma once
#include "ddpg_policy.h"
#include "rollout_env.h"

#ifdef __HIPCC__

// One training step of `agent`; every thread of the workgroup calls it.  `scratch` = LDS, at least
// ddpg_policy_lds_floats(d) + 4 floats.  Returns 1 when learn() would run (size > max(warmup, batch)).
// `ro` points to device memory (a copy of the handle's views): passing it by pointer keeps the kernel's own
// by-value argument out of scratch memory.
__device__ __noinline__ int rlc_train_step_device(const RlcRollout* ro, int agent, float* scratch,
                                                  int reset_noise_after_act) {
    const RlcDev& dv = ro->dv;
    const RlcEnvDev& env = ro->env;
    const RlcDims& d = dv.d;
    const int S = d.S, A = d.A;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const DdpgPolicyLds L = ddpg_policy_carve(d, scratch);
    int* learn_flag = (int*)(L.act + ((A + 3) & ~3));
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    double* obs = env.obs + (size_t)agent * S;

    __syncthreads();                                       // scratch may alias the caller's buffers
    const int fresh = env.need_reset[agent];
    if (fresh) {
        // run_episode_train: env.reset(); agent.reset()  (experiment.py:103-107)
        if (tid == 0) rlc_env_begin_episode(dv.rep, env, agent);
        if (tid < A) dv.ou_state[agent * A + tid] = dv.ou_mu;
        __syncthreads();
    }
    for (int i = tid; i < S; i += nthr)
        L.x[i] = clip_state_val((float)obs[i], dv.clip_state, dv.smin[i], dv.smax[i]);
    ddpg_greedy_forward(d, th, L, dv.amax);
    if (tid < A) {
        L.act[tid] = ddpg_ou_explore(dv, agent, tid, L.act[tid], dv.noise_ctr[agent]);
        if (reset_noise_after_act && !fresh) dv.ou_state[agent * A + tid] = dv.ou_mu;     // Q8
    }
    __syncthreads();
    if (tid == 0) {
        dv.noise_ctr[agent] += 1;
        *learn_flag = rlc_env_advance_store(dv.rep, env, agent, L.act);
    }
    __syncthreads();
    const int learn = *learn_flag;
    __syncthreads();
    return learn;
}

#endif  // __HIPCC__
