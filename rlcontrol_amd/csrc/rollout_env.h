// rollout_env.h -- the simulated environment and the algorithm-independent half of one training step of the
// on-device experiment loop: env.step, BaseAgent.update's insert rule, the learn
// gate and the episode bookkeeping of experiment.py:113-135 / agents/base_agent.py:54-70.  Shared by the DDPG and
// SAC train steps (ddpg_rollout_device.h, sac_rollout_device.h) and by the evaluation kernels.
//
// Environment: Pendulum-v0 restated from the public gym 0.18.0 definition (third-party; see
// rlcontrol_amd/environments/pendulum.py), simulated in float64 like gym does.  Reset draws come from a
// Philox stream per agent, in the order a sequential run would draw them.
#pragma once
#include "rlc_common.h"

#ifdef __HIPCC__

#define RLC_KEY_ENV_TRAIN 0x7261696Eull
#define RLC_KEY_ENV_TEST 0x74657374ull
#define RLC_PI 3.14159265358979323846

__device__ inline double rlc_u01(unsigned int hi, unsigned int lo) {
    return (double)((((unsigned long long)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

// ---- Pendulum-v0 -------------------------------------------------------------------------------
__device__ inline void pendulum_reset(double* sim, unsigned long long key, unsigned long long ctr) {
    const Philox4 p = philox4x32_10(key, ctr, 0);
    sim[0] = -RLC_PI + 2.0 * RLC_PI * rlc_u01(p.x, p.y);   // np_random.uniform(-[pi,1], [pi,1])
    sim[1] = -1.0 + 2.0 * rlc_u01(p.z, p.w);
}
__device__ inline void pendulum_obs(const double* sim, double* obs) {
    obs[0] = cos(sim[0]); obs[1] = sin(sim[0]); obs[2] = sim[1];
}
// returns the reward; advances sim
__device__ inline double pendulum_step(double* sim, const float* action) {
    const double th = sim[0], thdot = sim[1];
    const double u = fmin(fmax((double)action[0], -2.0), 2.0);
    double wrapped = fmod(th + RLC_PI, 2.0 * RLC_PI);
    if (wrapped < 0.0) wrapped += 2.0 * RLC_PI;          // Python's % is non-negative
    wrapped -= RLC_PI;
    const double cost = wrapped * wrapped + 0.1 * thdot * thdot + 0.001 * (u * u);
    double nthdot = thdot + (-3.0 * 10.0 / (2.0 * 1.0) * sin(th + RLC_PI) + 3.0 / (1.0 * 1.0 * 1.0) * u) * 0.05;
    const double nth = th + nthdot * 0.05;
    nthdot = fmin(fmax(nthdot, -8.0), 8.0);
    sim[0] = nth; sim[1] = nthdot;
    return -cost;
}

__device__ inline void env_reset(int env_id, double* sim, double* obs, unsigned long long key, unsigned long long ctr) {
    (void)env_id;
    pendulum_reset(sim, key, ctr);
    pendulum_obs(sim, obs);
}
// one simulator step: reward out, obs <- next observation, returns 1 when the environment reports done
__device__ inline int env_step(int env_id, double* sim, const float* action, double* obs, double* reward,
                               int ep_step, int limit) {
    (void)env_id;
    *reward = pendulum_step(sim, action);
    pendulum_obs(sim, obs);
    return ep_step >= limit;                              // gym.wrappers.TimeLimit
}

// Episode start of the training environment (run_episode_train: env.reset(), experiment.py:103-107); thread 0 only.
__device__ inline void rlc_env_begin_episode(const RlcReplayDev& rep, const RlcEnvDev& env, int agent) {
    double* sim = env.sim + (size_t)agent * RLC_ENV_STATE;
    double* obs = env.obs + (size_t)agent * rep.S;
    const unsigned long long c = env.reset_ctr[agent];
    env_reset(env.env_id, sim, obs, rep.seed[agent] ^ RLC_KEY_ENV_TRAIN, c);
    env.reset_ctr[agent] = c + 1;
    env.ep_step[agent] = 0;
    env.ep_ret[agent] = 0.0;
}

// env.step(action) -> BaseAgent.update (store unless truncated, gamma_i = 0 at terminals) -> bookkeeping; thread 0
// only.  Returns 1 when learn() would run (size > max(warmup, batch), agents/base_agent.py:65-70).
__device__ inline int rlc_env_advance_store(const RlcReplayDev& rep, const RlcEnvDev& env, int agent, const float* act) {
    const int S = rep.S, A = rep.A;
    double* sim = env.sim + (size_t)agent * RLC_ENV_STATE;
    double* obs = env.obs + (size_t)agent * S;
    const int step = env.ep_step[agent] + 1;
    double s_prev[8], reward;
    for (int i = 0; i < S && i < 8; i++) s_prev[i] = obs[i];
    const int done = env_step(env.env_id, sim, act, obs, &reward, step, env.episode_limit);
    const double ret = env.ep_ret[agent] + reward;
    const int truncated = done && step == env.episode_limit;
    RlcRingMeta m = rep.ring[agent];
    if (!truncated) {
        const long long cap = rep.cap;
        long long slot = m.start + m.size;
        if (slot >= cap) slot -= cap;
        if (m.size == cap) m.start = (m.start + 1 == cap) ? 0 : m.start + 1;
        else m.size += 1;
        const size_t at = (size_t)agent * cap + slot;
        for (int i = 0; i < S; i++) {
            rep.rs[at * S + i] = (float)s_prev[i];
            rep.rs2[at * S + i] = (float)obs[i];
        }
        for (int j = 0; j < A; j++) rep.ra[at * A + j] = act[j];
        rep.rr[at] = reward;
        rep.rg[at] = done ? 0.0 : env.gamma;
        rep.ring[agent] = m;
    }
    env.total_steps[agent] += 1;
    env.ep_step[agent] = step;
    env.ep_ret[agent] = ret;
    if (done || step == env.episode_limit) {
        const int e = env.n_train_ep[agent];
        if (e < env.max_episodes) {
            env.train_ret[(size_t)agent * env.max_episodes + e] = ret;
            env.train_len[(size_t)agent * env.max_episodes + e] = step;
            env.train_cum[(size_t)agent * env.max_episodes + e] = env.total_steps[agent];
        }
        env.n_train_ep[agent] = e + 1;
        env.need_reset[agent] = 1;
    } else {
        env.need_reset[agent] = 0;
    }
    __threadfence();                                   // the replay slot is read back by this workgroup's gather
    return m.size > env.learn_threshold ? 1 : 0;
}

#endif  // __HIPCC__
