// rlc_api_rollout.hip -- C ABI of the on-device experiment loop (include/rlcontrol_hip.h, "rollout" block).
// Host side = the schedule of experiment.py:52-161 (evaluation 0, then train step / update / periodic
// evaluation); all per-agent state lives on the GPU (rollout_kernels.hip).
#include "rlc_handle.h"
#include "sac_rollout_device.h"
#include "naf_rollout_device.h"

#define RLC_NEED_DDPG(h) RLC_REQUIRE((h) && (h)->algo == RLC_ALGO_DDPG, "handle is not a DDPG population")
#define RLC_NEED_ENV(h) RLC_REQUIRE((h) && (h)->has_env, "no rollout configured on this handle (rlc_ddpg_rollout_create)")

// n training steps of every agent in ONE launch of the fused update kernel
static int launch_steps(rlc_handle* h, int n, int q8_first) {
    const RlcDev& dv = h->dv;
    const bool mfma = h->variant == 2 || (h->variant == 0 && rlc_mfma_supported(dv.d));
    if (mfma)
        return rlc_launch_ddpg_update_mfma(dv, 0, dv.n_agents, n, RLC_SRC_REPLAY_DEVICE_SAMPLER, nullptr, 0, h->st,
                                           h->rollout_dev, q8_first);
    return rlc_launch_ddpg_update_generic(dv, 0, dv.n_agents, n, RLC_SRC_REPLAY_DEVICE_SAMPLER, nullptr, 0, h->st,
                                          h->rollout_dev, q8_first);
}

extern "C" {

// environment state, bookkeeping and logs of a population (any algorithm)
static int rollout_alloc(rlc_handle* h, const rlc_rollout_config* cfg) {
    RLC_REQUIRE(!h->has_env, "rollout already configured on this handle");
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(cfg->env_id == RLC_ENV_PENDULUM_V0, "unknown env_id %d (built in: RLC_ENV_PENDULUM_V0)", cfg->env_id);
    RLC_REQUIRE(h->rep.S == 3 && h->rep.A == 1, "Pendulum-v0 needs state_dim 3 / action_dim 1 (handle has %d / %d)",
                h->rep.S, h->rep.A);
    RLC_REQUIRE(cfg->episode_steps_limit >= 1 && cfg->total_steps_limit >= 0 && cfg->eval_interval >= 1 &&
                cfg->eval_episodes >= 0 && cfg->warmup_steps >= 0 && cfg->max_train_episodes >= 1,
                "bad rollout configuration");
    RlcEnvDev& e = h->env;
    e.env_id = RLC_ENV_PENDULUM;
    e.episode_limit = cfg->episode_steps_limit;
    e.learn_threshold = cfg->warmup_steps > h->B ? cfg->warmup_steps : h->B;
    e.eval_episodes = cfg->eval_episodes;
    e.max_episodes = cfg->max_train_episodes;
    e.max_evals = (int)(cfg->total_steps_limit / cfg->eval_interval) + 1;
    e.gamma = cfg->gamma;
    const size_t NA = h->rep.n_agents, S = h->rep.S;
    const size_t nev = NA * (size_t)e.max_evals * (e.eval_episodes ? e.eval_episodes : 1);
    if (rlc_h_malloc(h, &e.sim, NA * RLC_ENV_STATE) || rlc_h_malloc(h, &e.obs, NA * S) ||
        rlc_h_malloc(h, &e.ep_step, NA) || rlc_h_malloc(h, &e.ep_ret, NA) || rlc_h_malloc(h, &e.need_reset, NA) ||
        rlc_h_malloc(h, &e.total_steps, NA) || rlc_h_malloc(h, &e.reset_ctr, NA) ||
        rlc_h_malloc(h, &e.train_ret, NA * e.max_episodes) || rlc_h_malloc(h, &e.train_len, NA * e.max_episodes) ||
        rlc_h_malloc(h, &e.train_cum, NA * e.max_episodes) || rlc_h_malloc(h, &e.n_train_ep, NA) ||
        rlc_h_malloc(h, &e.eval_ret, nev) || rlc_h_malloc(h, &e.eval_len, nev))
        return 1;
    std::vector<int> ones(NA, 1);
    RLC_HIP(hipMemcpyAsync(e.need_reset, ones.data(), NA * sizeof(int), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    h->ro_total_limit = cfg->total_steps_limit;
    h->ro_eval_interval = cfg->eval_interval;
    h->ro_steps = 0;
    h->ro_evals = 0;
    h->ro_pending_q8 = 0;
    return 0;
}

int rlc_ddpg_rollout_create(rlc_handle* h, const rlc_rollout_config* cfg) {
    RLC_REQUIRE(h && cfg, "null argument");
    RLC_NEED_DDPG(h);
    if (rollout_alloc(h, cfg)) return 1;
    // device-resident argument block of the fused launches
    if (rlc_h_malloc(h, &h->rollout_dev, 1)) return 1;
    RlcRollout ro;
    ro.dv = h->dv;
    ro.env = h->env;
    RLC_HIP(hipMemcpyAsync(h->rollout_dev, &ro, sizeof(ro), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    h->has_env = true;
    return 0;
}

// SoftActorCritic and the KL agents share the device view (RlcSacDev) and the train step (sac_rollout_device.h)
static int sacfam_rollout_create(rlc_handle* h, const rlc_rollout_config* cfg) {
    if (rollout_alloc(h, cfg)) return 1;
    if (rlc_h_malloc(h, &h->sac_rollout_dev, 1)) return 1;
    RlcSacRollout ro;
    ro.dv = h->sac;
    ro.env = h->env;
    RLC_HIP(hipMemcpyAsync(h->sac_rollout_dev, &ro, sizeof(ro), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    h->has_env = true;
    return 0;
}

int rlc_sac_rollout_create(rlc_handle* h, const rlc_rollout_config* cfg) {
    RLC_REQUIRE(h && cfg, "null argument");
    RLC_REQUIRE(h->algo == RLC_ALGO_SAC, "handle is not a SoftActorCritic population");
    return sacfam_rollout_create(h, cfg);
}

int rlc_kl_rollout_create(rlc_handle* h, const rlc_rollout_config* cfg) {
    RLC_REQUIRE(h && cfg, "null argument");
    RLC_REQUIRE(h->algo == RLC_ALGO_KL, "handle is not a ReverseKL / ForwardKL population");
    return sacfam_rollout_create(h, cfg);
}

int rlc_naf_rollout_create(rlc_handle* h, const rlc_rollout_config* cfg, const float* noise_scale) {
    RLC_REQUIRE(h && cfg && noise_scale, "null argument");
    RLC_REQUIRE(h->algo == RLC_ALGO_NAF, "handle is not a NAF population");
    if (rollout_alloc(h, cfg)) return 1;
    const size_t NA = h->naf.n_agents;
    float* ns_dev;
    unsigned long long* ctr_dev;
    if (rlc_h_malloc(h, &ns_dev, NA) || rlc_h_malloc(h, &ctr_dev, NA) || rlc_h_malloc(h, &h->naf_rollout_dev, 1)) return 1;
    for (size_t i = 0; i < NA; i++) RLC_REQUIRE(noise_scale[i] >= 0.0f, "negative noise_scale");
    RLC_HIP(hipMemcpyAsync(ns_dev, noise_scale, NA * sizeof(float), hipMemcpyHostToDevice, h->st));
    RlcNafRollout ro;
    ro.dv = h->naf;
    ro.env = h->env;
    ro.noise_scale = ns_dev;
    ro.noise_ctr = ctr_dev;
    RLC_HIP(hipMemcpyAsync(h->naf_rollout_dev, &ro, sizeof(ro), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    h->has_env = true;
    return 0;
}

// the schedule of experiment.py:52-161 for any of the algorithms
static int rollout_run(rlc_handle* h, int64_t n_steps, int64_t* out_total_steps) {
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n_steps >= 0, "negative n_steps");
    const bool sac = h->algo == RLC_ALGO_SAC || h->algo == RLC_ALGO_KL, naf = h->algo == RLC_ALGO_NAF;
    auto eval_now = [&]() -> int {
        if (h->env.eval_episodes > 0) {
            const int rc = sac ? rlc_launch_sac_eval(h->sac, h->env, (int)h->ro_evals, h->st)
                           : naf ? rlc_launch_naf_eval(h->naf, h->env, (int)h->ro_evals, h->st)
                                 : rlc_launch_ddpg_eval(h->dv, h->env, (int)h->ro_evals, h->st);
            if (rc) return 1;
        }
        h->ro_evals += 1;
        return 0;
    };
    if (h->ro_steps == 0 && h->ro_evals == 0)          // experiment.py:57-58: evaluate before any training
        if (eval_now()) return 1;
    // one launch per stretch between evaluations (experiment.py:131-133); kMaxPerLaunch bounds a launch's run time
    const long long kMaxPerLaunch = (sac || naf) ? 500 : 2000;
    long long todo = n_steps;
    if (todo > h->ro_total_limit - h->ro_steps) todo = h->ro_total_limit - h->ro_steps;
    while (todo > 0) {
        long long n = h->ro_eval_interval - h->ro_steps % h->ro_eval_interval;      // steps to the next evaluation
        if (n > todo) n = todo;
        if (n > kMaxPerLaunch) n = kMaxPerLaunch;
        if (sac || naf) {
            const int taps = h->grad_taps;
            h->grad_taps = 0;                      // the fused loop writes no gradient taps
            const int rc = sac ? rlc_h_sac_launch_update(h, 0, h->sac.n_agents, (int)n, RLC_SRC_REPLAY_DEVICE_SAMPLER, nullptr,
                                                         nullptr, h->sac_rollout_dev)
                               : rlc_h_naf_launch_update(h, 0, h->naf.n_agents, (int)n, RLC_SRC_REPLAY_DEVICE_SAMPLER, nullptr,
                                                         h->naf_rollout_dev);
            h->grad_taps = taps;
            if (rc) return 1;
        } else if (launch_steps(h, (int)n, h->ro_pending_q8)) {
            return 1;
        }
        h->ro_pending_q8 = 0;
        h->ro_steps += n;
        todo -= n;
        if (h->ro_steps % h->ro_eval_interval == 0) {
            if (eval_now()) return 1;
            h->ro_pending_q8 = h->env.eval_episodes > 0;   // agent.reset() happens inside run_episode_eval only
        }
    }
    // the replay grew on the device: refresh the host mirror of the ring metadata
    RLC_HIP(hipMemcpyAsync(h->ring.data(), h->rep.ring, sizeof(RlcRingMeta) * h->ring.size(), hipMemcpyDeviceToHost,
                           h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    if (out_total_steps) *out_total_steps = h->ro_steps;
    return 0;
}

int rlc_ddpg_rollout_run(rlc_handle* h, int64_t n_steps, int64_t* out_total_steps) {
    RLC_NEED_ENV(h);
    RLC_NEED_DDPG(h);
    return rollout_run(h, n_steps, out_total_steps);
}

int rlc_sac_rollout_run(rlc_handle* h, int64_t n_steps, int64_t* out_total_steps) {
    RLC_NEED_ENV(h);
    RLC_REQUIRE(h->algo == RLC_ALGO_SAC, "handle is not a SoftActorCritic population");
    return rollout_run(h, n_steps, out_total_steps);
}

int rlc_kl_rollout_run(rlc_handle* h, int64_t n_steps, int64_t* out_total_steps) {
    RLC_NEED_ENV(h);
    RLC_REQUIRE(h->algo == RLC_ALGO_KL, "handle is not a ReverseKL / ForwardKL population");
    return rollout_run(h, n_steps, out_total_steps);
}

int rlc_naf_rollout_run(rlc_handle* h, int64_t n_steps, int64_t* out_total_steps) {
    RLC_NEED_ENV(h);
    RLC_REQUIRE(h->algo == RLC_ALGO_NAF, "handle is not a NAF population");
    return rollout_run(h, n_steps, out_total_steps);
}

int rlc_rollout_counts(rlc_handle* h, int32_t agent, int64_t* n_train_episodes, int64_t* n_evals,
                       int64_t* total_steps) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ENV(h);
    int ne = 0;
    RLC_HIP(hipMemcpyAsync(&ne, h->env.n_train_ep + agent, sizeof(int), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    if (n_train_episodes) *n_train_episodes = ne;
    if (n_evals) *n_evals = h->ro_evals;
    if (total_steps) *total_steps = h->ro_steps;
    return 0;
}

int rlc_rollout_train_log(rlc_handle* h, int32_t agent, int64_t n, double* returns, int32_t* lengths,
                          int64_t* cum_steps) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ENV(h);
    RLC_REQUIRE(n >= 0 && n <= h->env.max_episodes, "n=%lld outside the log capacity %d", (long long)n,
                h->env.max_episodes);
    if (n == 0) return 0;
    RLC_REQUIRE(returns && lengths && cum_steps, "null output array");
    const size_t off = (size_t)agent * h->env.max_episodes;
    RLC_HIP(hipMemcpyAsync(returns, h->env.train_ret + off, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipMemcpyAsync(lengths, h->env.train_len + off, sizeof(int) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipMemcpyAsync(cum_steps, h->env.train_cum + off, sizeof(long long) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_rollout_eval_log(rlc_handle* h, int32_t agent, int64_t n, double* returns, int32_t* lengths) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ENV(h);
    RLC_REQUIRE(n >= 0 && n <= h->ro_evals && n <= h->env.max_evals, "n=%lld exceeds the %lld evaluations run",
                (long long)n, h->ro_evals);
    const size_t E = h->env.eval_episodes;
    if (n == 0 || E == 0) return 0;
    RLC_REQUIRE(returns && lengths, "null output array");
    const size_t off = (size_t)agent * h->env.max_evals * E;
    RLC_HIP(hipMemcpyAsync(returns, h->env.eval_ret + off, sizeof(double) * n * E, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipMemcpyAsync(lengths, h->env.eval_len + off, sizeof(int) * n * E, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_rollout_observation(rlc_handle* h, int32_t agent, double* obs, int32_t* episode_step) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ENV(h);
    RLC_REQUIRE(obs && episode_step, "null output");
    const size_t S = h->rep.S;
    RLC_HIP(hipMemcpyAsync(obs, h->env.obs + (size_t)agent * S, sizeof(double) * S, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipMemcpyAsync(episode_step, h->env.ep_step + agent, sizeof(int), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

}  // extern "C"
