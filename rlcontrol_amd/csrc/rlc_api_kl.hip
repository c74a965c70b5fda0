// rlc_api_kl.hip -- C ABI of the ReverseKL / ForwardKL populations (declared in include/rlcontrol_hip.h).
// The two agents share SoftActorCritic's network family (a Gaussian policy, a Q and a V network with a V target), so
// the handle reuses the device view RlcSacDev (with the Q network's action rows at the INPUT layer: RlcSacDims::qcat)
// and the blob / act / update / tap bodies of rlc_api_sac.hip; what is particular to them -- the action-integral
// update kernel, torch's Adam bookkeeping, the quadrature nodes -- lives here and in kl_generic.hip.
#include <algorithm>
#include <string.h>

#include <vector>

#include "rlc_handle.h"

#define RLC_NEED_KL(h) RLC_REQUIRE((h) && (h)->algo == RLC_ALGO_KL, "handle is not a ReverseKL / ForwardKL population")

extern "C" {

int rlc_kl_create(const rlc_kl_config* cfg, rlc_handle** out) {
    RLC_REQUIRE(cfg && out, "null argument");
    RLC_REQUIRE(cfg->actor_l1_dim >= 1 && cfg->actor_l2_dim >= 1 && cfg->critic_l1_dim >= 1 && cfg->critic_l2_dim >= 1,
                "layer widths must be >= 1");
    RLC_REQUIRE(cfg->pi_lr && cfg->qf_vf_lr && cfg->entropy_scale, "null per-agent array");
    RLC_REQUIRE(cfg->kind == RLC_KL_REVERSE || cfg->kind == RLC_KL_FORWARD, "kind must be RLC_KL_REVERSE or RLC_KL_FORWARD");
    RLC_REQUIRE(cfg->optim_type >= RLC_KL_OPTIM_INTG && cfg->optim_type <= RLC_KL_OPTIM_HARD_LL, "unknown optim_type %d",
                cfg->optim_type);
    // forwardkl_network.py:153-158: 'll' raises NotImplementedError, the other names are never matched
    RLC_REQUIRE(cfg->kind == RLC_KL_REVERSE || cfg->optim_type == RLC_KL_OPTIM_INTG,
                "ForwardKL implements optim_type 'intg' only");
    RLC_REQUIRE(cfg->q_update_type == RLC_KL_Q_NON_SAC || cfg->q_update_type == RLC_KL_Q_SAC, "unknown q_update_type %d",
                cfg->q_update_type);
    RLC_REQUIRE(cfg->action_dim >= 1 && cfg->action_dim <= 6, "action_dim %d outside [1,6]", cfg->action_dim);
    const bool integral = cfg->optim_type == RLC_KL_OPTIM_INTG || cfg->optim_type == RLC_KL_OPTIM_HARD_INTG;
    RLC_REQUIRE(!integral || (cfg->n_nodes >= 1 && cfg->node_actions && cfg->node_weights),
                "the integral updates need n_nodes >= 1 quadrature nodes and weights");
    RLC_REQUIRE(cfg->n_nodes >= 0 && cfg->n_nodes <= 4096, "n_nodes %d outside [0,4096]", cfg->n_nodes);
    RLC_REQUIRE(cfg->action_max0 > 0.0f, "action_max0 must be positive");
    for (int i = 0; i < cfg->n_agents; i++)
        RLC_REQUIRE(cfg->kind == RLC_KL_REVERSE || cfg->entropy_scale[i] > 0.0f,
                    "agent %d: ForwardKL divides Q by entropy_scale, which must be positive", i);
    if (integral)   // atanh of the normalised node must be finite (the reference cuts the end points for this reason)
        for (int k = 0; k < cfg->n_nodes * cfg->action_dim; k++)
            RLC_REQUIRE(cfg->node_actions[k] > -cfg->action_max0 && cfg->node_actions[k] < cfg->action_max0,
                        "node %d component %d (%g) is not strictly inside (-action_max, action_max)", k / cfg->action_dim,
                        k % cfg->action_dim, (double)cfg->node_actions[k]);
    rlc_handle* h = new rlc_handle();
    int rc = rlc_h_init_common(h, RLC_ALGO_KL, cfg->device, cfg->n_agents, cfg->state_dim, cfg->action_dim,
                               cfg->batch_size, cfg->buffer_size, cfg->seed);
    if (rc) { rlc_h_destroy(h); return rc; }
    RlcSacDev& dv = h->sac;
    dv.d = rlc_sac_make_dims(cfg->state_dim, cfg->action_dim, cfg->actor_l1_dim, cfg->actor_l2_dim, cfg->critic_l1_dim,
                             cfg->critic_l2_dim, cfg->batch_size, 0, 1);
    // the tile-blocked weight layout goes with the MFMA kernel (the default whenever it supports the shape)
    {
        const bool integral_ = cfg->optim_type == RLC_KL_OPTIM_INTG || cfg->optim_type == RLC_KL_OPTIM_HARD_INTG;
        if (rlc_kl_mfma_supported(dv.d, integral_ ? cfg->n_nodes : 0))
            dv.d = rlc_sac_make_dims(cfg->state_dim, cfg->action_dim, cfg->actor_l1_dim, cfg->actor_l2_dim,
                                     cfg->critic_l1_dim, cfg->critic_l2_dim, cfg->batch_size, 1, 1);
    }
    dv.rep = h->rep;
    dv.n_agents = cfg->n_agents;
    dv.clip_state = 0;   // the networks never apply the input normaliser they are handed (reversekl_network.py:43)
    dv.tau = cfg->tau;
    dv.smin0 = dv.smax0 = 0.0f;
    dv.amax0 = cfg->action_max0;
    dv.kl_kind = cfg->kind; dv.kl_optim = cfg->optim_type; dv.kl_qupdate = cfg->q_update_type;
    dv.kl_nodes = integral ? cfg->n_nodes : 0;
    const size_t NA = cfg->n_agents, PP = dv.d.Ppad, K = dv.kl_nodes;
#define TRY(x) do { rc = (x); if (rc) { rlc_h_destroy(h); return rc; } } while (0)
    TRY(rlc_h_malloc(h, &dv.theta, NA * PP));
    TRY(rlc_h_malloc(h, &dv.theta_t, NA * PP));
    TRY(rlc_h_malloc(h, &dv.m, NA * PP));
    TRY(rlc_h_malloc(h, &dv.v, NA * PP));
    dv.pw = nullptr;
    TRY(rlc_h_malloc(h, &dv.kl_step, NA));
    float *lp, *lq, *al, *na, *nw;
    TRY(rlc_h_malloc(h, &lp, NA)); TRY(rlc_h_malloc(h, &lq, NA)); TRY(rlc_h_malloc(h, &al, NA));
    TRY(rlc_h_malloc(h, &na, K * (size_t)cfg->action_dim)); TRY(rlc_h_malloc(h, &nw, K));
    dv.pi_lr = lp; dv.qv_lr = lq; dv.alpha = al; dv.kl_node_a = na; dv.kl_node_w = nw;
    TRY(rlc_h_malloc(h, &dv.noise_ctr, NA));
    TRY(rlc_h_malloc(h, &dv.tap_q, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_v, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_logp, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_qpi, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_loss, NA * 4));
    TRY(rlc_h_malloc(h, &dv.kl_tap_iq, NA * (size_t)cfg->batch_size * K));
    dv.tap_g = nullptr;
    {
        size_t need = rlc_kl_scratch_floats(dv.d, dv.kl_nodes);
        if (rlc_kl_mfma_supported(dv.d, dv.kl_nodes)) need = std::max(need, rlc_kl_mfma_scratch_floats(dv.d, dv.kl_nodes));
        dv.scratch_stride = (long long)((need + 63) & ~(size_t)63);
    }
    TRY(rlc_h_malloc(h, &dv.scratch, NA * (size_t)dv.scratch_stride, false));
#undef TRY
    hipError_t e = hipMemcpyAsync(lp, cfg->pi_lr, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(lq, cfg->qf_vf_lr, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(al, cfg->entropy_scale, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess && K)
        e = hipMemcpyAsync(na, cfg->node_actions, K * cfg->action_dim * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess && K) e = hipMemcpyAsync(nw, cfg->node_weights, K * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) {
        rlc_set_error("rlc_kl_create: upload failed: %s", hipGetErrorString(e));
        rlc_h_destroy(h);
        return 1;
    }
    *out = h;
    return 0;
}

int rlc_kl_set_step(rlc_handle* h, int32_t agent, int32_t step) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_KL(h);
    RLC_REQUIRE(step >= 0, "negative step");
    RLC_HIP(hipMemcpyAsync(h->sac.kl_step + agent, &step, sizeof(int), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_kl_get_step(rlc_handle* h, int32_t agent, int32_t* step) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_KL(h);
    RLC_REQUIRE(step, "null step");
    RLC_HIP(hipMemcpyAsync(step, h->sac.kl_step + agent, sizeof(int), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_kl_set_kernel(rlc_handle* h, int32_t variant) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_KL(h);
    RLC_REQUIRE(variant >= 0 && variant <= 2, "kernel variant must be 0 (auto), 1 (generic) or 2 (mfma)");
    RLC_REQUIRE(variant != 2 || rlc_kl_mfma_supported(h->sac.d, h->sac.kl_nodes),
                "MFMA KL kernel does not support these dimensions");
    h->variant = variant;
    if (rlc_h_kl_variant(h) != 2) h->split_c = 1;        // latency mode belongs to the MFMA kernel
    return rlc_h_sac_relayout(h, rlc_h_kl_variant(h) == 2 ? 1 : 0);
}

int rlc_kl_set_split(rlc_handle* h, int32_t n_workgroups) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_KL(h);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n_workgroups >= 1 && n_workgroups <= 8, "workgroups per agent must be in [1,8]");
    RLC_REQUIRE(!h->has_env, "the on-device experiment loop runs the one-workgroup kernels");
    if (n_workgroups == 1) { h->split_c = 1; return 0; }
    RLC_REQUIRE(rlc_h_kl_variant(h) == 2, "latency mode is a variant of the MFMA kernel (these dimensions run the any-shape one)");
    RLC_REQUIRE(h->sac.kl_optim == RLC_KL_OPTIM_INTG || h->sac.kl_optim == RLC_KL_OPTIM_HARD_INTG,
                "latency mode splits the action integral; the 'll' updates have none");
    hipDeviceProp_t prop;
    RLC_HIP(hipGetDeviceProperties(&prop, h->device));
    const int grid = rlc_kl_split_grid(h->sac.n_agents, n_workgroups);
    RLC_REQUIRE(grid <= prop.multiProcessorCount, "%d agents x %d workgroups need %d co-resident workgroups; the GPU has %d CUs",
                h->sac.n_agents, n_workgroups, grid, prop.multiProcessorCount);
    if (!h->split_bar) {
        if (rlc_h_malloc(h, &h->split_bar, (size_t)h->sac.n_agents) || rlc_h_malloc(h, &h->split_err, (size_t)1)) return 1;
        if (rlc_h_malloc(h, &h->split_part, (size_t)h->sac.n_agents * rlc_kl_split_zbuf_floats(h->sac.d))) return 1;
    }
    h->split_c = n_workgroups;
    h->split_poisoned = false;           // re-armed by the caller
    return 0;
}

int rlc_kl_get_kernel(const rlc_handle* h, int32_t* variant_in_use) {
    RLC_REQUIRE(h && variant_in_use, "null argument");
    RLC_NEED_KL(h);
    *variant_in_use = rlc_h_kl_variant(h);
    return 0;
}

// ---- the KL names of the bodies shared with SoftActorCritic (rlc_api_sac.hip) ----
int rlc_kl_param_count(const rlc_handle* h, int64_t* out_p) { return rlc_sacfam_param_count(RLC_ALGO_KL, h, out_p); }
int rlc_kl_set_blob(rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n) {
    return rlc_sacfam_set_blob(RLC_ALGO_KL, h, agent, which, src, n);
}
int rlc_kl_get_blob(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    return rlc_sacfam_get_blob(RLC_ALGO_KL, h, agent, which, dst, n);
}
int rlc_kl_init_target(rlc_handle* h, int32_t agent) { return rlc_sacfam_init_target(RLC_ALGO_KL, h, agent); }
int rlc_kl_act(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps,
               float* out_actions) {
    return rlc_sacfam_act(RLC_ALGO_KL, h, first_agent, n, states, sample, eps, out_actions);
}
int rlc_kl_act_queue(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps) {
    return rlc_sacfam_act_queue(RLC_ALGO_KL, h, first_agent, n, states, sample, eps);
}
int rlc_kl_act_fetch(rlc_handle* h, int32_t first_agent, int32_t n, float* out_actions) {
    return rlc_sacfam_act_fetch(RLC_ALGO_KL, h, first_agent, n, out_actions);
}
int rlc_kl_update(rlc_handle* h, int32_t n_updates, const int64_t* host_indices, const float* eps) {
    return rlc_sacfam_update(RLC_ALGO_KL, h, n_updates, host_indices, eps);
}
int rlc_kl_update_batch(rlc_handle* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                        const double* next_states, const double* rewards, const double* gammas, const float* eps) {
    return rlc_sacfam_update_batch(RLC_ALGO_KL, h, agent, batch, states, actions, next_states, rewards, gammas, eps);
}
int rlc_kl_enable_grad_taps(rlc_handle* h, int32_t on) { return rlc_sacfam_enable_grad_taps(RLC_ALGO_KL, h, on); }
int rlc_kl_last_tap(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    return rlc_sacfam_last_tap(RLC_ALGO_KL, h, agent, which, dst, n);
}

}  // extern "C"
