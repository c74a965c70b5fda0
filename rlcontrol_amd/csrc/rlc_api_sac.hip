// rlc_api_sac.hip -- C ABI of the SoftActorCritic population (declared in include/rlcontrol_hip.h).
#include <string.h>

#include <algorithm>

#include "rlc_handle.h"

#define RLC_NEED_SAC(h) RLC_REQUIRE((h) && (h)->algo == RLC_ALGO_SAC, "handle is not a SoftActorCritic population")
// the bodies below serve the SoftActorCritic handles and the ReverseKL / ForwardKL handles (rlc_api_kl.hip), which
// share the device view RlcSacDev: `algo` is the one the calling entry point belongs to
#define RLC_NEED_ALGO(h, algo)                                                                                  \
    RLC_REQUIRE((h) && (h)->algo == (algo), "handle is not a %s population",                                    \
                (algo) == RLC_ALGO_SAC ? "SoftActorCritic" : "ReverseKL / ForwardKL")

namespace {

float* sac_blob(rlc_handle* h, int which) {
    switch (which) {
        case 0: return h->sac.theta;
        case 1: return h->sac.theta_t;
        case 2: return h->sac.m;
        case 3: return h->sac.v;
        default: return nullptr;
    }
}

int sac_fetch_blob(rlc_handle* h, const float* dev_src, float* dst) {
    const RlcSacDims& d = h->sac.d;
    std::vector<float> padded(d.Ppad);
    RLC_HIP(hipMemcpyAsync(padded.data(), dev_src, sizeof(float) * d.Ppad, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    rlc_unpack_segs(d, padded.data(), dst);
    return 0;
}

// Re-pack the four per-agent blobs when the kernel variant (and with it the weight layout) changes.
int sac_relayout(rlc_handle* h, int blocked) {
    if (h->sac.d.blocked == blocked) return 0;
    if (rlc_h_use_device(h)) return 1;
    const RlcSacDims od = h->sac.d;
    const RlcSacDims nd = rlc_sac_make_dims(od.S, od.A, od.L1A, od.L2A, od.L1C, od.L2C, od.B, blocked, od.qcat, od.norm);
    const size_t NA = h->sac.n_agents, PP = od.Ppad;
    std::vector<float> dev(NA * PP), compact(od.P), out(NA * PP);
    for (int which = 0; which < 4; which++) {
        float* base = sac_blob(h, which);
        RLC_HIP(hipMemcpyAsync(dev.data(), base, sizeof(float) * NA * PP, hipMemcpyDeviceToHost, h->st));
        RLC_HIP(hipStreamSynchronize(h->st));
        std::fill(out.begin(), out.end(), 0.0f);
        for (size_t a = 0; a < NA; a++) {
            rlc_unpack_segs(od, &dev[a * PP], compact.data());
            rlc_pack_segs(nd, compact.data(), &out[a * PP]);
        }
        RLC_HIP(hipMemcpyAsync(base, out.data(), sizeof(float) * NA * PP, hipMemcpyHostToDevice, h->st));
        RLC_HIP(hipStreamSynchronize(h->st));
    }
    h->sac.d = nd;
    return 0;
}

// host eps [count] -> device buffer behind the index upload area; returns device pointer (or null if eps null)
int upload_eps(rlc_handle* h, const float* eps, size_t count, const float** out_dev, size_t idx_count) {
    *out_dev = nullptr;
    if (!eps) return 0;
    // layout of idx_dev: [idx_count long long][count floats]
    const size_t need_ll = idx_count + (count * sizeof(float) + 7) / 8;
    if (rlc_h_ensure_idx(h, need_ll)) return 1;
    float* dst = (float*)(h->idx_dev + idx_count);
    RLC_HIP(hipMemcpyAsync(dst, eps, sizeof(float) * count, hipMemcpyHostToDevice, h->st));
    *out_dev = dst;
    return 0;
}

}  // namespace

int rlc_h_sac_relayout(rlc_handle* h, int blocked) { return sac_relayout(h, blocked); }

int rlc_h_sac_launch_update(rlc_handle* h, int first, int n, int n_updates, int source, const long long* idx_dev,
                            const float* eps_dev, const RlcSacRollout* rollout) {
    if (h->algo == RLC_ALGO_KL) {
        if (rlc_h_kl_variant(h) == 2 && h->split_c > 1 && !rollout) {
            if (rlc_h_split_before_launch(h)) return 1;
            if (rlc_launch_kl_update_mfma_split(h->sac, h->split_part, h->split_bar, h->split_err, h->split_c, first, n,
                                                n_updates, source, idx_dev, eps_dev, h->grad_taps, h->st))
                return 1;
            return rlc_h_split_after_launch(h);
        }
        if (rlc_h_kl_variant(h) == 2)
            return rlc_launch_kl_update_mfma(h->sac, first, n, n_updates, source, idx_dev, eps_dev, h->grad_taps, h->st,
                                             rollout);
        return rlc_launch_kl_update(h->sac, first, n, n_updates, source, idx_dev, eps_dev, h->grad_taps, h->st, rollout);
    }
    if (rlc_h_sac_variant(h) == 2) {
        RLC_REQUIRE(rlc_sac_mfma_supported(h->sac.d), "MFMA SAC kernel does not support these dimensions");
        return rlc_launch_sac_update_mfma(h->sac, first, n, n_updates, source, idx_dev, eps_dev, h->grad_taps, h->st, rollout);
    }
    return rlc_launch_sac_update(h->sac, first, n, n_updates, source, idx_dev, eps_dev, h->grad_taps, h->st, rollout);
}

extern "C" {

int rlc_sac_create(const rlc_sac_config* cfg, rlc_handle** out) {
    RLC_REQUIRE(cfg && out, "null argument");
    RLC_REQUIRE(cfg->actor_l1_dim >= 1 && cfg->actor_l2_dim >= 1 && cfg->critic_l1_dim >= 1 && cfg->critic_l2_dim >= 1,
                "layer widths must be >= 1");
    RLC_REQUIRE(cfg->pi_lr && cfg->qf_vf_lr && cfg->entropy_scale, "null per-agent array");
    RLC_REQUIRE(cfg->norm_type == RLC_NORM_NONE || cfg->norm_type == RLC_NORM_LAYER,
                "norm_type %d: 'batch' (fused batch norm with moving averages, base_network.py:57-59) is not implemented",
                cfg->norm_type);
    const int norm = cfg->norm_type == RLC_NORM_LAYER ? 1 : 0;
    RLC_REQUIRE(!norm || (cfg->actor_l1_dim <= 1024 && cfg->actor_l2_dim <= 1024 && cfg->critic_l1_dim <= 1024 &&
                          cfg->critic_l2_dim <= 1024), "layer norm: layer widths must be <= 1024");
    rlc_handle* h = new rlc_handle();
    int rc = rlc_h_init_common(h, RLC_ALGO_SAC, cfg->device, cfg->n_agents, cfg->state_dim, cfg->action_dim,
                               cfg->batch_size, cfg->buffer_size, cfg->seed);
    if (rc) { rlc_h_destroy(h); return rc; }
    RlcSacDev& dv = h->sac;
    dv.d = rlc_sac_make_dims(cfg->state_dim, cfg->action_dim, cfg->actor_l1_dim, cfg->actor_l2_dim,
                             cfg->critic_l1_dim, cfg->critic_l2_dim, cfg->batch_size, 0, 0, norm);
    // the tile-blocked weight layout goes with the MFMA kernel (the default whenever it supports the shape)
    if (rlc_sac_mfma_supported(dv.d))
        dv.d = rlc_sac_make_dims(cfg->state_dim, cfg->action_dim, cfg->actor_l1_dim, cfg->actor_l2_dim,
                                 cfg->critic_l1_dim, cfg->critic_l2_dim, cfg->batch_size, 1, 0, norm);
    dv.rep = h->rep;
    dv.n_agents = cfg->n_agents;
    dv.clip_state = cfg->clip_state;
    dv.tau = cfg->tau;
    dv.smin0 = cfg->state_min0; dv.smax0 = cfg->state_max0; dv.amax0 = cfg->action_max0;
    const size_t NA = cfg->n_agents, PP = dv.d.Ppad;
#define TRY(x) do { rc = (x); if (rc) { rlc_h_destroy(h); return rc; } } while (0)
    TRY(rlc_h_malloc(h, &dv.theta, NA * PP));
    TRY(rlc_h_malloc(h, &dv.theta_t, NA * PP));
    TRY(rlc_h_malloc(h, &dv.m, NA * PP));
    TRY(rlc_h_malloc(h, &dv.v, NA * PP));
    TRY(rlc_h_malloc(h, &dv.pw, NA * 4));
    float *lp, *lq, *al;
    TRY(rlc_h_malloc(h, &lp, NA)); TRY(rlc_h_malloc(h, &lq, NA)); TRY(rlc_h_malloc(h, &al, NA));
    dv.pi_lr = lp; dv.qv_lr = lq; dv.alpha = al;
    TRY(rlc_h_malloc(h, &dv.noise_ctr, NA));
    TRY(rlc_h_malloc(h, &dv.tap_q, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_v, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_logp, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_qpi, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_loss, NA * 4));
    dv.tap_g = nullptr;
    dv.scratch_stride = (long long)((rlc_sac_scratch_floats(dv.d) + 63) & ~(size_t)63);
    TRY(rlc_h_malloc(h, &dv.scratch, NA * (size_t)dv.scratch_stride, false));
#undef TRY
    std::vector<float> pw(NA * 4);
    for (size_t i = 0; i < NA; i++) { pw[4 * i] = 0.9f; pw[4 * i + 1] = 0.999f; pw[4 * i + 2] = 0.9f; pw[4 * i + 3] = 0.999f; }
    hipError_t e = hipMemcpyAsync(dv.pw, pw.data(), NA * 4 * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(lp, cfg->pi_lr, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(lq, cfg->qf_vf_lr, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(al, cfg->entropy_scale, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) {
        rlc_set_error("rlc_sac_create: upload failed: %s", hipGetErrorString(e));
        rlc_h_destroy(h);
        return 1;
    }
    *out = h;
    return 0;
}

int rlc_sacfam_param_count(int algo, const rlc_handle* h, int64_t* out_p) {
    RLC_REQUIRE(h && out_p, "null argument");
    RLC_NEED_ALGO(h, algo);
    *out_p = h->sac.d.P;
    return 0;
}

int rlc_sacfam_set_blob(int algo, rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ALGO(h, algo);
    float* base = sac_blob(h, which);
    RLC_REQUIRE(base && src, "bad blob selector %d or null src", which);
    const RlcSacDims& d = h->sac.d;
    RLC_REQUIRE(n == d.P, "blob length %lld != parameter count %d", (long long)n, d.P);
    std::vector<float> padded(d.Ppad, 0.0f);
    rlc_pack_segs(d, src, padded.data());
    RLC_HIP(hipMemcpyAsync(base + (size_t)agent * d.Ppad, padded.data(), sizeof(float) * d.Ppad, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_sacfam_get_blob(int algo, rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ALGO(h, algo);
    float* base = sac_blob(h, which);
    RLC_REQUIRE(base && dst, "bad blob selector %d or null dst", which);
    RLC_REQUIRE(n == h->sac.d.P, "blob length %lld != parameter count %d", (long long)n, h->sac.d.P);
    return sac_fetch_blob(h, base + (size_t)agent * h->sac.d.Ppad, dst);
}

int rlc_sac_set_beta_powers(rlc_handle* h, int32_t agent, const float* pw4) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_SAC(h);
    RLC_REQUIRE(pw4, "null pw4");
    RLC_HIP(hipMemcpyAsync(h->sac.pw + agent * 4, pw4, 4 * sizeof(float), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_sac_get_beta_powers(rlc_handle* h, int32_t agent, float* pw4) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_SAC(h);
    RLC_REQUIRE(pw4, "null pw4");
    RLC_HIP(hipMemcpyAsync(pw4, h->sac.pw + agent * 4, 4 * sizeof(float), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_sacfam_init_target(int algo, rlc_handle* h, int32_t agent) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ALGO(h, algo);
    const size_t off = (size_t)agent * h->sac.d.Ppad;
    RLC_HIP(hipMemcpyAsync(h->sac.theta_t + off, h->sac.theta + off, h->sac.d.Ppad * sizeof(float),
                           hipMemcpyDeviceToDevice, h->st));
    return 0;
}

int rlc_sacfam_act(int algo, rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps,
                float* out_actions) {
    RLC_NEED_ALGO(h, algo);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->sac.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states && out_actions, "null array");
    const size_t S = h->sac.d.S, A = h->sac.d.A;
    const size_t in_f = n * S, eps_f = (sample && eps) ? n * A : 0, out_f = n * A;
    if (rlc_h_ensure_io(h, sizeof(float) * (in_f + eps_f + out_f))) return 1;
    float* hin = (float*)h->io_host;
    for (size_t i = 0; i < in_f; i++) hin[i] = (float)states[i];
    for (size_t i = 0; i < eps_f; i++) hin[in_f + i] = eps[i];
    RLC_HIP(hipMemcpyAsync(h->io_dev, hin, sizeof(float) * (in_f + eps_f), hipMemcpyHostToDevice, h->st));
    float* dout = h->io_dev + in_f + eps_f;
    if ((algo == RLC_ALGO_KL ? rlc_launch_kl_act : rlc_launch_sac_act)(h->sac, first_agent, n, h->io_dev,
                                                                        eps_f ? h->io_dev + in_f : nullptr,
                                                                        sample ? 1 : 0, dout, h->st, nullptr, 0))
        return 1;
    RLC_HIP(hipMemcpyAsync(hin + in_f + eps_f, dout, sizeof(float) * out_f, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    memcpy(out_actions, hin + in_f + eps_f, sizeof(float) * out_f);
    return 0;
}

// the acting forward queued behind the update that was just launched (see rlc_ddpg_act_queue, rlc_api.hip)
int rlc_sacfam_act_queue(int algo, rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample,
                         const float* eps) {
    RLC_NEED_ALGO(h, algo);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->sac.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states, "null array");
    const size_t S = h->sac.d.S, A = h->sac.d.A;
    const size_t in_f = n * S, eps_f = (sample && eps) ? n * A : 0, out_f = n * A;
    if (rlc_h_aq_begin(h, in_f + eps_f + out_f, n == 1)) return 1;
    for (size_t i = 0; i < in_f; i++) h->aq_host[i] = (float)states[i];
    for (size_t i = 0; i < eps_f; i++) h->aq_host[in_f + i] = eps[i];
    if ((algo == RLC_ALGO_KL ? rlc_launch_kl_act : rlc_launch_sac_act)(h->sac, first_agent, n, h->aq_host,
                                                                        eps_f ? h->aq_host + in_f : nullptr, sample ? 1 : 0,
                                                                        h->aq_host + in_f + eps_f, h->st, rlc_h_aq_flag(h),
                                                                        h->aq_seq))
        return 1;
    h->aq_first = first_agent; h->aq_n = n;
    h->aq_out = in_f + eps_f;
    return 0;
}

int rlc_sacfam_act_fetch(int algo, rlc_handle* h, int32_t first_agent, int32_t n, float* out_actions) {
    RLC_NEED_ALGO(h, algo);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(out_actions, "null array");
    if (rlc_h_aq_wait(h, first_agent, n)) return 1;
    memcpy(out_actions, h->aq_host + h->aq_out, sizeof(float) * n * h->sac.d.A);
    return 0;
}

int rlc_sacfam_update(int algo, rlc_handle* h, int32_t n_updates, const int64_t* host_indices, const float* eps) {
    RLC_NEED_ALGO(h, algo);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n_updates >= 0, "negative n_updates");
    if (n_updates == 0) return 0;
    const int B = h->B, NA = h->sac.n_agents, A = h->sac.d.A;
    for (int a = 0; a < NA; a++)   // utils/replaybuffer.py:34
        RLC_REQUIRE(h->ring[a].size >= B, "agent %d: replay holds %lld transitions < batch_size %d", a, h->ring[a].size, B);
    int source = RLC_SRC_REPLAY_DEVICE_SAMPLER;
    const size_t count = (size_t)NA * n_updates * B;
    const size_t idx_count = host_indices ? count : 0;
    const float* eps_dev = nullptr;
    if (host_indices) {
        for (int a = 0; a < NA; a++) {
            const long long size = h->ring[a].size;
            const int64_t* p = host_indices + (size_t)a * n_updates * B;
            for (size_t i = 0; i < (size_t)n_updates * B; i++)
                RLC_REQUIRE(p[i] >= 0 && p[i] < size, "agent %d: sample index %lld out of range (size %lld)", a,
                            (long long)p[i], size);
        }
        if (rlc_h_ensure_idx(h, count + (eps ? (count * A * sizeof(float) + 7) / 8 : 0))) return 1;
        RLC_HIP(hipMemcpyAsync(h->idx_dev, host_indices, sizeof(long long) * count, hipMemcpyHostToDevice, h->st));
        source = RLC_SRC_REPLAY_HOST_INDICES;
    }
    if (upload_eps(h, eps, count * A, &eps_dev, idx_count)) return 1;
    return rlc_h_sac_launch_update(h, 0, NA, n_updates, source, h->idx_dev, eps_dev, nullptr);
}

int rlc_sacfam_update_batch(int algo, rlc_handle* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                         const double* next_states, const double* rewards, const double* gammas, const float* eps) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ALGO(h, algo);
    RLC_REQUIRE(batch == h->B, "minibatch has %d rows; the handle was created for batch_size %d", batch, h->B);
    RLC_REQUIRE(states && actions && next_states && rewards && gammas, "null minibatch array");
    const size_t S = h->sac.d.S, A = h->sac.d.A, B = batch;
    const size_t fbytes = sizeof(float) * B * (2 * S + A), dbytes = sizeof(double) * 2 * B;
    if (rlc_h_ensure_io(h, fbytes + dbytes)) return 1;
    RLC_HIP(hipStreamSynchronize(h->st));
    double* hd = (double*)h->io_host;
    float* hf = (float*)(hd + 2 * B);
    for (size_t i = 0; i < B; i++) { hd[i] = rewards[i]; hd[B + i] = gammas[i]; }
    for (size_t i = 0; i < B * S; i++) { hf[i] = (float)states[i]; hf[B * S + i] = (float)next_states[i]; }
    for (size_t i = 0; i < B * A; i++) hf[2 * B * S + i] = (float)actions[i];
    const size_t slot = (size_t)agent * RLC_MAX_BATCH;
    RLC_HIP(hipMemcpyAsync(h->rep.gr + slot, hd, sizeof(double) * B, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipMemcpyAsync(h->rep.gg + slot, hd + B, sizeof(double) * B, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipMemcpyAsync(h->rep.gs + slot * S, hf, sizeof(float) * B * S, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipMemcpyAsync(h->rep.gs2 + slot * S, hf + B * S, sizeof(float) * B * S, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipMemcpyAsync(h->rep.ga + slot * A, hf + 2 * B * S, sizeof(float) * B * A, hipMemcpyHostToDevice, h->st));
    const float* eps_dev = nullptr;
    if (upload_eps(h, eps, B * A, &eps_dev, 0)) return 1;
    h->io_pending = true;
    return rlc_h_sac_launch_update(h, agent, 1, 1, RLC_SRC_STAGING, nullptr, eps_dev, nullptr);
}

int rlc_sac_set_kernel(rlc_handle* h, int32_t variant) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_SAC(h);
    RLC_REQUIRE(variant >= 0 && variant <= 2, "kernel variant must be 0 (auto), 1 (generic) or 2 (mfma)");
    RLC_REQUIRE(variant != 2 || rlc_sac_mfma_supported(h->sac.d), "MFMA SAC kernel does not support these dimensions");
    RLC_REQUIRE(!h->has_env, "the kernel variant cannot change once a rollout is attached to the handle");
    h->variant = variant;
    return sac_relayout(h, rlc_h_sac_variant(h) == 2 ? 1 : 0);
}

int rlc_sac_get_kernel(const rlc_handle* h, int32_t* variant_in_use) {
    RLC_REQUIRE(h && variant_in_use, "null argument");
    RLC_NEED_SAC(h);
    *variant_in_use = rlc_h_sac_variant(h);
    return 0;
}

int rlc_sacfam_enable_grad_taps(int algo, rlc_handle* h, int32_t on) {
    RLC_NEED_ALGO(h, algo);
    if (rlc_h_use_device(h)) return 1;
    if (on && !h->sac.tap_g) {
        if (rlc_h_malloc(h, &h->sac.tap_g, (size_t)h->sac.n_agents * h->sac.d.Ppad)) return 1;
    }
    h->grad_taps = on ? 1 : 0;
    return 0;
}

int rlc_sacfam_last_tap(int algo, rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_ALGO(h, algo);
    RLC_REQUIRE(dst, "null dst");
    const int B = h->B, P = h->sac.d.P;
    const float* src = nullptr;
    long long want = 0;
    switch (which) {
        case 0: src = h->sac.tap_q + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 1: src = h->sac.tap_v + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 2: src = h->sac.tap_logp + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 3: src = h->sac.tap_qpi + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 4: src = h->sac.tap_loss + (size_t)agent * 4; want = 3; break;
        case 5: src = h->sac.tap_g ? h->sac.tap_g + (size_t)agent * h->sac.d.Ppad : nullptr; want = P; break;
        case 6:
            if (algo == RLC_ALGO_KL && h->sac.kl_tap_iq) {
                want = (long long)B * h->sac.kl_nodes;
                src = h->sac.kl_tap_iq + (size_t)agent * want;
            }
            break;
        default: break;
    }
    RLC_REQUIRE(src, "tap %d not available (gradient taps need *_enable_grad_taps)", which);
    RLC_REQUIRE(n == want, "tap %d holds %lld floats, caller asked for %lld", which, want, (long long)n);
    if (which == 5) return sac_fetch_blob(h, src, dst);
    RLC_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

// ---- the SoftActorCritic names of the shared bodies ----
int rlc_sac_param_count(const rlc_handle* h, int64_t* out_p) { return rlc_sacfam_param_count(RLC_ALGO_SAC, h, out_p); }
int rlc_sac_set_blob(rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n) {
    return rlc_sacfam_set_blob(RLC_ALGO_SAC, h, agent, which, src, n);
}
int rlc_sac_get_blob(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    return rlc_sacfam_get_blob(RLC_ALGO_SAC, h, agent, which, dst, n);
}
int rlc_sac_init_target(rlc_handle* h, int32_t agent) { return rlc_sacfam_init_target(RLC_ALGO_SAC, h, agent); }
int rlc_sac_act(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps,
                float* out_actions) {
    return rlc_sacfam_act(RLC_ALGO_SAC, h, first_agent, n, states, sample, eps, out_actions);
}
int rlc_sac_act_queue(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps) {
    return rlc_sacfam_act_queue(RLC_ALGO_SAC, h, first_agent, n, states, sample, eps);
}
int rlc_sac_act_fetch(rlc_handle* h, int32_t first_agent, int32_t n, float* out_actions) {
    return rlc_sacfam_act_fetch(RLC_ALGO_SAC, h, first_agent, n, out_actions);
}
int rlc_sac_update(rlc_handle* h, int32_t n_updates, const int64_t* host_indices, const float* eps) {
    return rlc_sacfam_update(RLC_ALGO_SAC, h, n_updates, host_indices, eps);
}
int rlc_sac_update_batch(rlc_handle* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                         const double* next_states, const double* rewards, const double* gammas, const float* eps) {
    return rlc_sacfam_update_batch(RLC_ALGO_SAC, h, agent, batch, states, actions, next_states, rewards, gammas, eps);
}
int rlc_sac_enable_grad_taps(rlc_handle* h, int32_t on) { return rlc_sacfam_enable_grad_taps(RLC_ALGO_SAC, h, on); }
int rlc_sac_last_tap(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    return rlc_sacfam_last_tap(RLC_ALGO_SAC, h, agent, which, dst, n);
}

}  // extern "C"
