#include <vector>
// rlc_api_naf.hip -- C ABI of the NAF population (declared in include/rlcontrol_hip.h).
#include <string.h>

#include <algorithm>

#include "rlc_handle.h"

#define RLC_NEED_NAF(h) RLC_REQUIRE((h) && (h)->algo == RLC_ALGO_NAF, "handle is not a NAF population")

namespace {

float* naf_blob(rlc_handle* h, int which) {
    switch (which) {
        case 0: return h->naf.theta;
        case 1: return h->naf.theta_t;
        case 2: return h->naf.m;
        case 3: return h->naf.v;
        default: return nullptr;
    }
}

int naf_fetch_blob(rlc_handle* h, const float* dev_src, float* dst) {
    const RlcNafDims& d = h->naf.d;
    std::vector<float> padded(d.Ppad);
    RLC_HIP(hipMemcpyAsync(padded.data(), dev_src, sizeof(float) * d.Ppad, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    rlc_unpack_segs(d, padded.data(), dst);
    return 0;
}

// Re-pack the four per-agent blobs when the kernel variant (and with it the weight layout) changes.
int naf_relayout(rlc_handle* h, int blocked) {
    if (h->naf.d.blocked == blocked) return 0;
    if (rlc_h_use_device(h)) return 1;
    const RlcNafDims od = h->naf.d;
    const RlcNafDims nd = rlc_naf_make_dims(od.S, od.A, od.L1, od.L2, od.B, blocked, od.norm);
    const size_t NA = h->naf.n_agents, PP = od.Ppad;
    std::vector<float> dev(NA * PP), compact(od.P), out(NA * PP);
    for (int which = 0; which < 4; which++) {
        float* base = naf_blob(h, which);
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
    h->naf.d = nd;
    return 0;
}

}  // namespace

int rlc_h_naf_launch_update(rlc_handle* h, int first, int n, int n_updates, int source, const long long* idx_dev,
                            const RlcNafRollout* rollout) {
    if (rlc_h_naf_variant(h) == 2) {
        RLC_REQUIRE(rlc_naf_mfma_supported(h->naf.d), "MFMA NAF kernel does not support these dimensions");
        return rlc_launch_naf_update_mfma(h->naf, first, n, n_updates, source, idx_dev, h->grad_taps, h->st, rollout);
    }
    return rlc_launch_naf_update(h->naf, first, n, n_updates, source, idx_dev, h->grad_taps, h->st, rollout);
}

extern "C" {

int rlc_naf_create(const rlc_naf_config* cfg, rlc_handle** out) {
    RLC_REQUIRE(cfg && out, "null argument");
    RLC_REQUIRE(cfg->l1_dim >= 1 && cfg->l2_dim >= 1, "layer widths must be >= 1");
    RLC_REQUIRE(cfg->action_dim <= RLC_NAF_MAX_A, "NAF supports action_dim <= %d (got %d)", RLC_NAF_MAX_A, cfg->action_dim);
    RLC_REQUIRE(cfg->state_min && cfg->state_max && cfg->action_max && cfg->learning_rate, "null array");
    RLC_REQUIRE(cfg->norm_type == RLC_NORM_NONE || cfg->norm_type == RLC_NORM_LAYER,
                "norm_type %d: 'batch' (fused batch norm with moving averages, base_network.py:57-59) is not implemented",
                cfg->norm_type);
    const int norm = cfg->norm_type == RLC_NORM_LAYER ? 1 : 0;
    RLC_REQUIRE(!norm || (cfg->l1_dim <= 1024 && cfg->l2_dim <= 1024), "layer norm: layer widths must be <= 1024");
    rlc_handle* h = new rlc_handle();
    int rc = rlc_h_init_common(h, RLC_ALGO_NAF, cfg->device, cfg->n_agents, cfg->state_dim, cfg->action_dim,
                               cfg->batch_size, cfg->buffer_size, cfg->seed);
    if (rc) { rlc_h_destroy(h); return rc; }
    RlcNafDev& dv = h->naf;
    dv.d = rlc_naf_make_dims(cfg->state_dim, cfg->action_dim, cfg->l1_dim, cfg->l2_dim, cfg->batch_size, 0, norm);
    // the tile-blocked weight layout goes with the MFMA kernel (the default whenever it supports the shape)
    if (rlc_naf_mfma_supported(dv.d))
        dv.d = rlc_naf_make_dims(cfg->state_dim, cfg->action_dim, cfg->l1_dim, cfg->l2_dim, cfg->batch_size, 1, norm);
    dv.rep = h->rep;
    dv.n_agents = cfg->n_agents;
    dv.clip_state = cfg->clip_state;
    dv.tau = cfg->tau;
    const size_t NA = cfg->n_agents, PP = dv.d.Ppad, S = dv.d.S, A = dv.d.A;
#define TRY(x) do { rc = (x); if (rc) { rlc_h_destroy(h); return rc; } } while (0)
    TRY(rlc_h_malloc(h, &dv.theta, NA * PP));
    TRY(rlc_h_malloc(h, &dv.theta_t, NA * PP));
    TRY(rlc_h_malloc(h, &dv.m, NA * PP));
    TRY(rlc_h_malloc(h, &dv.v, NA * PP));
    TRY(rlc_h_malloc(h, &dv.pw, NA * 2));
    float *lr, *smin, *smax, *amax, *amin;
    TRY(rlc_h_malloc(h, &lr, NA)); TRY(rlc_h_malloc(h, &smin, S)); TRY(rlc_h_malloc(h, &smax, S)); TRY(rlc_h_malloc(h, &amax, A));
    TRY(rlc_h_malloc(h, &amin, A));
    dv.lr = lr; dv.smin = smin; dv.smax = smax; dv.amax = amax; dv.amin = amin;
    TRY(rlc_h_malloc(h, &dv.tap_q, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_y, NA * RLC_MAX_BATCH));
    TRY(rlc_h_malloc(h, &dv.tap_V, NA * RLC_MAX_BATCH));
    dv.tap_g = nullptr;
    dv.scratch_stride = (long long)((rlc_naf_scratch_floats(dv.d) + 63) & ~(size_t)63);
    TRY(rlc_h_malloc(h, &dv.scratch, NA * (size_t)dv.scratch_stride, false));
#undef TRY
    std::vector<float> pw(NA * 2);
    for (size_t i = 0; i < NA; i++) { pw[2 * i] = 0.9f; pw[2 * i + 1] = 0.999f; }
    hipError_t e = hipMemcpyAsync(dv.pw, pw.data(), NA * 2 * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(lr, cfg->learning_rate, NA * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(smin, cfg->state_min, S * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(smax, cfg->state_max, S * sizeof(float), hipMemcpyHostToDevice, h->st);
    if (e == hipSuccess) e = hipMemcpyAsync(amax, cfg->action_max, A * sizeof(float), hipMemcpyHostToDevice, h->st);
    std::vector<float> amin_host(A);
    for (int j = 0; j < A; j++) amin_host[j] = cfg->action_min ? cfg->action_min[j] : -cfg->action_max[j];
    if (e == hipSuccess) e = hipMemcpy(amin, amin_host.data(), A * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) {
        rlc_set_error("rlc_naf_create: upload failed: %s", hipGetErrorString(e));
        rlc_h_destroy(h);
        return 1;
    }
    *out = h;
    return 0;
}

int rlc_naf_param_count(const rlc_handle* h, int64_t* out_p) {
    RLC_REQUIRE(h && out_p, "null argument");
    RLC_NEED_NAF(h);
    *out_p = h->naf.d.P;
    return 0;
}

int rlc_naf_set_blob(rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    float* base = naf_blob(h, which);
    RLC_REQUIRE(base && src, "bad blob selector %d or null src", which);
    const RlcNafDims& d = h->naf.d;
    RLC_REQUIRE(n == d.P, "blob length %lld != parameter count %d", (long long)n, d.P);
    std::vector<float> padded(d.Ppad, 0.0f);
    rlc_pack_segs(d, src, padded.data());
    RLC_HIP(hipMemcpyAsync(base + (size_t)agent * d.Ppad, padded.data(), sizeof(float) * d.Ppad, hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_naf_get_blob(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    float* base = naf_blob(h, which);
    RLC_REQUIRE(base && dst, "bad blob selector %d or null dst", which);
    RLC_REQUIRE(n == h->naf.d.P, "blob length %lld != parameter count %d", (long long)n, h->naf.d.P);
    return naf_fetch_blob(h, base + (size_t)agent * h->naf.d.Ppad, dst);
}

int rlc_naf_get_beta_powers(rlc_handle* h, int32_t agent, float* pw2) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    RLC_REQUIRE(pw2, "null pw2");
    RLC_HIP(hipMemcpyAsync(pw2, h->naf.pw + agent * 2, 2 * sizeof(float), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_naf_init_target(rlc_handle* h, int32_t agent) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    const size_t off = (size_t)agent * h->naf.d.Ppad;
    RLC_HIP(hipMemcpyAsync(h->naf.theta_t + off, h->naf.theta + off, h->naf.d.Ppad * sizeof(float),
                           hipMemcpyDeviceToDevice, h->st));
    return 0;
}

int rlc_naf_act(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, float* out_mu, float* out_lcols) {
    RLC_NEED_NAF(h);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->naf.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states && out_mu, "null array");
    const size_t S = h->naf.d.S, A = h->naf.d.A, NL = A * (A + 1) / 2;
    const size_t in_f = n * S, mu_f = n * A, lc_f = n * NL;
    if (rlc_h_ensure_io(h, sizeof(float) * (in_f + mu_f + lc_f))) return 1;
    float* hin = (float*)h->io_host;
    for (size_t i = 0; i < in_f; i++) hin[i] = (float)states[i];
    RLC_HIP(hipMemcpyAsync(h->io_dev, hin, sizeof(float) * in_f, hipMemcpyHostToDevice, h->st));
    float* dmu = h->io_dev + in_f;
    float* dlc = dmu + mu_f;
    if (rlc_launch_naf_act(h->naf, first_agent, n, h->io_dev, dmu, dlc, h->st)) return 1;
    RLC_HIP(hipMemcpyAsync(hin + in_f, dmu, sizeof(float) * (mu_f + lc_f), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    memcpy(out_mu, hin + in_f, sizeof(float) * mu_f);
    if (out_lcols) memcpy(out_lcols, hin + in_f + mu_f, sizeof(float) * lc_f);
    return 0;
}

// the acting forward queued behind the update that was just launched (see rlc_ddpg_act_queue, rlc_api.hip)
int rlc_naf_act_queue(rlc_handle* h, int32_t first_agent, int32_t n, const double* states) {
    RLC_NEED_NAF(h);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->naf.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states, "null array");
    const size_t S = h->naf.d.S, A = h->naf.d.A, NL = A * (A + 1) / 2;
    const size_t in_f = n * S, mu_f = n * A, lc_f = n * NL;
    if (rlc_h_aq_begin(h, in_f + mu_f + lc_f, n == 1)) return 1;
    for (size_t i = 0; i < in_f; i++) h->aq_host[i] = (float)states[i];
    if (rlc_launch_naf_act(h->naf, first_agent, n, h->aq_host, h->aq_host + in_f, h->aq_host + in_f + mu_f, h->st,
                           rlc_h_aq_flag(h), h->aq_seq))
        return 1;
    h->aq_first = first_agent; h->aq_n = n;
    h->aq_out = in_f;
    return 0;
}

int rlc_naf_act_fetch(rlc_handle* h, int32_t first_agent, int32_t n, float* out_mu, float* out_lcols) {
    RLC_NEED_NAF(h);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(out_mu, "null array");
    if (rlc_h_aq_wait(h, first_agent, n)) return 1;
    const size_t A = h->naf.d.A, NL = A * (A + 1) / 2;
    memcpy(out_mu, h->aq_host + h->aq_out, sizeof(float) * n * A);
    if (out_lcols) memcpy(out_lcols, h->aq_host + h->aq_out + (size_t)n * A, sizeof(float) * n * NL);
    return 0;
}

int rlc_naf_update(rlc_handle* h, int32_t n_updates, const int64_t* host_indices) {
    RLC_NEED_NAF(h);
    if (rlc_h_use_device(h)) return 1;
    RLC_REQUIRE(n_updates >= 0, "negative n_updates");
    if (n_updates == 0) return 0;
    const int B = h->B, NA = h->naf.n_agents;
    for (int a = 0; a < NA; a++)
        RLC_REQUIRE(h->ring[a].size >= B, "agent %d: replay holds %lld transitions < batch_size %d", a, h->ring[a].size, B);
    int source = RLC_SRC_REPLAY_DEVICE_SAMPLER;
    if (host_indices) {
        const size_t count = (size_t)NA * n_updates * B;
        for (int a = 0; a < NA; a++) {
            const long long size = h->ring[a].size;
            const int64_t* p = host_indices + (size_t)a * n_updates * B;
            for (size_t i = 0; i < (size_t)n_updates * B; i++)
                RLC_REQUIRE(p[i] >= 0 && p[i] < size, "agent %d: sample index %lld out of range (size %lld)", a,
                            (long long)p[i], size);
        }
        if (rlc_h_ensure_idx(h, count)) return 1;
        RLC_HIP(hipMemcpyAsync(h->idx_dev, host_indices, sizeof(long long) * count, hipMemcpyHostToDevice, h->st));
        source = RLC_SRC_REPLAY_HOST_INDICES;
    }
    return rlc_h_naf_launch_update(h, 0, NA, n_updates, source, h->idx_dev, nullptr);
}

int rlc_naf_update_batch(rlc_handle* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                         const double* next_states, const double* rewards, const double* gammas) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    RLC_REQUIRE(batch == h->B, "minibatch has %d rows; the handle was created for batch_size %d", batch, h->B);
    RLC_REQUIRE(states && actions && next_states && rewards && gammas, "null minibatch array");
    const size_t S = h->naf.d.S, A = h->naf.d.A, B = batch;
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
    h->io_pending = true;
    return rlc_h_naf_launch_update(h, agent, 1, 1, RLC_SRC_STAGING, nullptr, nullptr);
}

int rlc_naf_set_kernel(rlc_handle* h, int32_t variant) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_NAF(h);
    RLC_REQUIRE(variant >= 0 && variant <= 2, "kernel variant must be 0 (auto), 1 (generic) or 2 (mfma)");
    RLC_REQUIRE(variant != 2 || rlc_naf_mfma_supported(h->naf.d), "MFMA NAF kernel does not support these dimensions");
    RLC_REQUIRE(!h->has_env, "the kernel variant cannot change once a rollout is attached to the handle");
    h->variant = variant;
    return naf_relayout(h, rlc_h_naf_variant(h) == 2 ? 1 : 0);
}

int rlc_naf_get_kernel(const rlc_handle* h, int32_t* variant_in_use) {
    RLC_REQUIRE(h && variant_in_use, "null argument");
    RLC_NEED_NAF(h);
    *variant_in_use = rlc_h_naf_variant(h);
    return 0;
}

int rlc_naf_enable_grad_taps(rlc_handle* h, int32_t on) {
    RLC_NEED_NAF(h);
    if (rlc_h_use_device(h)) return 1;
    if (on && !h->naf.tap_g) {
        if (rlc_h_malloc(h, &h->naf.tap_g, (size_t)h->naf.n_agents * h->naf.d.Ppad)) return 1;
    }
    h->grad_taps = on ? 1 : 0;
    return 0;
}

int rlc_naf_last_tap(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (rlc_h_check_agent(h, agent) || rlc_h_use_device(h)) return 2;
    RLC_NEED_NAF(h);
    RLC_REQUIRE(dst, "null dst");
    const int B = h->B, P = h->naf.d.P;
    const float* src = nullptr;
    long long want = 0;
    switch (which) {
        case 0: src = h->naf.tap_q + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 1: src = h->naf.tap_y + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 2: src = h->naf.tap_V + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 3: src = h->naf.tap_g ? h->naf.tap_g + (size_t)agent * h->naf.d.Ppad : nullptr; want = P; break;
        default: break;
    }
    RLC_REQUIRE(src, "tap %d not available (gradient taps need rlc_naf_enable_grad_taps)", which);
    RLC_REQUIRE(n == want, "tap %d holds %lld floats, caller asked for %lld", which, want, (long long)n);
    if (which == 3) return naf_fetch_blob(h, src, dst);
    RLC_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

}  // extern "C"
