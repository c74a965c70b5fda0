// rlc_api.hip -- the C ABI of librlcontrol_hip.so (declared in include/rlcontrol_hip.h).
// Host-side handle management; all compute is in the kernel translation units.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "rlc_handle.h"

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void rlc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}



namespace {

template <typename T>
int dmalloc(rlc_handle* h, T** out, size_t count, bool zero = true) { return rlc_h_malloc(h, out, count, zero); }

int ensure_io_impl(rlc_handle* h, size_t bytes) {
    // every CPU write into io_host goes through here first: wait for the copies a previous update_batch queued out of it
    if (h->io_pending) {
        RLC_HIP(hipStreamSynchronize(h->st));
        h->io_pending = false;
    }
    if (bytes > h->io_cap) {
        RLC_HIP(hipStreamSynchronize(h->st));
        if (h->io_dev) RLC_HIP(hipFree(h->io_dev));
        h->io_dev = nullptr;
        h->io_cap = 0;
        const size_t cap = bytes * 2;
        RLC_HIP(hipMalloc((void**)&h->io_dev, cap));
        h->io_cap = cap;
    }
    if (bytes > h->io_host_cap) {
        RLC_HIP(hipStreamSynchronize(h->st));
        if (h->io_host) RLC_HIP(hipHostFree(h->io_host));
        h->io_host = nullptr;
        h->io_host_cap = 0;
        const size_t cap = bytes * 2;
        RLC_HIP(hipHostMalloc(&h->io_host, cap, hipHostMallocDefault));
        h->io_host_cap = cap;
    }
    return 0;
}

int check_agent(const rlc_handle* h, int agent) {
    RLC_REQUIRE(h != nullptr, "null handle");
    RLC_REQUIRE(agent >= 0 && agent < h->rep.n_agents, "agent %d out of range [0,%d)", agent, h->rep.n_agents);
    return 0;
}
int ensure_io(rlc_handle* h, size_t bytes) { return ensure_io_impl(h, bytes); }

float* blob_ptr(rlc_handle* h, int which) {
    switch (which) {
        case 0: return h->dv.theta;
        case 1: return h->dv.theta_t;
        case 2: return h->dv.m_a;
        case 3: return h->dv.v_a;
        case 4: return h->dv.m_c;
        case 5: return h->dv.v_c;
        default: return nullptr;
    }
}

int use_device(const rlc_handle* h) {
    RLC_HIP(hipSetDevice(h->device));
    return 0;
}

}  // namespace

extern "C" {

const char* rlc_last_error(void) { return g_err; }

int rlc_version(void) { return 100; }

int rlc_device_count(int* out_count) {
    RLC_REQUIRE(out_count != nullptr, "null out_count");
    int n = 0;
    RLC_HIP(hipGetDeviceCount(&n));
    *out_count = n;
    return 0;
}

}  // extern "C"

// ---- shared construction / destruction (C++ linkage; used by the per-algorithm ABI files) -----
int rlc_h_check_agent(const rlc_handle* h, int agent) { return check_agent(h, agent); }
int rlc_h_use_device(const rlc_handle* h) { return use_device(h); }
int rlc_h_ensure_io(rlc_handle* h, size_t bytes) { return ensure_io(h, bytes); }

// ---- the queued acting forward of a drop-in agent (rlc_*_act_queue / rlc_*_act_fetch) ----------------------------
// Zero-copy: the acting kernel reads its inputs from, and stores its outputs (and, for one agent, a completion word) into,
// pinned host memory -- no copy operations on the stream, one kernel launch behind the update, one wait per step.
// rlc_h_aq_begin: `floats` of pinned staging (inputs then outputs) are available in h->aq_host and no earlier queued forward
// still uses them; flagged = the launch reports through the completion word (one-workgroup launches only).
int rlc_h_aq_begin(rlc_handle* h, size_t floats, bool flagged) {
    const size_t need = sizeof(float) * (floats + 4);
    if (need > h->aq_cap) {
        RLC_HIP(hipStreamSynchronize(h->st));
        if (h->aq_host) RLC_HIP(hipHostFree(h->aq_host));
        h->aq_host = nullptr; h->aq_cap = 0;
        RLC_HIP(hipHostMalloc((void**)&h->aq_host, need * 2, hipHostMallocDefault));
        h->aq_cap = need * 2;
        memset(h->aq_host, 0, h->aq_cap);
    } else if (h->aq_n) {
        RLC_HIP(hipStreamSynchronize(h->st));     // a queued forward nobody fetched still reads / writes the buffer
    }
    h->aq_n = 0;
    h->aq_flagged = flagged;
    if (flagged) h->aq_seq += 1;
    return 0;
}

// the completion word (null when the queued launch is not flagged); the kernel stores h->aq_seq into it
int* rlc_h_aq_flag(rlc_handle* h) {
    return h->aq_flagged ? (int*)(h->aq_host + h->aq_cap / sizeof(float) - 1) : nullptr;
}

// wait for the forward queued for agents [first_agent, first_agent + n); its outputs are then in h->aq_host
int rlc_h_aq_wait(rlc_handle* h, int first_agent, int n) {
    RLC_REQUIRE(h->aq_n > 0, "no acting forward is queued (rlc_*_act_queue)");
    RLC_REQUIRE(first_agent == h->aq_first && n == h->aq_n, "queued forward is for agents [%d,%d), asked for [%d,%d)",
                h->aq_first, h->aq_first + h->aq_n, first_agent, first_agent + n);
    bool done = false;
    if (h->aq_flagged) {
        // spin on the completion word the kernel stores after its outputs (microseconds after the kernel ends; a stream
        // synchronisation wakes up tens of microseconds later); bounded, then the ordinary wait
        volatile int* flag = (volatile int*)(h->aq_host + h->aq_cap / sizeof(float) - 1);
        for (long spin = 0; spin < 2000000L; spin++) {      // ~20 ms; a non-coherent mapping never shows the word: fall back
            if (*flag == h->aq_seq) { done = true; break; }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    if (!done) RLC_HIP(hipStreamSynchronize(h->st));
    h->aq_n = 0;
    return 0;
}

int rlc_h_ensure_idx(rlc_handle* h, size_t count) {
    if (count > h->idx_cap) {
        RLC_HIP(hipStreamSynchronize(h->st));
        if (h->idx_dev) RLC_HIP(hipFree(h->idx_dev));
        h->idx_dev = nullptr; h->idx_cap = 0;
        RLC_HIP(hipMalloc((void**)&h->idx_dev, sizeof(long long) * count * 2));
        h->idx_cap = count * 2;
    }
    return 0;
}

// device check, stream/events, replay ring + staging minibatch + sampler state (common to every algorithm)
int rlc_h_init_common(rlc_handle* h, int algo, int device, int n_agents, int S, int A, int B, long long cap,
                      const uint64_t* seeds) {
    RLC_REQUIRE(n_agents >= 1, "n_agents must be >= 1 (got %d)", n_agents);
    RLC_REQUIRE(S >= 1 && A >= 1, "state_dim/action_dim must be >= 1");
    RLC_REQUIRE(B >= 1 && B <= RLC_MAX_BATCH, "batch_size %d outside [1,%d]", B, RLC_MAX_BATCH);
    RLC_REQUIRE(cap >= 1, "buffer_size must be >= 1");
    RLC_REQUIRE(seeds != nullptr, "null per-agent seed array");
    int ndev = 0;
    RLC_HIP(hipGetDeviceCount(&ndev));
    RLC_REQUIRE(device >= 0 && device < ndev, "device %d not present (%d visible)", device, ndev);
    RLC_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RLC_HIP(hipGetDeviceProperties(&prop, device));
    RLC_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
                "librlcontrol_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    h->algo = algo;
    h->device = device;
    h->B = B;
    h->idx_dev = nullptr; h->idx_cap = 0;
    h->io_dev = nullptr; h->io_cap = 0;
    h->io_host = nullptr; h->io_host_cap = 0;
    h->aq_dev = nullptr; h->aq_host = nullptr; h->aq_cap = 0; h->aq_first = 0; h->aq_n = 0;
    h->aq_seq = 0; h->aq_flagged = false;
    h->idx_pin = nullptr; h->idx_pin_cap = 0; h->idx_ev = nullptr; h->idx_ev_armed = false;
    h->io_pending = false;
    h->variant = 0;
    h->split_c = 1; h->split_part = nullptr; h->split_bar = nullptr; h->split_err = nullptr;
    h->split_poisoned = false; h->split_fail_next = false;
    h->grad_taps = 0;
    h->has_env = false;
    memset(&h->env, 0, sizeof(h->env));
    h->rollout_dev = nullptr;
    h->sac_rollout_dev = nullptr;
    h->naf_rollout_dev = nullptr;
    h->ro_total_limit = h->ro_eval_interval = h->ro_steps = h->ro_evals = 0;
    h->ro_pending_q8 = 0;
    h->st = nullptr;
    memset(&h->rep, 0, sizeof(h->rep));
    memset(&h->dv, 0, sizeof(h->dv));
    memset(&h->sac, 0, sizeof(h->sac));
    memset(&h->naf, 0, sizeof(h->naf));
    RLC_HIP(hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking));
    (void)hipEventCreate(&h->ev0);
    (void)hipEventCreate(&h->ev1);
    RlcReplayDev& rp = h->rep;
    rp.S = S; rp.A = A; rp.n_agents = n_agents; rp.cap = cap;
    const size_t NA = n_agents, c = (size_t)cap;
    // replay SoA (not zeroed: slots are written before they are read)
    if (dmalloc(h, &rp.rs, NA * c * S, false) || dmalloc(h, &rp.rs2, NA * c * S, false) ||
        dmalloc(h, &rp.ra, NA * c * A, false) || dmalloc(h, &rp.rr, NA * c, false) || dmalloc(h, &rp.rg, NA * c, false) ||
        dmalloc(h, &rp.ring, NA) || dmalloc(h, &rp.gs, NA * RLC_MAX_BATCH * S) || dmalloc(h, &rp.gs2, NA * RLC_MAX_BATCH * S) ||
        dmalloc(h, &rp.ga, NA * RLC_MAX_BATCH * A) || dmalloc(h, &rp.gr, NA * RLC_MAX_BATCH) ||
        dmalloc(h, &rp.gg, NA * RLC_MAX_BATCH) || dmalloc(h, &rp.sample_ctr, NA))
        return 1;
    unsigned long long* seed_dev;
    if (dmalloc(h, &seed_dev, NA)) return 1;
    rp.seed = seed_dev;
    RLC_HIP(hipMemcpyAsync(seed_dev, seeds, NA * sizeof(unsigned long long), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    h->ring.assign(NA, RlcRingMeta{0, 0});
    return 0;
}

void rlc_h_destroy(rlc_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->st) (void)hipStreamSynchronize(h->st);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->idx_dev) (void)hipFree(h->idx_dev);
    if (h->io_dev) (void)hipFree(h->io_dev);
    if (h->io_host) (void)hipHostFree(h->io_host);
    if (h->aq_dev) (void)hipFree(h->aq_dev);
    if (h->aq_host) (void)hipHostFree(h->aq_host);
    if (h->idx_pin) (void)hipHostFree(h->idx_pin);
    if (h->idx_ev) (void)hipEventDestroy(h->idx_ev);
    if (h->st) {
        (void)hipEventDestroy(h->ev0);
        (void)hipEventDestroy(h->ev1);
        (void)hipStreamDestroy(h->st);
    }
    delete h;
}

extern "C" {

int rlc_ddpg_create(const rlc_ddpg_config* cfg, rlc_handle** out) {
    RLC_REQUIRE(cfg && out, "null argument");
    RLC_REQUIRE(cfg->shared_l1_dim >= 1 && cfg->actor_l2_dim >= 1 && cfg->critic_l2_dim >= 1,
                "layer widths must be >= 1");
    RLC_REQUIRE(cfg->state_min && cfg->state_max && cfg->action_min && cfg->action_max, "null bounds array");
    RLC_REQUIRE(cfg->actor_lr && cfg->critic_lr, "null per-agent array");
    rlc_handle* h = new rlc_handle();
    int rc = rlc_h_init_common(h, RLC_ALGO_DDPG, cfg->device, cfg->n_agents, cfg->state_dim, cfg->action_dim,
                               cfg->batch_size, cfg->buffer_size, cfg->seed);
    if (rc) { rlc_h_destroy(h); return rc; }

    RlcDev& dv = h->dv;
    // the tile-blocked weight layout goes with the MFMA kernel (the default whenever it supports the shape)
    RLC_REQUIRE(cfg->norm_type == RLC_NORM_NONE || cfg->norm_type == RLC_NORM_LAYER,
                "norm_type %d: the library implements 'none' / 'input_norm' (0) and 'layer' (1); 'batch' (fused batch "
                "norm with moving averages, agents/network/base_network.py:57-59) is not implemented", cfg->norm_type);
    RLC_REQUIRE(cfg->separate_networks == 0 || cfg->separate_networks == 1, "separate_networks must be 0 or 1");
    RLC_REQUIRE(!cfg->norm_type || (cfg->shared_l1_dim <= 1024 && cfg->actor_l2_dim <= 1024 && cfg->critic_l2_dim <= 1024),
                "layer norm supports layer widths up to 1024");
    dv.d = rlc_make_dims(cfg->state_dim, cfg->action_dim, cfg->shared_l1_dim, cfg->actor_l2_dim,
                         cfg->critic_l2_dim, cfg->batch_size, 0, cfg->norm_type, cfg->separate_networks);
    if (rlc_mfma_supported(dv.d))
        dv.d = rlc_make_dims(cfg->state_dim, cfg->action_dim, cfg->shared_l1_dim, cfg->actor_l2_dim,
                             cfg->critic_l2_dim, cfg->batch_size, 1, cfg->norm_type, cfg->separate_networks);
    dv.rep = h->rep;
    dv.n_agents = cfg->n_agents;
    dv.clip_state = cfg->clip_state;
    dv.tau = cfg->tau;
    dv.ou_theta = cfg->ou_theta; dv.ou_mu = cfg->ou_mu; dv.ou_sigma = cfg->ou_sigma;
    const size_t NA = cfg->n_agents, PP = dv.d.Ppad, S = dv.d.S, A = dv.d.A;

#define TRY(x) do { rc = (x); if (rc) { rlc_h_destroy(h); return rc; } } while (0)
    TRY(dmalloc(h, &dv.theta, NA * PP));
    TRY(dmalloc(h, &dv.theta_t, NA * PP));
    TRY(dmalloc(h, &dv.m_a, NA * PP));
    TRY(dmalloc(h, &dv.v_a, NA * PP));
    TRY(dmalloc(h, &dv.m_c, NA * PP));
    TRY(dmalloc(h, &dv.v_c, NA * PP));
    TRY(dmalloc(h, &dv.pw, NA * 4));
    float *lr_a, *lr_c, *smin, *smax, *amin, *amax;
    TRY(dmalloc(h, &lr_a, NA)); TRY(dmalloc(h, &lr_c, NA));
    TRY(dmalloc(h, &smin, S)); TRY(dmalloc(h, &smax, S)); TRY(dmalloc(h, &amin, A)); TRY(dmalloc(h, &amax, A));
    dv.actor_lr = lr_a; dv.critic_lr = lr_c;
    dv.smin = smin; dv.smax = smax; dv.amin = amin; dv.amax = amax;
    TRY(dmalloc(h, &dv.noise_ctr, NA));
    TRY(dmalloc(h, &dv.ou_state, NA * A));
    TRY(dmalloc(h, &dv.tap_q, NA * RLC_MAX_BATCH));
    TRY(dmalloc(h, &dv.tap_y, NA * RLC_MAX_BATCH));
    TRY(dmalloc(h, &dv.tap_aout, NA * RLC_MAX_BATCH * A));
    TRY(dmalloc(h, &dv.tap_dqda, NA * RLC_MAX_BATCH * A));
    dv.tap_gc = nullptr; dv.tap_ga = nullptr;
    dv.scratch_stride = (long long)((rlc_generic_scratch_floats(dv.d) + 63) & ~(size_t)63);
    TRY(dmalloc(h, &dv.scratch, NA * (size_t)dv.scratch_stride, false));
#undef TRY

    // constants
    std::vector<float> pw(NA * 4);
    for (size_t i = 0; i < NA; i++) { pw[4 * i] = 0.9f; pw[4 * i + 1] = 0.999f; pw[4 * i + 2] = 0.9f; pw[4 * i + 3] = 0.999f; }
    std::vector<float> ou(NA * A, cfg->ou_mu);
    hipError_t e = hipSuccess;
    auto up = [&](void* dst, const void* src, size_t bytes) {
        if (e == hipSuccess) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->st);
    };
    up(dv.pw, pw.data(), NA * 4 * sizeof(float));
    up(dv.ou_state, ou.data(), NA * A * sizeof(float));
    up(lr_a, cfg->actor_lr, NA * sizeof(float));
    up(lr_c, cfg->critic_lr, NA * sizeof(float));
    up(smin, cfg->state_min, S * sizeof(float));
    up(smax, cfg->state_max, S * sizeof(float));
    up(amin, cfg->action_min, A * sizeof(float));
    up(amax, cfg->action_max, A * sizeof(float));
    if (e == hipSuccess) e = hipStreamSynchronize(h->st);
    if (e != hipSuccess) {
        rlc_set_error("rlc_ddpg_create: upload failed: %s", hipGetErrorString(e));
        rlc_h_destroy(h);
        return 1;
    }
    *out = h;
    return 0;
}

int rlc_destroy(rlc_handle* h) {
    rlc_h_destroy(h);
    return 0;
}

#define RLC_NEED_DDPG(h) RLC_REQUIRE((h) && (h)->algo == RLC_ALGO_DDPG, "handle is not a DDPG population")

int rlc_ddpg_param_count(const rlc_handle* h, int64_t* out_p) {
    RLC_REQUIRE(h && out_p, "null argument");
    RLC_NEED_DDPG(h);
    *out_p = h->dv.d.P;
    return 0;
}

int rlc_sync(rlc_handle* h) {
    RLC_REQUIRE(h, "null handle");
    if (use_device(h)) return 1;
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_ddpg_set_blob(rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    float* base = blob_ptr(h, which);
    RLC_REQUIRE(base && src, "bad blob selector %d or null src", which);
    const RlcDims& d = h->dv.d;
    RLC_REQUIRE(n == d.P, "blob length %lld != parameter count %d", (long long)n, d.P);
    std::vector<float> padded(d.Ppad, 0.0f);     // compact ABI blob -> device layout
    rlc_pack_blob(d, src, padded.data());
    RLC_HIP(hipMemcpyAsync(base + (size_t)agent * d.Ppad, padded.data(), sizeof(float) * d.Ppad,
                           hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

static int fetch_blob(rlc_handle* h, const float* dev_src, float* dst) {
    const RlcDims& d = h->dv.d;
    std::vector<float> padded(d.Ppad);
    RLC_HIP(hipMemcpyAsync(padded.data(), dev_src, sizeof(float) * d.Ppad, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    rlc_unpack_blob(d, padded.data(), dst);
    return 0;
}

int rlc_ddpg_get_blob(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    float* base = blob_ptr(h, which);
    RLC_REQUIRE(base && dst, "bad blob selector %d or null dst", which);
    RLC_REQUIRE(n == h->dv.d.P, "blob length %lld != parameter count %d", (long long)n, h->dv.d.P);
    return fetch_blob(h, base + (size_t)agent * h->dv.d.Ppad, dst);
}

int rlc_ddpg_set_beta_powers(rlc_handle* h, int32_t agent, const float* pw4) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(pw4, "null pw4");
    RLC_HIP(hipMemcpyAsync(h->dv.pw + agent * 4, pw4, 4 * sizeof(float), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_ddpg_get_beta_powers(rlc_handle* h, int32_t agent, float* pw4) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(pw4, "null pw4");
    RLC_HIP(hipMemcpyAsync(pw4, h->dv.pw + agent * 4, 4 * sizeof(float), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_ddpg_init_target(rlc_handle* h, int32_t agent) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    const size_t off = (size_t)agent * h->dv.d.Ppad;
    RLC_HIP(hipMemcpyAsync(h->dv.theta_t + off, h->dv.theta + off, h->dv.d.Ppad * sizeof(float),
                           hipMemcpyDeviceToDevice, h->st));
    return 0;
}

// ---------------------------------------------------------------------------------------- replay
int rlc_replay_add(rlc_handle* h, int32_t agent, const double* state, const double* action, double reward,
                   const double* next_state, double transition_gamma) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_REQUIRE(state && action && next_state, "null transition field");
    const int S = h->rep.S, A = h->rep.A;
    if (2 * S + A > RLC_PUT1_MAX_FLOATS)
        return rlc_replay_add_batch(h, agent, 1, state, action, &reward, next_state, &transition_gamma);
    RlcRingMeta& m = h->ring[agent];
    RlcPut1 t;
    for (int i = 0; i < S; i++) { t.sas[i] = (float)state[i]; t.sas[S + i] = (float)next_state[i]; }
    for (int j = 0; j < A; j++) t.sas[2 * S + j] = (float)action[j];
    t.r = reward;
    t.g = transition_gamma;
    // append on the right; when full the oldest (left) is evicted (custom_collections.py:83-101)
    t.slot = (m.start + m.size) % h->rep.cap;
    if (m.size == h->rep.cap) m.start = (m.start + 1) % h->rep.cap;
    else m.size += 1;
    t.new_start = m.start;
    t.new_size = m.size;
    return rlc_launch_replay_put1(h->rep, agent, t, h->st);
}

int rlc_replay_add_batch(rlc_handle* h, int32_t agent, int64_t n, const double* states, const double* actions,
                         const double* rewards, const double* next_states, const double* gammas) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_REQUIRE(n >= 0, "negative count");
    if (n == 0) return 0;
    RLC_REQUIRE(states && actions && rewards && next_states && gammas, "null transition field");
    const size_t S = h->rep.S, A = h->rep.A;
    const long long cap = h->rep.cap;
    // only the newest `cap` of a longer batch can survive FIFO eviction
    long long skip = n > cap ? n - cap : 0;
    const long long m_eff = n - skip;
    const size_t fbytes = sizeof(float) * m_eff * (2 * S + A);
    const size_t dbytes = sizeof(double) * m_eff * 2;
    if (ensure_io(h, fbytes + dbytes + 64)) return 1;
    RLC_HIP(hipStreamSynchronize(h->st));
    double* hd = (double*)h->io_host;                 // doubles first (8-byte alignment)
    float* hf = (float*)(hd + 2 * m_eff);
    for (long long i = 0; i < m_eff; i++) {
        hd[i] = rewards[skip + i];
        hd[m_eff + i] = gammas[skip + i];
    }
    float* hs = hf; float* hs2 = hf + m_eff * S; float* ha = hf + 2 * m_eff * S;
    for (size_t i = 0; i < (size_t)m_eff * S; i++) { hs[i] = (float)states[skip * S + i]; hs2[i] = (float)next_states[skip * S + i]; }
    for (size_t i = 0; i < (size_t)m_eff * A; i++) ha[i] = (float)actions[skip * A + i];
    RLC_HIP(hipMemcpyAsync(h->io_dev, h->io_host, dbytes + fbytes, hipMemcpyHostToDevice, h->st));
    double* dd = (double*)h->io_dev;
    float* df = (float*)(dd + 2 * m_eff);
    RlcRingMeta& m = h->ring[agent];
    // state after `skip` virtual appends followed by m_eff real ones
    long long start = m.start, size = m.size;
    auto advance = [&](long long k) {
        const long long room = cap - size;
        const long long grow = k < room ? k : room;
        size += grow;
        start = (start + (k - grow)) % cap;
    };
    advance(skip);
    const long long first_slot = (start + size) % cap;   // == start when full
    advance(m_eff);
    if (rlc_launch_replay_scatter(h->rep, agent, first_slot, m_eff, df, df + 2 * m_eff * S, dd, df + m_eff * S,
                                  dd + m_eff, h->st))
        return 1;
    m.start = start; m.size = size;
    if (rlc_launch_set_ring(h->rep, agent, start, size, h->st)) return 1;
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_replay_fill_all_dev(rlc_handle* h, int64_t n, const float* s_dev, const float* a_dev, const double* r_dev,
                            const float* s2_dev, const double* g_dev) {
    RLC_REQUIRE(h, "null handle");
    if (use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && n <= h->rep.cap, "fill count %lld outside [1, capacity %lld]", (long long)n, h->rep.cap);
    RLC_REQUIRE(s_dev && a_dev && r_dev && s2_dev && g_dev, "null device array");
    if (rlc_launch_replay_fill_all(h->rep, n, s_dev, a_dev, r_dev, s2_dev, g_dev, h->st)) return 1;
    for (auto& m : h->ring) { m.start = 0; m.size = n; }
    return 0;
}

int rlc_replay_size(const rlc_handle* h, int32_t agent, int64_t* out_size) {
    if (check_agent(h, agent)) return 2;
    RLC_REQUIRE(out_size, "null out_size");
    *out_size = h->ring[agent].size;
    return 0;
}

int rlc_replay_gather(rlc_handle* h, int32_t agent, const int64_t* logical_idx, int32_t k, double* states,
                      double* actions, double* rewards, double* next_states, double* gammas) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_REQUIRE(k >= 0, "negative k");
    if (k == 0) return 0;
    RLC_REQUIRE(logical_idx && states && actions && rewards && next_states && gammas, "null array");
    const long long size = h->ring[agent].size;
    for (int i = 0; i < k; i++)
        RLC_REQUIRE(logical_idx[i] >= 0 && logical_idx[i] < size, "RandomAccessQueue index out of range: %lld (size %lld)",
                    (long long)logical_idx[i], size);
    const size_t S = h->rep.S, A = h->rep.A;
    const size_t ibytes = sizeof(long long) * k, dbytes = sizeof(double) * 2 * k, fbytes = sizeof(float) * k * (2 * S + A);
    if (ensure_io(h, ibytes + dbytes + fbytes)) return 1;
    RLC_HIP(hipStreamSynchronize(h->st));
    long long* di = (long long*)h->io_dev;
    double* dd = (double*)(di + k);
    float* df = (float*)(dd + 2 * k);
    memcpy(h->io_host, logical_idx, ibytes);
    RLC_HIP(hipMemcpyAsync(di, h->io_host, ibytes, hipMemcpyHostToDevice, h->st));
    if (rlc_launch_replay_gather(h->rep, agent, di, k, df, df + 2 * k * S, dd, df + k * S, dd + k, h->st)) return 1;
    RLC_HIP(hipMemcpyAsync(h->io_host, h->io_dev, ibytes + dbytes + fbytes, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    const double* hd = (const double*)((const long long*)h->io_host + k);
    const float* hf = (const float*)(hd + 2 * k);
    for (size_t i = 0; i < (size_t)k * S; i++) { states[i] = hf[i]; next_states[i] = hf[k * S + i]; }
    for (size_t i = 0; i < (size_t)k * A; i++) actions[i] = hf[2 * k * S + i];
    for (int i = 0; i < k; i++) { rewards[i] = hd[i]; gammas[i] = hd[k + i]; }
    return 0;
}

int rlc_replay_sample_indices(rlc_handle* h, int32_t agent, int32_t k, int64_t* out_idx) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    const long long size = h->ring[agent].size;
    // utils/custom_collections.py:110-111
    RLC_REQUIRE(k >= 0 && k <= size, "Sample larger than population or is negative (k=%d, n=%lld)", k, size);
    RLC_REQUIRE(k <= RLC_MAX_BATCH, "k=%d exceeds RLC_MAX_BATCH=%d", k, RLC_MAX_BATCH);
    if (k == 0) return 0;
    RLC_REQUIRE(out_idx, "null out_idx");
    if (ensure_io(h, sizeof(long long) * k)) return 1;
    if (rlc_launch_sample_indices(h->rep, agent, k, (long long*)h->io_dev, h->st)) return 1;
    RLC_HIP(hipMemcpyAsync(out_idx, h->io_dev, sizeof(long long) * k, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

// ---------------------------------------------------------------------------------------- acting
static int act_common(rlc_handle* h, int first_agent, int n, const double* states, float* out_actions, int explore) {
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->dv.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states && out_actions, "null array");
    const size_t S = h->dv.d.S, A = h->dv.d.A;
    const size_t in_b = sizeof(float) * n * S, out_b = sizeof(float) * n * A;
    if (ensure_io(h, in_b + out_b)) return 1;
    float* hin = (float*)h->io_host;
    for (size_t i = 0; i < (size_t)n * S; i++) hin[i] = (float)states[i];
    float* din = h->io_dev;
    float* dout = h->io_dev + n * S;
    RLC_HIP(hipMemcpyAsync(din, hin, in_b, hipMemcpyHostToDevice, h->st));
    if (rlc_launch_act(h->dv, first_agent, n, din, dout, explore, h->st)) return 1;
    RLC_HIP(hipMemcpyAsync(hin + n * S, dout, out_b, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    memcpy(out_actions, hin + n * S, out_b);
    return 0;
}

int rlc_ddpg_act(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, float* out_actions) {
    return act_common(h, first_agent, n, states, out_actions, 0);
}

int rlc_ddpg_act_explore(rlc_handle* h, int32_t first_agent, int32_t n, const double* states, float* out_actions) {
    return act_common(h, first_agent, n, states, out_actions, 1);
}

// update(..., next_state, ...) is followed by step(next_state) (agents/base_agent.py:54-63, experiment.py:132-135): the
// acting forward for next_state is queued on the handle's stream right behind the update that was just launched, its
// result lands in a pinned buffer, and the host is not synchronised here -- one synchronisation per environment step,
// in rlc_ddpg_act_fetch.
int rlc_ddpg_act_queue(rlc_handle* h, int32_t first_agent, int32_t n, const double* states) {
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->dv.n_agents, "agent range [%d,%d) invalid",
                first_agent, first_agent + n);
    RLC_REQUIRE(states, "null array");
    const size_t S = h->dv.d.S, A = h->dv.d.A;
    if (rlc_h_aq_begin(h, n * (S + A), n == 1)) return 1;
    for (size_t i = 0; i < (size_t)n * S; i++) h->aq_host[i] = (float)states[i];
    if (rlc_launch_act(h->dv, first_agent, n, h->aq_host, h->aq_host + n * S, 0, h->st, rlc_h_aq_flag(h), h->aq_seq))
        return 1;
    h->aq_first = first_agent; h->aq_n = n;
    return 0;
}

int rlc_ddpg_act_fetch(rlc_handle* h, int32_t first_agent, int32_t n, float* out_actions) {
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(out_actions, "null array");
    if (rlc_h_aq_wait(h, first_agent, n)) return 1;
    memcpy(out_actions, h->aq_host + (size_t)n * h->dv.d.S, sizeof(float) * n * h->dv.d.A);
    return 0;
}

int rlc_ddpg_reset_noise(rlc_handle* h, int32_t first_agent, int32_t n) {
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(n >= 1 && first_agent >= 0 && first_agent + n <= h->dv.n_agents, "agent range invalid");
    return rlc_launch_reset_noise(h->dv, first_agent, n, h->st);
}

int rlc_ddpg_qval(rlc_handle* h, int32_t agent, int32_t n, const double* states, const double* actions, float* out_q) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(n >= 1 && states && actions && out_q, "bad arguments");
    const size_t S = h->dv.d.S, A = h->dv.d.A;
    const size_t in_b = sizeof(float) * n * (S + A), out_b = sizeof(float) * n;
    if (ensure_io(h, in_b + out_b)) return 1;
    float* hin = (float*)h->io_host;
    for (size_t i = 0; i < (size_t)n * S; i++) hin[i] = (float)states[i];
    for (size_t i = 0; i < (size_t)n * A; i++) hin[n * S + i] = (float)actions[i];
    RLC_HIP(hipMemcpyAsync(h->io_dev, hin, in_b, hipMemcpyHostToDevice, h->st));
    float* dout = h->io_dev + n * (S + A);
    if (rlc_launch_qval(h->dv, agent, n, h->io_dev, h->io_dev + n * S, dout, h->st)) return 1;
    RLC_HIP(hipMemcpyAsync(hin + n * (S + A), dout, out_b, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    memcpy(out_q, hin + n * (S + A), out_b);
    return 0;
}

// -------------------------------------------------------------------------------------- learning
static int pick_variant(const rlc_handle* h) {
    if (h->variant == 1 || h->variant == 2) return h->variant;
    return rlc_mfma_supported(h->dv.d) ? 2 : 1;
}

// Re-pack the six per-agent blobs when the kernel variant (and with it the weight layout) changes.
static int relayout(rlc_handle* h, int blocked) {
    if (h->dv.d.blocked == blocked) return 0;
    if (use_device(h)) return 1;
    const RlcDims od = h->dv.d;
    const RlcDims nd = rlc_make_dims(od.S, od.A, od.H1, od.HA, od.HC, od.B, blocked, od.norm, od.sep);
    const size_t NA = h->dv.n_agents, PP = od.Ppad;
    std::vector<float> dev(NA * PP), compact(od.P), out(NA * PP);
    for (int which = 0; which < 6; which++) {
        float* base = blob_ptr(h, which);
        RLC_HIP(hipMemcpyAsync(dev.data(), base, sizeof(float) * NA * PP, hipMemcpyDeviceToHost, h->st));
        RLC_HIP(hipStreamSynchronize(h->st));
        std::fill(out.begin(), out.end(), 0.0f);
        for (size_t a = 0; a < NA; a++) {
            rlc_unpack_blob(od, &dev[a * PP], compact.data());
            rlc_pack_blob(nd, compact.data(), &out[a * PP]);
        }
        RLC_HIP(hipMemcpyAsync(base, out.data(), sizeof(float) * NA * PP, hipMemcpyHostToDevice, h->st));
        RLC_HIP(hipStreamSynchronize(h->st));
    }
    h->dv.d = nd;
    return 0;
}

int rlc_h_split_before_launch(rlc_handle* h) {
    RLC_REQUIRE(!h->split_poisoned, "latency mode: an earlier update of this handle failed at a cross-workgroup barrier; its "
                "parameters are those of the last completed phase of that update -- reload them and call set_split again");
    const int word = h->split_fail_next ? 1 : 0;
    h->split_fail_next = false;
    RLC_HIP(hipMemcpyAsync(h->split_err, &word, sizeof(int), hipMemcpyHostToDevice, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));          // `word` is a stack variable
    return 0;
}

int rlc_h_split_after_launch(rlc_handle* h) {
    // a barrier that did not complete (a peer workgroup was not resident) must not pass for a finished update
    int err = 0;
    RLC_HIP(hipMemcpyAsync(&err, h->split_err, sizeof(int), hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    if (err != 0) h->split_poisoned = true;
    RLC_REQUIRE(err == 0, "split update: a cross-workgroup barrier did not complete (the GPU is shared with other work?); "
                "every workgroup stopped at it, the update is incomplete and the handle refuses further latency-mode updates");
    return 0;
}

int rlc_debug_fail_next_split(rlc_handle* h) {
    RLC_REQUIRE(h != nullptr && h->split_err != nullptr, "latency mode is not armed on this handle (set_split)");
    h->split_fail_next = true;
    return 0;
}

static int launch_update(rlc_handle* h, int first, int n, int n_updates, int source, const long long* idx_dev) {
    const int v = pick_variant(h);
    if (v == 2 && h->split_c > 1) {
        if (rlc_h_split_before_launch(h)) return 1;
        if (rlc_launch_ddpg_update_split(h->dv, h->split_part, h->split_bar, h->split_err, h->split_c, first, n, n_updates,
                                         source, idx_dev, h->grad_taps, h->st))
            return 1;
        return rlc_h_split_after_launch(h);
    }
    if (v == 2) {
        RLC_REQUIRE(rlc_mfma_supported(h->dv.d), "MFMA kernel does not support these dimensions");
        return rlc_launch_ddpg_update_mfma(h->dv, first, n, n_updates, source, idx_dev, h->grad_taps, h->st);
    }
    return rlc_launch_ddpg_update_generic(h->dv, first, n, n_updates, source, idx_dev, h->grad_taps, h->st);
}

int rlc_ddpg_update(rlc_handle* h, int32_t n_updates, const int64_t* host_indices) {
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(n_updates >= 0, "negative n_updates");
    if (n_updates == 0) return 0;
    const int B = h->dv.d.B, NA = h->dv.n_agents;
    for (int a = 0; a < NA; a++)   // utils/replaybuffer.py:34
        RLC_REQUIRE(h->ring[a].size >= B, "agent %d: replay holds %lld transitions < batch_size %d", a,
                    h->ring[a].size, B);
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
        source = RLC_SRC_REPLAY_HOST_INDICES;
        if (count <= 1024) {
            // the step loop of a drop-in agent (one agent, one update): the kernel reads the indices straight from pinned
            // host memory -- no copy operation on the stream; an event guards the buffer against the next call
            if (!h->idx_pin) {
                RLC_HIP(hipHostMalloc((void**)&h->idx_pin, sizeof(long long) * 1024, hipHostMallocDefault));
                h->idx_pin_cap = 1024;
                RLC_HIP(hipEventCreateWithFlags(&h->idx_ev, hipEventDisableTiming));
            }
            if (h->idx_ev_armed) RLC_HIP(hipEventSynchronize(h->idx_ev));
            memcpy(h->idx_pin, host_indices, sizeof(long long) * count);
            const int rc = launch_update(h, 0, NA, n_updates, source, h->idx_pin);
            RLC_HIP(hipEventRecord(h->idx_ev, h->st));
            h->idx_ev_armed = true;
            return rc;
        }
        if (rlc_h_ensure_idx(h, count)) return 1;
        RLC_HIP(hipMemcpyAsync(h->idx_dev, host_indices, sizeof(long long) * count, hipMemcpyHostToDevice, h->st));
    }
    return launch_update(h, 0, NA, n_updates, source, h->idx_dev);
}

int rlc_ddpg_update_batch(rlc_handle* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                          const double* next_states, const double* rewards, const double* gammas) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(batch == h->dv.d.B, "minibatch has %d rows; the handle was created for batch_size %d", batch, h->dv.d.B);
    RLC_REQUIRE(states && actions && next_states && rewards && gammas, "null minibatch array");
    const size_t S = h->dv.d.S, A = h->dv.d.A, B = batch;
    const size_t fbytes = sizeof(float) * B * (2 * S + A), dbytes = sizeof(double) * 2 * B;
    if (ensure_io(h, fbytes + dbytes)) return 1;
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
    return launch_update(h, agent, 1, 1, RLC_SRC_STAGING, nullptr);
}

int rlc_ddpg_set_kernel(rlc_handle* h, int32_t variant) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(variant >= 0 && variant <= 2, "kernel variant must be 0 (auto), 1 (generic) or 2 (mfma)");
    RLC_REQUIRE(variant != 2 || rlc_mfma_supported(h->dv.d), "MFMA kernel does not support these dimensions");
    RLC_REQUIRE(!h->has_env, "the kernel variant cannot change once a rollout is attached to the handle");
    h->variant = variant;
    return relayout(h, pick_variant(h) == 2 ? 1 : 0);
}

int rlc_ddpg_set_split(rlc_handle* h, int32_t n_workgroups) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    RLC_REQUIRE(n_workgroups >= 1 && n_workgroups <= 8, "workgroups per agent must be in [1,8]");
    RLC_REQUIRE(!h->has_env, "the on-device experiment loop runs the one-workgroup kernels");
    if (n_workgroups == 1) { h->split_c = 1; return 0; }
    RLC_REQUIRE(pick_variant(h) == 2, "the split update is a variant of the MFMA kernel (these dimensions run the generic one)");
    RLC_REQUIRE(!h->dv.d.sep, "latency mode is built for the hydra network (network: separate runs the one-workgroup kernels)");
    RLC_REQUIRE(rlc_split_mt(h->dv.d.B, n_workgroups) > 0, "batch_size %d does not fit %d workgroups of at most 64 rows",
                h->dv.d.B, n_workgroups);
    hipDeviceProp_t prop;
    RLC_HIP(hipGetDeviceProperties(&prop, h->device));
    const int grid = (h->dv.n_agents + 7) / 8 * 8 * n_workgroups;
    RLC_REQUIRE(grid <= prop.multiProcessorCount, "%d agents x %d workgroups need %d co-resident workgroups; the GPU has %d CUs",
                h->dv.n_agents, n_workgroups, grid, prop.multiProcessorCount);
    if (!h->split_bar) {
        if (rlc_h_malloc(h, &h->split_bar, (size_t)h->dv.n_agents) || rlc_h_malloc(h, &h->split_err, (size_t)1)) return 1;
    }
    // partial-gradient blobs: zeroed once, the kernels only ever write real parameter slots
    if (rlc_h_malloc(h, &h->split_part, (size_t)h->dv.n_agents * n_workgroups * h->dv.d.Ppad)) return 1;
    h->split_c = n_workgroups;
    h->split_poisoned = false;           // re-armed by the caller
    return 0;
}

int rlc_ddpg_get_kernel(const rlc_handle* h, int32_t* variant_in_use) {
    RLC_REQUIRE(h && variant_in_use, "null argument");
    *variant_in_use = pick_variant(h);
    return 0;
}

int rlc_ddpg_last_tap(rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n) {
    if (check_agent(h, agent) || use_device(h)) return 2;
    RLC_NEED_DDPG(h);
    RLC_REQUIRE(dst, "null dst");
    const int B = h->dv.d.B, A = h->dv.d.A, P = h->dv.d.P;
    const float* src = nullptr;
    long long want = 0;
    switch (which) {
        case 0: src = h->dv.tap_q + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 1: src = h->dv.tap_y + (size_t)agent * RLC_MAX_BATCH; want = B; break;
        case 2: src = h->dv.tap_aout + (size_t)agent * RLC_MAX_BATCH * A; want = (long long)B * A; break;
        case 3: src = h->dv.tap_dqda + (size_t)agent * RLC_MAX_BATCH * A; want = (long long)B * A; break;
        case 4: src = h->dv.tap_gc ? h->dv.tap_gc + (size_t)agent * h->dv.d.Ppad : nullptr; want = P; break;
        case 5: src = h->dv.tap_ga ? h->dv.tap_ga + (size_t)agent * h->dv.d.Ppad : nullptr; want = P; break;
        default: break;
    }
    RLC_REQUIRE(src, "tap %d not available (gradient taps need rlc_ddpg_enable_grad_taps)", which);
    RLC_REQUIRE(n == want, "tap %d holds %lld floats, caller asked for %lld", which, want, (long long)n);
    if (which >= 4) return fetch_blob(h, src, dst);     // gradient blobs use the padded device layout
    RLC_HIP(hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost, h->st));
    RLC_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int rlc_ddpg_enable_grad_taps(rlc_handle* h, int32_t on) {
    RLC_REQUIRE(h, "null handle");
    RLC_NEED_DDPG(h);
    if (use_device(h)) return 1;
    if (on && !h->dv.tap_gc) {
        const size_t n = (size_t)h->dv.n_agents * h->dv.d.Ppad;
        if (dmalloc(h, &h->dv.tap_gc, n)) return 1;
        if (dmalloc(h, &h->dv.tap_ga, n)) return 1;
    }
    h->grad_taps = on ? 1 : 0;
    return 0;
}

int rlc_timer_begin(rlc_handle* h) {
    RLC_REQUIRE(h, "null handle");
    if (use_device(h)) return 1;
    RLC_HIP(hipEventRecord(h->ev0, h->st));
    return 0;
}

int rlc_timer_end(rlc_handle* h, float* out_ms) {
    RLC_REQUIRE(h && out_ms, "null argument");
    if (use_device(h)) return 1;
    RLC_HIP(hipEventRecord(h->ev1, h->st));
    RLC_HIP(hipEventSynchronize(h->ev1));
    RLC_HIP(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
    return 0;
}

}  // extern "C"
