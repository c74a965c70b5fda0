// rlc_handle.h -- the one opaque handle type behind rlc_ddpg / rlc_sac / rlc_naf (include/rlcontrol_hip.h):
// a population of independent agents of ONE algorithm on one GPU.  The replay ring, the staging buffers,
// the stream and the timer are common; the networks and optimizer state are per algorithm.
#pragma once
#include "rlc_common.h"
#include "sac_common.h"
#include "naf_common.h"

enum RlcAlgo { RLC_ALGO_DDPG = 1, RLC_ALGO_SAC = 2, RLC_ALGO_NAF = 3, RLC_ALGO_KL = 4 };   // KL: ReverseKL / ForwardKL, on RlcSacDev

struct rlc_handle {
    int algo;
    int device;
    int B;                               // batch_size
    hipStream_t st;
    hipEvent_t ev0, ev1;
    RlcReplayDev rep;                    // device view of the replay (also copied into the per-algorithm views)
    std::vector<RlcRingMeta> ring;       // host mirror of the ring metadata
    std::vector<void*> allocs;           // every hipMalloc of this handle
    long long* idx_dev; size_t idx_cap;  // host-index upload buffer
    float* io_dev; size_t io_cap;        // act / qval / gather staging (device, bytes)
    void* io_host; size_t io_host_cap;   // pinned host staging (bytes)
    bool io_pending;                     // async copies out of io_host may still be in flight (the update_batch paths)
    // queued acting forward (rlc_ddpg_act_queue / rlc_ddpg_act_fetch): buffers of their own, nothing else writes them
    float* aq_dev; float* aq_host; size_t aq_cap;   // device / pinned host staging: [n][S] states then [n][A] actions
    int aq_first, aq_n;                  // agent range of the queued forward; aq_n == 0: nothing queued
    int aq_seq;                          // completion flag protocol of one-agent forwards: the kernel stores ++aq_seq
    size_t aq_out;                       // float offset of the queued forward's outputs in aq_host
    bool aq_flagged;                     // the queued forward reports through the flag word (aq_host[aq_cap / 4 - 1])
    long long* idx_pin; size_t idx_pin_cap;   // pinned, device-readable index staging for small host-index updates
    hipEvent_t idx_ev; bool idx_ev_armed;     // recorded behind the launch that reads idx_pin
    // ---- DDPG
    RlcDev dv;
    int variant;                         // requested kernel: 0 auto, 1 generic, 2 mfma
    int grad_taps;
    int split_c;                         // > 1: latency mode, one agent's minibatch over split_c workgroups (ddpg_split.hip)
    float* split_part; unsigned int* split_bar; int* split_err;
    bool split_poisoned;                 // a latency-mode update failed at a cross-workgroup barrier: refuse further ones
    bool split_fail_next;                // test hook (rlc_debug_fail_next_split): the next launch finds the error word set
    // ---- SAC
    RlcSacDev sac;
    // ---- NAF
    RlcNafDev naf;
    // ---- on-device experiment loop (rlc_api_rollout.hip)
    bool has_env;
    RlcEnvDev env;
    RlcRollout* rollout_dev;             // device copy of {dv, env}: argument block of the fused step launches
    struct RlcSacRollout* sac_rollout_dev;   // same for a SoftActorCritic population
    struct RlcNafRollout* naf_rollout_dev;   // same for a NAF population
    long long ro_total_limit, ro_eval_interval, ro_steps, ro_evals;
    int ro_pending_q8;                   // an evaluation ran after the last update: next step resets OU after acting
};

// kernel variant in use (1 generic, 2 mfma): the request h->variant (0 auto) resolved against the shape support
// Latency-mode launches (ddpg_split.hip / kl_mfma.hip): the error word is cleared before every launch (set instead when
// the test hook asked for a failure) and read back after it.  On failure every workgroup has left the kernel at the
// barrier that failed, before any store of the phase behind it: parameters / optimizer state are those of the last
// COMPLETED phase of the failed update.  The handle then refuses latency-mode updates until rlc_*_set_split re-arms it
// (after the caller has reloaded or accepted the state).
extern "C" {
int rlc_h_split_before_launch(rlc_handle* h);
int rlc_h_split_after_launch(rlc_handle* h);
}

inline int rlc_h_sac_variant(const rlc_handle* h) {
    if (h->variant == 1 || h->variant == 2) return h->variant;
    return rlc_sac_mfma_supported(h->sac.d) ? 2 : 1;
}
inline int rlc_h_kl_variant(const rlc_handle* h) {
    if (h->variant == 1 || h->variant == 2) return h->variant;
    return rlc_kl_mfma_supported(h->sac.d, h->sac.kl_nodes) ? 2 : 1;
}
inline int rlc_h_naf_variant(const rlc_handle* h) {
    if (h->variant == 1 || h->variant == 2) return h->variant;
    return rlc_naf_mfma_supported(h->naf.d) ? 2 : 1;
}
// fused update launch of the variant in use (rlc_api_sac.hip / rlc_api_naf.hip)
int rlc_h_sac_launch_update(rlc_handle* h, int first, int n, int n_updates, int source, const long long* idx_dev,
                            const float* eps_dev, const struct RlcSacRollout* rollout);
int rlc_h_naf_launch_update(rlc_handle* h, int first, int n, int n_updates, int source, const long long* idx_dev,
                            const struct RlcNafRollout* rollout);

// re-pack the four per-agent blobs of an RlcSacDev handle between the row-major and the tile-blocked layout
int rlc_h_sac_relayout(rlc_handle* h, int blocked);
// bodies shared by the rlc_sac_* and rlc_kl_* entry points (rlc_api_sac.hip); algo = RLC_ALGO_SAC or RLC_ALGO_KL
extern "C" {
int rlc_sacfam_param_count(int algo, const rlc_handle* h, int64_t* out_p);
int rlc_sacfam_set_blob(int algo, rlc_handle* h, int32_t agent, int32_t which, const float* src, int64_t n);
int rlc_sacfam_get_blob(int algo, rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_sacfam_init_target(int algo, rlc_handle* h, int32_t agent);
int rlc_sacfam_act(int algo, rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample,
                   const float* eps, float* out_actions);
int rlc_sacfam_act_queue(int algo, rlc_handle* h, int32_t first_agent, int32_t n, const double* states, int32_t sample,
                         const float* eps);
int rlc_sacfam_act_fetch(int algo, rlc_handle* h, int32_t first_agent, int32_t n, float* out_actions);
int rlc_sacfam_update(int algo, rlc_handle* h, int32_t n_updates, const int64_t* host_indices, const float* eps);
int rlc_sacfam_update_batch(int algo, rlc_handle* h, int32_t agent, int32_t batch, const double* states,
                            const double* actions, const double* next_states, const double* rewards,
                            const double* gammas, const float* eps);
int rlc_sacfam_enable_grad_taps(int algo, rlc_handle* h, int32_t on);
int rlc_sacfam_last_tap(int algo, rlc_handle* h, int32_t agent, int32_t which, float* dst, int64_t n);
}

// shared helpers (rlc_api.hip)
int rlc_h_check_agent(const rlc_handle* h, int agent);
int rlc_h_use_device(const rlc_handle* h);
// queued acting forward (rlc_api.hip): pinned staging + completion word
int rlc_h_aq_begin(rlc_handle* h, size_t floats, bool flagged);
int* rlc_h_aq_flag(rlc_handle* h);
int rlc_h_aq_wait(rlc_handle* h, int first_agent, int n);
int rlc_h_ensure_io(rlc_handle* h, size_t bytes);
int rlc_h_ensure_idx(rlc_handle* h, size_t count);
int rlc_h_init_common(rlc_handle* h, int algo, int device, int n_agents, int S, int A, int B, long long cap,
                      const uint64_t* seeds);
void rlc_h_destroy(rlc_handle* h);
template <typename T>
int rlc_h_malloc(rlc_handle* h, T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    const size_t bytes = (count ? count : 1) * sizeof(T);
    RLC_HIP(hipMalloc(&p, bytes));
    if (zero) RLC_HIP(hipMemsetAsync(p, 0, bytes, h->st));
    h->allocs.push_back(p);
    *out = (T*)p;
    return 0;
}
