/*
 * rlcontrol_hip.h -- C ABI of librlcontrol_hip.so: the MI355X (gfx950) replacement for the
 * replay-sampling + actor-critic-update hot path of samuelfneumann/RLControl.
 *
 * The reference has no FFI; its boundary is Python duck typing.  Every entry
 * point below names the reference interface it replaces (file:line under /root/reference) -- these
 * are the calls a maintainer's ctypes stub binds (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - plain C: opaque handle, pointers + sizes, no C++/torch types.
 *   - every function returns 0 on success, non-zero on failure; rlc_last_error() gives the message
 *     (the reference raises Python exceptions: AssertionError utils/replaybuffer.py:34,
 *      ValueError utils/custom_collections.py:110, NotImplementedError agents/base_agent.py:46).
 *   - a handle is a POPULATION of n_agents independent DDPG agents (seeds / sweep settings --
 *     the reference's INDEX axis, main.py:111-141) resident on ONE GPU; agent == 0 with
 *     n_agents == 1 is the reference's one-agent-per-process case.
 *   - not thread-safe per handle; one HIP stream per handle; all device state owned by the library.
 *   - host pointers unless the name ends in _dev.
 *
 * Parameter blob ("theta", P floats; P from rlc_ddpg_param_count), variable creation order of
 * agents/network/hydra_ddpg_network.py:100-140, weights W[in][out] row-major:
 *   W1[S][H1] b1[H1] | Wa2[H1][HA] ba2[HA] Wa3[HA][A] ba3[A] | Wc2[H1+A][HC] bc2[HC] Wc3[HC] bc3[1]
 * Adam slots are blob-aligned arrays of P floats (unused positions stay 0).
 */
#ifndef RLCONTROL_HIP_H
#define RLCONTROL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One opaque handle type: a population of independent agents of ONE algorithm on one GPU.
 * rlc_ddpg / rlc_sac name the algorithm a handle was created for; the replay, timing, sync and destroy entry
 * points take any of them. */
typedef struct rlc_handle rlc_handle;
typedef rlc_handle rlc_ddpg;
typedef rlc_handle rlc_sac;
typedef rlc_handle rlc_naf;

/* Mirrors what DDPG_Network_Manager.__init__ / BaseNetwork.__init__ read from Config
 * (agents/DDPG.py:17-32, agents/network/base_network.py:14-28, hydra_ddpg_network.py:9-17,
 *  agents/base_agent.py:22-25, utils/config.py:8-21). */
typedef struct rlc_ddpg_config {
    int32_t device;          /* HIP device ordinal (one process per GPU: LOCAL_RANK) */
    int32_t n_agents;        /* independent agents resident on this GPU (>= 1) */
    int32_t state_dim;       /* config.state_dim */
    int32_t action_dim;      /* config.action_dim */
    int32_t shared_l1_dim;   /* config.shared_l1_dim  (jsonfiles/agent/ddpg.json:8) */
    int32_t actor_l2_dim;    /* config.actor_l2_dim */
    int32_t critic_l2_dim;   /* config.critic_l2_dim */
    int32_t batch_size;      /* config.batch_size (reference default 32, utils/config.py:12) */
    int64_t buffer_size;     /* config.buffer_size: replay capacity PER AGENT (utils/config.py:13) */
    int32_t clip_state;      /* 1 when config.norm_type != 'none' (hydra_ddpg_network.py:86-87, quirk Q6) */
    int32_t norm_type;       /* RLC_NORM_NONE: config.norm_type 'none' / 'input_norm' (activation only);
                              * RLC_NORM_LAYER: 'layer' -- tf.contrib.layers.layer_norm(center, scale) before every
                              * hidden relu (agents/network/base_network.py:53-56); each layer adds beta then gamma to
                              * the blob.  'batch' (base_network.py:57-59) is not implemented: create fails. */
    float tau;               /* config.tau */
    float reserved1;
    const float* state_min;  /* [state_dim] */
    const float* state_max;  /* [state_dim] */
    const float* action_min; /* [action_dim] (device OU clip) */
    const float* action_max; /* [action_dim] (tanh scale, hydra_ddpg_network.py:92) */
    const float* actor_lr;   /* [n_agents] per-agent (sweep settings differ in lr) */
    const float* critic_lr;  /* [n_agents] */
    const uint64_t* seed;    /* [n_agents] Philox keys of the device sampler / OU generator */
    float ou_theta, ou_mu, ou_sigma; /* utils/config.py:19-21 (device OU generator) */
    int32_t separate_networks; /* 0: the hydra network agents/DDPG.py builds (shared first layer, :26);
                                * 1: separate actor / critic networks (agents/network/actor_network.py:73-96,
                                *    critic_network.py:77-99; commented out in agents/DDPG.py:8-9,24-25): the critic
                                *    gets a first layer of its own.  Blob: W1 b1 [l1b l1g] Wa2 ba2 [l2b l2g] Wa3 ba3
                                *    [Wc1 bc1 [lcb lcg]] Wc2 bc2 [l3b l3g] Wc3 bc3 */
} rlc_ddpg_config;

#define RLC_NORM_NONE 0
#define RLC_NORM_LAYER 1

const char* rlc_last_error(void);
int rlc_version(void);
int rlc_device_count(int* out_count);

/* -- lifetime: DDPG_Network_Manager.__init__ (agents/DDPG.py:17-32) + ReplayBuffer.__init__
 *    (utils/replaybuffer.py:16-23).  Networks start at zero; load them with rlc_ddpg_set_params. */
int rlc_ddpg_create(const rlc_ddpg_config* cfg, rlc_ddpg** out);
int rlc_destroy(rlc_handle* h);
int rlc_ddpg_param_count(const rlc_ddpg* h, int64_t* out_p);
int rlc_sync(rlc_handle* h);                         /* hipStreamSynchronize on the handle's stream */

/* -- parameters / optimizer state (parity taps; also checkpointing).
 *    which: 0 online theta, 1 target theta', 2 actor-Adam m, 3 actor-Adam v, 4 critic-Adam m, 5 critic-Adam v */
int rlc_ddpg_set_blob(rlc_ddpg* h, int32_t agent, int32_t which, const float* src, int64_t n);
int rlc_ddpg_get_blob(rlc_ddpg* h, int32_t agent, int32_t which, float* dst, int64_t n);
/* beta powers {actor b1^t, actor b2^t, critic b1^t, critic b2^t} (TF Adam accumulators, quirk Q2) */
int rlc_ddpg_set_beta_powers(rlc_ddpg* h, int32_t agent, const float* pw4);
int rlc_ddpg_get_beta_powers(rlc_ddpg* h, int32_t agent, float* pw4);
/* theta' <- theta : HydraDDPGNetwork.init_target_network (hydra_ddpg_network.py:32,223-224) */
int rlc_ddpg_init_target(rlc_ddpg* h, int32_t agent);

/* -- replay: ReplayBuffer.add / get_size (utils/replaybuffer.py:25-30), FIFO eviction of the oldest
 *    (utils/custom_collections.py:83-101).  Logical index 0 is the OLDEST stored transition. */
int rlc_replay_add(rlc_handle* h, int32_t agent, const double* state, const double* action, double reward,
                   const double* next_state, double transition_gamma);
int rlc_replay_add_batch(rlc_handle* h, int32_t agent, int64_t n, const double* states, const double* actions,
                         const double* rewards, const double* next_states, const double* gammas);
/* every agent receives the same n transitions, already resident in HBM as fp32/fp64 SoA (bench path) */
int rlc_replay_fill_all_dev(rlc_handle* h, int64_t n, const float* s_dev, const float* a_dev,
                            const double* r_dev, const float* s2_dev, const double* g_dev);
int rlc_replay_size(const rlc_handle* h, int32_t agent, int64_t* out_size);
/* ReplayBuffer.sample_batch's gather (utils/replaybuffer.py:32-37 -> custom_collections.py:37-58) for
 * caller-chosen logical indices: out arrays [k,S] [k,A] [k] [k,S] [k] float64 like the reference returns. */
int rlc_replay_gather(rlc_handle* h, int32_t agent, const int64_t* logical_idx, int32_t k, double* states,
                      double* actions, double* rewards, double* next_states, double* gammas);
/* RandomAccessQueue.sample_n_k on the device (Philox; k distinct uniform logical indices in [0,size)) */
int rlc_replay_sample_indices(rlc_handle* h, int32_t agent, int32_t k, int64_t* out_idx);

/* -- acting: DDPG_Network_Manager.take_action's greedy part, predict_action on B=1
 *    (agents/DDPG.py:36, hydra_ddpg_network.py:162-171).  states [n][S] for agents first..first+n-1,
 *    out [n][A] = tanh(.)*action_max, fp32 like the Session.run fetch. */
int rlc_ddpg_act(rlc_ddpg* h, int32_t first_agent, int32_t n, const double* states, float* out_actions);
/* The same forward, split in two for the step loop of experiment.py:132-135 -- `agent.update(obs, obs_n, ...)` then
 * `agent.step(obs_n)`: BaseAgent.update already holds next_state (agents/base_agent.py:54-63), so the forward for it is
 * QUEUED behind the update just launched (no host synchronisation) and step() FETCHES the action -- one launch
 * sequence and one synchronisation per environment step.  fetch fails unless the same agent range is queued; anything
 * that changes the weights between the two calls (another update, set_blob) makes the queued result stale: the
 * Python agent then drops it and calls rlc_ddpg_act. */
int rlc_ddpg_act_queue(rlc_ddpg* h, int32_t first_agent, int32_t n, const double* states);
int rlc_ddpg_act_fetch(rlc_ddpg* h, int32_t first_agent, int32_t n, float* out_actions);
/* same + device OU noise and clip (utils/exploration_policy.py:18-21); reset -> noise = mu (:23-24) */
int rlc_ddpg_act_explore(rlc_ddpg* h, int32_t first_agent, int32_t n, const double* states, float* out_actions);
int rlc_ddpg_reset_noise(rlc_ddpg* h, int32_t first_agent, int32_t n);
/* predict_qval (hydra_ddpg_network.py:183-193): q[n] for n (state, action) rows on ONE agent's online net */
int rlc_ddpg_qval(rlc_ddpg* h, int32_t agent, int32_t n, const double* states, const double* actions, float* out_q);

/* -- learning.
 *  rlc_ddpg_update: BaseAgent.learn (agents/base_agent.py:65-70) for EVERY agent of the handle,
 *    n_updates times: sample_batch + update_network (agents/DDPG.py:74-95) fused in one launch.
 *    host_indices: NULL -> device Philox sampler; else int64 [n_agents][n_updates][batch] logical
 *    indices (what the reference's RandomState produced) for exact-minibatch parity.
 *    Fails if any agent holds fewer than batch_size transitions (utils/replaybuffer.py:34).
 *  rlc_ddpg_update_batch: DDPG_Network_Manager.update_network(state, action, next_state, reward, gamma)
 *    (agents/DDPG.py:74) on a caller-supplied minibatch for ONE agent. */
int rlc_ddpg_update(rlc_ddpg* h, int32_t n_updates, const int64_t* host_indices);
int rlc_ddpg_update_batch(rlc_ddpg* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                          const double* next_states, const double* rewards, const double* gammas);
/* kernel selection for A/B tests: 0 auto, 1 generic (any dims), 2 MFMA-tiled (gfx950 fp32 matrix cores) */
int rlc_ddpg_set_kernel(rlc_ddpg* h, int32_t variant);
int rlc_ddpg_get_kernel(const rlc_ddpg* h, int32_t* variant_in_use);
/* latency mode (no reference counterpart): split every agent's minibatch over n_workgroups CUs (1 = off, at most 8;
 * MFMA shapes only; n_agents rounded up to 8, times n_workgroups, must not exceed the CU count, and nothing else may
 * occupy the GPU while an update runs: the workgroups of an agent meet at four barriers per update).  Results equal the
 * one-workgroup kernel up to the summation order over the batch. */
int rlc_ddpg_set_split(rlc_ddpg* h, int32_t n_workgroups);
/* Failure behaviour of latency mode (DDPG and KL): when a cross-workgroup barrier does not complete (a peer workgroup
 * was not resident: the GPU is shared), every workgroup of the launch leaves at that barrier before any store of the
 * phase behind it, the update call fails, and the handle refuses further latency-mode updates until set_split is called
 * again (parameters / optimizer state are those of the last completed phase of the failed update: reload them first).
 * Test hook: the next latency-mode launch of `h` behaves as if its first barrier had failed. */
int rlc_debug_fail_next_split(rlc_handle* h);

/* -- debug taps of the LAST update of one agent (the 1e-5 checks): which: 0 q before the critic step
 *    (train_critic's fetch, hydra_ddpg_network.py:155), 1 TD target y, 2 scaled actor output (DDPG.py:90),
 *    3 dQ/da (DDPG.py:91).  n = batch (0,1) or batch*A (2,3). */
int rlc_ddpg_last_tap(rlc_ddpg* h, int32_t agent, int32_t which, float* dst, int64_t n);
/* which 4 / 5: the critic / actor optimizer's gradient blob (P floats, zeros where the gradient is None:
 * hydra_ddpg_network.py:37,72) of the last update -- written only while enabled (extra HBM stores). */
int rlc_ddpg_enable_grad_taps(rlc_ddpg* h, int32_t on);


/* ================================ SoftActorCritic (SAC-v1) ========================================
 * Mirrors what SoftActorCritic_Network_Manager.__init__ / SoftActorCriticNetwork.__init__ read from Config
 * (agents/SoftActorCritic.py:16-53, agents/network/sac_network.py:10-45; jsonfiles/agent/sac.json).
 * Parameter blob (P floats), variable creation order under 'main' (sac_network.py:152-172):
 *   pi: W1[S][L1a] b1 W2[L1a][L2a] b2 Wm[L2a][A] bm Ws[L2a][A] bs | qf: W1[S][L1c] b1 W2[L1c+A][L2c] b2 W3[L2c] b3 |
 *   vf: W1[S][L1c] b1 W2[L1c][L2c] b2 W3[L2c] b3.      blob selector: 0 theta, 1 target, 2 Adam m, 3 Adam v.
 * With norm_type 'layer' every hidden layer is followed by its layer-norm beta[width] gamma[width]:
 *   pi: W1 b1 beta1 gamma1 W2 b2 beta2 gamma2 Wm bm Ws bs | qf: W1 b1 beta1 gamma1 W2 b2 beta2 gamma2 W3 b3 | vf likewise. */
typedef struct rlc_sac_config {
    int32_t device, n_agents, state_dim, action_dim;
    int32_t actor_l1_dim, actor_l2_dim, critic_l1_dim, critic_l2_dim;   /* jsonfiles/agent/sac.json:9-12 */
    int32_t batch_size;      /* config.batch_size */
    int32_t clip_state;      /* 1 when config.norm_type != 'none' (sac_network.py:207,237; Q never clips: :176) */
    int64_t buffer_size;
    float tau;
    float state_min0, state_max0;   /* the SCALARS state_min[0] / state_max[0] the reference clips every dimension to */
    float action_max0;              /* action_max[0]: scale of mu and pi (sac_network.py:160-161) */
    const float* pi_lr;          /* [n_agents] */
    const float* qf_vf_lr;       /* [n_agents] */
    const float* entropy_scale;  /* [n_agents] */
    const uint64_t* seed;        /* [n_agents] Philox keys (sampler, eps) */
    int32_t norm_type;           /* RLC_NORM_NONE ('none' / 'input_norm': activation only) or RLC_NORM_LAYER ('layer':
                                  * tf.contrib.layers.layer_norm before every hidden relu of pi, qf and vf,
                                  * base_network.py:53-56; every hidden layer then adds beta, gamma behind its bias in the
                                  * blob).  'batch' is not implemented: create fails. */
    int32_t reserved0;
} rlc_sac_config;

int rlc_sac_create(const rlc_sac_config* cfg, rlc_sac** out);
int rlc_sac_param_count(const rlc_sac* h, int64_t* out_p);
int rlc_sac_set_blob(rlc_sac* h, int32_t agent, int32_t which, const float* src, int64_t n);
int rlc_sac_get_blob(rlc_sac* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_sac_set_beta_powers(rlc_sac* h, int32_t agent, const float* pw4);   /* {pi b1^t, pi b2^t, value b1^t, value b2^t} */
int rlc_sac_get_beta_powers(rlc_sac* h, int32_t agent, float* pw4);
int rlc_sac_init_target(rlc_sac* h, int32_t agent);                       /* sac_network.py:75-76,357-358 */
/* predict_action (sample = 0: tanh(mu)*action_max[0], sac_network.py:327-333) / sample_action (sample = 1, :336-343).
 * eps: [n][A] N(0,1) draws standing in for tf.random_normal (:286), or NULL -> device Philox. */
int rlc_sac_act(rlc_sac* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps,
                float* out_actions);
/* the same forward queued behind the update just launched / fetched by step() -- see rlc_ddpg_act_queue
 * (agents/base_agent.py:54-63, experiment.py:132-135).  eps is read at queue time. */
int rlc_sac_act_queue(rlc_sac* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps);
int rlc_sac_act_fetch(rlc_sac* h, int32_t first_agent, int32_t n, float* out_actions);
/* BaseAgent.learn for every agent: sample_batch + update_network + update_target_network
 * (agents/SoftActorCritic.py:113-126).  host_indices as in rlc_ddpg_update; eps [n_agents][n_updates][batch][A] or NULL. */
int rlc_sac_update(rlc_sac* h, int32_t n_updates, const int64_t* host_indices, const float* eps);
/* SoftActorCritic_Network_Manager.update_network on a caller-supplied minibatch (one agent); eps [batch][A] or NULL */
int rlc_sac_update_batch(rlc_sac* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                         const double* next_states, const double* rewards, const double* gammas, const float* eps);
/* taps of the last update: 0 q, 1 v, 2 logp_pi, 3 q_pi (n = batch); 4 {pi_loss, q_loss, v_loss} (n = 3) -- the
 * fetches of train_ops (sac_network.py:135-136); 5 gradient blob (n = P, needs rlc_sac_enable_grad_taps) */
int rlc_sac_last_tap(rlc_sac* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_sac_enable_grad_taps(rlc_sac* h, int32_t on);
/* kernel selection (no reference counterpart: one tf.Graph there): 0 auto, 1 generic fp32 VALU kernel (any shape),
 * 2 MFMA kernel (S <= 8, A <= 2, layer widths multiples of 4 in [16,256], LDS permitting).  Switching re-packs the
 * weights between the row-major and the tile-blocked device layout; results differ only in summation order. */
int rlc_sac_set_kernel(rlc_sac* h, int32_t variant);
int rlc_sac_get_kernel(const rlc_sac* h, int32_t* variant_in_use);

/* ============================== ReverseKL / ForwardKL ==============================================
 * Mirrors what ReverseKL_Network_Manager / ReverseKLNetwork (agents/ReverseKL.py:13-29,
 * agents/network/reversekl_network.py:23-76) and their ForwardKL twins (agents/ForwardKL.py,
 * agents/network/forwardkl_network.py:23-106) read from Config; jsonfiles/agent/reverse_kl.json, forward_kl.json.
 * The reference builds these two agents on torch.nn / torch.optim.Adam; this entry point replaces
 * network.update_network + network.update_target_network (agents/ReverseKL.py:83-93) and sample_action / predict_action.
 * Parameter blob (P floats), weights stored [in][out] (the transpose of nn.Linear.weight), module order
 * pi_net, q_net, v_net (reversekl_network.py:47-50):
 *   pi: W1[S][L1a] b1 W2[L1a][L2a] b2 Wm[L2a][A] bm Ws[L2a][A] bs | q: W1[S+A][L1c] b1 W2[L1c][L2c] b2 W3[L2c] b3 |
 *   v: W1[S][L1c] b1 W2[L1c][L2c] b2 W3[L2c] b3.      blob selector: 0 theta, 1 target (only the v block is live:
 *   target_v_net), 2 Adam exp_avg, 3 Adam exp_avg_sq.
 * action_dim 1 .. 6.  The caller supplies the nodes and weights of the action integral: the Clenshaw-Curtis line rule for
 * one action dimension (reversekl_network.py:64-76), the sparse grid of level l_param above it (:78-108).  Above one
 * dimension the policy is the reference's MultivariateNormal(mean, diag_embed(std)) -- covariance diag(std), i.e. a
 * variance of std per component (:383-389) -- reproduced as written; the MFMA kernel covers action_dim 1 only. */
#define RLC_KL_REVERSE 1
#define RLC_KL_FORWARD 2
#define RLC_KL_OPTIM_INTG 0        /* config.optim_type 'intg'      (reverse: soft RKL; forward: the only one implemented there) */
#define RLC_KL_OPTIM_HARD_INTG 1   /* 'hard_intg'  (reverse only, reversekl_network.py:196-207) */
#define RLC_KL_OPTIM_LL 2          /* 'll'         (reverse only, :167-171) */
#define RLC_KL_OPTIM_HARD_LL 3     /* 'hard_ll'    (reverse only, :173-175) */
#define RLC_KL_Q_NON_SAC 0         /* config.q_update_type 'non_sac': V target (r - alpha*logp) + gamma*V'(s') (:157-158) */
#define RLC_KL_Q_SAC 1             /* 'sac': V target Q(s, a_new) - alpha*logp (:154-155) */
typedef struct rlc_kl_config {
    int32_t device, n_agents, state_dim, action_dim;
    int32_t actor_l1_dim, actor_l2_dim, critic_l1_dim, critic_l2_dim;   /* jsonfiles/agent/reverse_kl.json:8-11 */
    int32_t batch_size;
    int32_t kind;            /* RLC_KL_REVERSE / RLC_KL_FORWARD */
    int32_t optim_type;      /* RLC_KL_OPTIM_* */
    int32_t q_update_type;   /* RLC_KL_Q_* */
    int32_t n_nodes;         /* quadrature nodes kept: N_param - 2 (the end points are cut, reversekl_network.py:70-72) */
    int32_t reserved0;
    int64_t buffer_size;
    float tau;
    float action_max0;               /* action_max[0]: scale of tanh (reversekl_network.py:47) */
    const float* node_actions;       /* [n_nodes][action_dim] self.intgrl_actions, fp32: scheme.points[1:-1] * action_max for
                                      * one action dimension (reversekl_network.py:64-72), the sparse grid of :78-108 above it */
    const float* node_weights;       /* [n_nodes] self.intgrl_weights, fp32 (either sign on the sparse grid) */
    const float* pi_lr;              /* [n_agents] */
    const float* qf_vf_lr;           /* [n_agents] */
    const float* entropy_scale;      /* [n_agents] */
    const uint64_t* seed;            /* [n_agents] Philox keys (sampler, eps) */
} rlc_kl_config;
typedef rlc_handle rlc_kl;

int rlc_kl_create(const rlc_kl_config* cfg, rlc_kl** out);
int rlc_kl_param_count(const rlc_kl* h, int64_t* out_p);
int rlc_kl_set_blob(rlc_kl* h, int32_t agent, int32_t which, const float* src, int64_t n);
int rlc_kl_get_blob(rlc_kl* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_kl_set_step(rlc_kl* h, int32_t agent, int32_t step);   /* state['step'] of the three torch optimizers (equal) */
int rlc_kl_get_step(rlc_kl* h, int32_t agent, int32_t* step);
int rlc_kl_init_target(rlc_kl* h, int32_t agent);              /* reversekl_network.py:52-54 */
/* predict_action (sample = 0: tanh(mean)*action_max[0], :120-128) / sample_action (sample = 1, :111-118).
 * eps: [n][A] N(0,1) draws standing in for normal.sample(), or NULL -> device Philox. */
int rlc_kl_act(rlc_kl* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps,
               float* out_actions);
/* queued / fetched as rlc_ddpg_act_queue / rlc_ddpg_act_fetch */
int rlc_kl_act_queue(rlc_kl* h, int32_t first_agent, int32_t n, const double* states, int32_t sample, const float* eps);
int rlc_kl_act_fetch(rlc_kl* h, int32_t first_agent, int32_t n, float* out_actions);
/* BaseAgent.learn for every agent: sample_batch + update_network + update_target_network.
 * host_indices as in rlc_ddpg_update; eps [n_agents][n_updates][batch][A] or NULL. */
int rlc_kl_update(rlc_kl* h, int32_t n_updates, const int64_t* host_indices, const float* eps);
/* *_Network_Manager.update_network on a caller-supplied minibatch (one agent); eps [batch][A] or NULL */
int rlc_kl_update_batch(rlc_kl* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                        const double* next_states, const double* rewards, const double* gammas, const float* eps);
/* taps of the last update: 0 q_val, 1 v_val, 2 log_prob, 3 new_q_val (n = batch); 4 {policy_loss, q_value_loss,
 * value_loss} (n = 3); 5 gradient blob (n = P, needs rlc_kl_enable_grad_taps); 6 intgrl_q_val (n = batch * n_nodes,
 * the integral updates only) */
int rlc_kl_last_tap(rlc_kl* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_kl_enable_grad_taps(rlc_kl* h, int32_t on);
/* kernel selection, as rlc_sac_set_kernel: 0 auto, 1 generic, 2 MFMA (state_dim <= 7, widths multiples of 4 in
 * [16,256], batch_size <= 32, at most 256 nodes, LDS permitting) */
int rlc_kl_set_kernel(rlc_kl* h, int32_t variant);
int rlc_kl_get_kernel(const rlc_kl* h, int32_t* variant_in_use);
/* latency mode, as rlc_ddpg_set_split: n_workgroups (1..8) CUs per agent.  The forward passes of Q at the (state, node)
 * pairs of the action integral -- two thirds of an update -- are dealt over them; results are bit-identical to the
 * one-workgroup kernel's.  MFMA kernel and the integral updates only; host loop only; the GPU must not be shared. */
int rlc_kl_set_split(rlc_kl* h, int32_t n_workgroups);


/* ===================================== NAF =========================================================
 * Mirrors what NAF_Network_Manager.__init__ / NAF_Network.__init__ read from Config (agents/NAF.py:11-21,
 * agents/network/naf_network.py:6-18; jsonfiles/agent/naf.json).  Parameter blob, variable creation order
 * (naf_network.py:79-107): W1[S][L1] b1 | Wa2[L1][L2] ba2 | Wa3[L2][A] ba3 | Wv2[L1][L2] bv2 | Wv3[L2] bv3 |
 * for c < A: Wd_c[L1] bd_c | for c < A-1: Wn_c[L1][A-1-c] bn_c.   blob selector: 0 theta, 1 target, 2 Adam m, 3 Adam v.
 * With norm_type 'layer' b1, ba2 and bv2 are each followed by their layer-norm beta[width] gamma[width]. */
typedef struct rlc_naf_config {
    int32_t device, n_agents, state_dim, action_dim;   /* action_dim <= 6 */
    int32_t l1_dim, l2_dim;                             /* jsonfiles/agent/naf.json:9-10 */
    int32_t batch_size, clip_state;
    int64_t buffer_size;
    float tau;
    int32_t norm_type;           /* RLC_NORM_NONE ('none' / 'input_norm': activation only) or RLC_NORM_LAYER ('layer':
                                  * tf.contrib.layers.layer_norm before the relu of the trunk and of both branches,
                                  * naf_network.py:83,87,93; each of the three adds beta, gamma behind its bias in the
                                  * blob; any-shape kernel only).  'batch' is not implemented: create fails. */
    const float* state_min;      /* [state_dim] (naf_network.py:73) */
    const float* state_max;
    const float* action_max;     /* [action_dim] scale of tanh (naf_network.py:89) */
    const float* learning_rate;  /* [n_agents] */
    const uint64_t* seed;        /* [n_agents] Philox keys of the device sampler */
    const float* action_min;     /* [action_dim] lower clip of the exploration draw in the on-device loop
                                  * (naf_network.py:176 clips to [action_min, action_max]); NULL: -action_max */
} rlc_naf_config;

int rlc_naf_create(const rlc_naf_config* cfg, rlc_naf** out);
int rlc_naf_param_count(const rlc_naf* h, int64_t* out_p);
int rlc_naf_set_blob(rlc_naf* h, int32_t agent, int32_t which, const float* src, int64_t n);
int rlc_naf_get_blob(rlc_naf* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_naf_get_beta_powers(rlc_naf* h, int32_t agent, float* pw2);
int rlc_naf_init_target(rlc_naf* h, int32_t agent);                       /* naf_network.py:57-58,178-179 */
/* predict_action (naf_network.py:144-149): out_mu [n][A]; out_lcols (may be NULL) [n][A(A+1)/2] = the Lmat_columns
 * fetch of sample_action (:157-158), column c = {exp(clip(diag_c)), below-diagonal entries}: the caller forms
 * noise_scale * pinv(L L^T) and samples on the host exactly as the reference does (:161-174). */
int rlc_naf_act(rlc_naf* h, int32_t first_agent, int32_t n, const double* states, float* out_mu, float* out_lcols);
/* queued / fetched as rlc_ddpg_act_queue / rlc_ddpg_act_fetch (the host draws the exploration sample after the fetch) */
int rlc_naf_act_queue(rlc_naf* h, int32_t first_agent, int32_t n, const double* states);
int rlc_naf_act_fetch(rlc_naf* h, int32_t first_agent, int32_t n, float* out_mu, float* out_lcols);
/* BaseAgent.learn for every agent: sample_batch + NAF_Network_Manager.update_network (agents/NAF.py:69-75) */
int rlc_naf_update(rlc_naf* h, int32_t n_updates, const int64_t* host_indices);
int rlc_naf_update_batch(rlc_naf* h, int32_t agent, int32_t batch, const double* states, const double* actions,
                         const double* next_states, const double* rewards, const double* gammas);
/* taps of the last update: 0 Q(s,a), 1 TD target y, 2 V(s) (n = batch); 3 gradient blob (n = P) */
int rlc_naf_last_tap(rlc_naf* h, int32_t agent, int32_t which, float* dst, int64_t n);
int rlc_naf_enable_grad_taps(rlc_naf* h, int32_t on);
/* kernel selection, as rlc_sac_set_kernel: 0 auto, 1 generic, 2 MFMA (S <= 8, A <= 2, widths multiples of 4 in [16,256]) */
int rlc_naf_set_kernel(rlc_naf* h, int32_t variant);
int rlc_naf_get_kernel(const rlc_naf* h, int32_t* variant_in_use);

/* ---------------------------------------------------------------------------------------------------
 * On-device experiment loop: Experiment.run of the reference
 * (experiment.py:52-217) for every agent of a DDPG population, with the environment simulated on the GPU.
 * Per training step: act (+OU) -> env.step -> BaseAgent.update (store unless truncated; gamma_i = 0 at
 * terminals; learn when size > max(warmup_steps, batch_size), agents/base_agent.py:54-70) -> one fused update
 * (device sampler) -> every eval_interval steps eval_episodes greedy test episodes (which reset the OU noise
 * mid-episode: quirk Q8).  Evaluation 0 runs before the first training step.  Random streams (environment
 * resets, OU normals, minibatch indices) are Philox streams keyed by the agent's seed.
 * ------------------------------------------------------------------------------------------------- */
#define RLC_ENV_PENDULUM_V0 1          /* gym 0.18 Pendulum-v0 (third-party; restated, float64 simulator) */
typedef struct rlc_rollout_config {
    int32_t env_id;                    /* RLC_ENV_PENDULUM_V0 */
    int32_t episode_steps_limit;       /* EPISODE_STEPS_LIMIT (environments/environments.py:40-46) */
    int64_t total_steps_limit;         /* TOTAL_STEPS_LIMIT */
    int64_t eval_interval;             /* training steps between evaluations (>= 1) */
    int32_t eval_episodes;
    int32_t warmup_steps;              /* utils/config.py:14 */
    int32_t max_train_episodes;        /* capacity of the per-agent training-episode log */
    int32_t reserved0;
    double gamma;                      /* utils/config.py:15 */
} rlc_rollout_config;

int rlc_ddpg_rollout_create(rlc_ddpg* h, const rlc_rollout_config* cfg);
/* advance every agent by up to n_steps training steps (stops at total_steps_limit); returns when the GPU is done.
 * out_total_steps (may be NULL) receives the training steps taken so far. */
int rlc_ddpg_rollout_run(rlc_ddpg* h, int64_t n_steps, int64_t* out_total_steps);
/* the same loop for a SoftActorCritic population (agents/SoftActorCritic.py:55-126): training actions are
 * reparameterised samples of the current policy (exploration_policy 'none'), evaluation uses the mean action
 * (sample_for_eval "False"); the N(0,1) draws come from the agent's Philox stream */
int rlc_sac_rollout_create(rlc_sac* h, const rlc_rollout_config* cfg);
int rlc_sac_rollout_run(rlc_sac* h, int64_t n_steps, int64_t* out_total_steps);
/* the same loop for a ReverseKL / ForwardKL population (agents/ReverseKL.py:31-81): training actions are samples
 * tanh(mean + std*eps)*action_max of the current policy, evaluation uses tanh(mean)*action_max (sample_for_eval "False") */
int rlc_kl_rollout_create(rlc_kl* h, const rlc_rollout_config* cfg);
int rlc_kl_rollout_run(rlc_kl* h, int64_t n_steps, int64_t* out_total_steps);
/* the same loop for a NAF population (agents/NAF.py:24-75): training actions are draws from
 * N(mu, noise_scale * pinv(L L^T)) clipped to the action bounds (naf_network.py:152-176; on the device
 * mu + sqrt(noise_scale) L^-T z with Philox normals z), evaluation uses the greedy action.
 * noise_scale: [n_agents] (the value naf.json sweeps) */
int rlc_naf_rollout_create(rlc_naf* h, const rlc_rollout_config* cfg, const float* noise_scale);
int rlc_naf_rollout_run(rlc_naf* h, int64_t n_steps, int64_t* out_total_steps);
/* counts of one agent: finished training episodes, evaluations run, training steps taken */
int rlc_rollout_counts(rlc_handle* h, int32_t agent, int64_t* n_train_episodes, int64_t* n_evals,
                       int64_t* total_steps);
/* first n finished training episodes: returns, lengths, cumulative step count at the end of each
 * (train_rewards_per_episode, train_steps_per_episode, train_cum_steps of experiment.py:96-98) */
int rlc_rollout_train_log(rlc_handle* h, int32_t agent, int64_t n, double* returns, int32_t* lengths,
                          int64_t* cum_steps);
/* first n evaluations: returns / lengths [n][eval_episodes] (eval_rewards_per_episode, eval_steps_per_episode) */
int rlc_rollout_eval_log(rlc_handle* h, int32_t agent, int64_t n, double* returns, int32_t* lengths);
/* current training observation [state_dim] and episode step of one agent (parity tap) */
int rlc_rollout_observation(rlc_handle* h, int32_t agent, double* obs, int32_t* episode_step);

/* -- timing on the handle's stream (hipEvents): bench.py's roofline.achieved */
int rlc_timer_begin(rlc_handle* h);
int rlc_timer_end(rlc_handle* h, float* out_ms);   /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* RLCONTROL_HIP_H */
