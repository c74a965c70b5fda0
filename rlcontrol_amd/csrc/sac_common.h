// sac_common.h -- geometry and device view of the SoftActorCritic (SAC-v1) population.
// Blob = variable creation order under 'main' (agents/network/sac_network.py:152-172):
//   pi: W1[S,L1a] b1 W2[L1a,L2a] b2 Wm[L2a,A] bm Ws[L2a,A] bs | qf: W1[S,L1c] b1 W2[L1c+A,L2c] b2 W3[L2c] b3 |
//   vf: W1[S,L1c] b1 W2[L1c,L2c] b2 W3[L2c] b3
// Device layout pads every tensor to 64 floats (same scheme as RlcDims); the three L1 x L2 matrices are tile-blocked
// when the MFMA kernel is in use (RlcSacDims::blocked); the ABI blob is compact row-major.
#pragma once
#include "rlc_common.h"

#define RLC_SAC_NSEG 32

struct RlcSacDims {
    int S, A, L1A, L2A, L1C, L2C, B;
    int blocked;     // 1: pW2 / qW2 / vW2 segments use the tile-blocked layout of rlc_common.h (MFMA kernel)
    int arow0;       // device row of qW2's first action row: L1C (row-major) or the next multiple of 16 (blocked)
    int qcat;        // 0: the action joins Q at layer 2 (sac_network.py:183-196); 1: [state, action] is Q's INPUT
                     // (the ReverseKL / ForwardKL SoftQNetwork, reversekl_network.py:257-276): qW1[S+A,L1c], qW2[L1c,L2c]
    int norm;        // 1: config.norm_type 'layer' -- tf.contrib.layers.layer_norm before every hidden relu
                     // (base_network.py:53-56): each hidden layer adds beta then gamma behind its bias
    int pW1, pb1, pW2, pb2, pWm, pbm, pWs, pbs, qW1, qb1, qW2, qb2, qW3, qb3, vW1, vb1, vW2, vb2, vW3, vb3;
    int pL1b, pL1g, pL2b, pL2g, qL1b, qL1g, qL2b, qL2g, vL1b, vL1g, vL2b, vL2g;      // layer-norm beta / gamma (norm only)
    int Ppi_dev;     // device offset where the qf block starts (pi optimizer owns [0, Ppi_dev))
    int P, Pdev, Ppad, nseg;
    int seg_len[RLC_SAC_NSEG], seg_compact[RLC_SAC_NSEG], seg_dev[RLC_SAC_NSEG];
    int seg_rows[RLC_SAC_NSEG], seg_cols[RLC_SAC_NSEG], seg_h[RLC_SAC_NSEG];
    char seg_big[RLC_SAC_NSEG];
};

inline RlcSacDims rlc_sac_make_dims(int S, int A, int L1A, int L2A, int L1C, int L2C, int B, int blocked = 0,
                                    int qcat = 0, int norm = 0) {
    RlcSacDims d;
    d.S = S; d.A = A; d.L1A = L1A; d.L2A = L2A; d.L1C = L1C; d.L2C = L2C; d.B = B;
    d.blocked = blocked; d.qcat = qcat; d.norm = norm;
    d.arow0 = blocked ? ((L1C + 15) & ~15) : L1C;
    int n = 0;
    int* slot[RLC_SAC_NSEG];
    auto seg = [&](int* where, int r, int c, int big, int h) {
        d.seg_rows[n] = r; d.seg_cols[n] = c; d.seg_big[n] = (char)big; d.seg_h[n] = h;
        slot[n++] = where;
    };
    auto vec = [&](int* where, int c) { seg(where, 1, c, 0, 1); };
    auto ln = [&](int* wb, int* wg, int c) { if (norm) { vec(wb, c); vec(wg, c); } };
    // pi
    seg(&d.pW1, S, L1A, 0, S); vec(&d.pb1, L1A); ln(&d.pL1b, &d.pL1g, L1A);
    seg(&d.pW2, L1A, L2A, 1, L1A); vec(&d.pb2, L2A); ln(&d.pL2b, &d.pL2g, L2A);
    seg(&d.pWm, L2A, A, 0, L2A); vec(&d.pbm, A); seg(&d.pWs, L2A, A, 0, L2A); vec(&d.pbs, A);
    // qf
    const int q1r = qcat ? S + A : S, q2r = qcat ? L1C : L1C + A;
    seg(&d.qW1, q1r, L1C, 0, q1r); vec(&d.qb1, L1C); ln(&d.qL1b, &d.qL1g, L1C);
    seg(&d.qW2, q2r, L2C, 1, L1C); vec(&d.qb2, L2C); ln(&d.qL2b, &d.qL2g, L2C);
    seg(&d.qW3, L2C, 1, 0, L2C); vec(&d.qb3, 1);
    // vf
    seg(&d.vW1, S, L1C, 0, S); vec(&d.vb1, L1C); ln(&d.vL1b, &d.vL1g, L1C);
    seg(&d.vW2, L1C, L2C, 1, L1C); vec(&d.vb2, L2C); ln(&d.vL2b, &d.vL2g, L2C);
    seg(&d.vW3, L2C, 1, 0, L2C); vec(&d.vb3, 1);
    d.nseg = n;
    d.pL1b = d.pL1g = d.pL2b = d.pL2g = d.qL1b = d.qL1g = d.qL2b = d.qL2g = d.vL1b = d.vL1g = d.vL2b = d.vL2g = 0;
    rlc_layout_segs(d);
    for (int i = 0; i < n; i++) *slot[i] = d.seg_dev[i];
    d.Ppi_dev = d.qW1;
    return d;
}

struct RlcSacDev {
    RlcSacDims d;
    RlcReplayDev rep;
    int n_agents;
    int clip_state;
    float tau, smin0, smax0, amax0;
    float *theta, *theta_t, *m, *v;      // [n_agents][Ppad]; pi-Adam owns the pi block, value-Adam the rest
    float* pw;                           // [n_agents][4] {pi b1^t, pi b2^t, value b1^t, value b2^t}
    const float *pi_lr, *qv_lr, *alpha;  // [n_agents]
    unsigned long long* noise_ctr;       // [n_agents] Philox draws of eps so far
    // taps of the last update [n_agents][RLC_MAX_BATCH]: q, v, logp, q_pi; losses [n_agents][4]; grads [n_agents][Ppad]
    float *tap_q, *tap_v, *tap_logp, *tap_qpi, *tap_loss, *tap_g;
    float* scratch;
    long long scratch_stride;
    // ---- ReverseKL / ForwardKL populations (kl_generic.hip; zero for SoftActorCritic) ----
    int kl_kind;                         // RLC_KL_REVERSE / RLC_KL_FORWARD
    int kl_optim;                        // RLC_KL_OPTIM_*
    int kl_qupdate;                      // RLC_KL_Q_NON_SAC / RLC_KL_Q_SAC
    int kl_nodes;                        // quadrature nodes of the action integral
    const float *kl_node_a, *kl_node_w;  // [kl_nodes] node actions (already scaled by action_max) and weights
    int* kl_step;                        // [n_agents] Adam steps taken (torch keeps the step, not the beta powers)
    float* kl_tap_iq;                    // [n_agents][B * kl_nodes] Q at the nodes of the last update
};

size_t rlc_sac_scratch_floats(const RlcSacDims& d);
// eps_dev: [n_agents][n_updates][B][A] injected N(0,1) draws, or null -> device Philox
struct RlcSacRollout;   // sac_rollout_device.h: {RlcSacDev, RlcEnvDev} in device memory
// rollout (device pointer, may be null): every iteration first takes one environment step of the on-device loop
int rlc_launch_sac_update(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                          const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                          const RlcSacRollout* rollout = nullptr);
// MFMA-tiled fused update (dims must satisfy rlc_sac_mfma_supported; tile-blocked layout)
bool rlc_sac_mfma_supported(const RlcSacDims& d);
int rlc_launch_sac_update_mfma(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                               const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                               const RlcSacRollout* rollout = nullptr);
// ReverseKL / ForwardKL fused update and acting (kl_generic.hip); same argument meaning as the SAC launches
size_t rlc_kl_scratch_floats(const RlcSacDims& d, int nodes);
int rlc_launch_kl_update(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                         const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                         const RlcSacRollout* rollout = nullptr);
bool rlc_kl_mfma_supported(const RlcSacDims& d, int nodes);
size_t rlc_kl_mfma_scratch_floats(const RlcSacDims& d, int nodes);   // floats of an agent's scratch row the MFMA kernel needs
size_t rlc_kl_split_zbuf_floats(const RlcSacDims& d);
int rlc_kl_split_grid(int n_agents, int C);
int rlc_launch_kl_update_mfma_split(const RlcSacDev& dv, float* zbuf, unsigned int* bar, int* err, int C, int first_agent,
                                    int n_agents, int n_updates, int source, const long long* idx_dev, const float* eps_dev,
                                    int grad_taps, hipStream_t st);
int rlc_launch_kl_update_mfma(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                              const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                              const RlcSacRollout* rollout = nullptr);
int rlc_launch_kl_act(const RlcSacDev& dv, int first_agent, int n, const float* states_dev, const float* eps_dev,
                      int sample, float* out_dev, hipStream_t st, int* done_flag = nullptr, int done_val = 0);
int rlc_launch_sac_eval(const RlcSacDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st);
// one state per agent; sample = 0 mean action, 1 reparameterised sample (eps_dev [n][A] or null -> Philox)
int rlc_launch_sac_act(const RlcSacDev& dv, int first_agent, int n, const float* states_dev, const float* eps_dev,
                       int sample, float* out_dev, hipStream_t st, int* done_flag = nullptr, int done_val = 0);
