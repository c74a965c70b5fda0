// naf_common.h -- geometry and device view of the NAF population.
// Blob = variable creation order (agents/network/naf_network.py:79-107):
//   W1[S,L1] b1 | Wa2[L1,L2] ba2 | Wa3[L2,A] ba3 | Wv2[L1,L2] bv2 | Wv3[L2] bv3 |
//   for c < A: Wd_c[L1] bd_c | for c < A-1: Wn_c[L1,A-1-c] bn_c[A-1-c]
// norm_type 'layer' (RlcNafDims::norm): every hidden layer is followed by its layer-norm beta, gamma
// (tf.contrib.layers.layer_norm creates them right after the fully_connected it normalises, base_network.py:53-56):
//   W1 b1 L1b L1g | Wa2 ba2 La2b La2g | Wa3 ba3 | Wv2 bv2 Lv2b Lv2g | Wv3 bv3 | ...
// Device layout pads every tensor to 64 floats; Wa2 / Wv2 are tile-blocked when the MFMA kernel is in use
// (RlcNafDims::blocked); the ABI blob is compact row-major.
#pragma once
#include "rlc_common.h"

#define RLC_NAF_MAX_A 6
#define RLC_NAF_MAX_SEG (16 + 4 * RLC_NAF_MAX_A)

struct RlcNafDims {
    int S, A, L1, L2, B, NN;    // NN = A(A-1)/2 below-diagonal entries
    int blocked;                // 1: Wa2 / Wv2 segments use the tile-blocked layout of rlc_common.h (MFMA kernel)
    int norm;                   // 1: config.norm_type 'layer' -- layer norm before every hidden relu (generic kernel only)
    int W1, b1, Wa2, ba2, Wa3, ba3, Wv2, bv2, Wv3, bv3;
    int L1b, L1g, La2b, La2g, Lv2b, Lv2g;     // layer-norm beta / gamma offsets (norm only)
    int Wd[RLC_NAF_MAX_A], bd[RLC_NAF_MAX_A], Wn[RLC_NAF_MAX_A], bn[RLC_NAF_MAX_A];
    int P, Pdev, Ppad, nseg;
    int seg_len[RLC_NAF_MAX_SEG], seg_compact[RLC_NAF_MAX_SEG], seg_dev[RLC_NAF_MAX_SEG];
    int seg_rows[RLC_NAF_MAX_SEG], seg_cols[RLC_NAF_MAX_SEG], seg_h[RLC_NAF_MAX_SEG];
    char seg_big[RLC_NAF_MAX_SEG];
};

inline RlcNafDims rlc_naf_make_dims(int S, int A, int L1, int L2, int B, int blocked = 0, int norm = 0) {
    RlcNafDims d;
    d.S = S; d.A = A; d.L1 = L1; d.L2 = L2; d.B = B; d.NN = A * (A - 1) / 2;
    d.blocked = blocked; d.norm = norm;
    d.L1b = d.L1g = d.La2b = d.La2g = d.Lv2b = d.Lv2g = 0;
    int n = 0;
    int* slot[RLC_NAF_MAX_SEG];     // where the device offset of segment i goes once the layout is known
    auto seg = [&](int* where, int r, int c, int big) {
        slot[n] = where; d.seg_rows[n] = r; d.seg_cols[n] = c; d.seg_h[n] = r; d.seg_big[n] = (char)big; n++;
    };
    auto ln = [&](int* b, int* g, int c) { if (norm) { seg(b, 1, c, 0); seg(g, 1, c, 0); } };
    seg(&d.W1, S, L1, 0); seg(&d.b1, 1, L1, 0); ln(&d.L1b, &d.L1g, L1);
    seg(&d.Wa2, L1, L2, 1); seg(&d.ba2, 1, L2, 0); ln(&d.La2b, &d.La2g, L2);
    seg(&d.Wa3, L2, A, 0); seg(&d.ba3, 1, A, 0);
    seg(&d.Wv2, L1, L2, 1); seg(&d.bv2, 1, L2, 0); ln(&d.Lv2b, &d.Lv2g, L2);
    seg(&d.Wv3, L2, 1, 0); seg(&d.bv3, 1, 1, 0);
    for (int c = 0; c < A; c++) { seg(&d.Wd[c], L1, 1, 0); seg(&d.bd[c], 1, 1, 0); }
    for (int c = 0; c < A - 1; c++) { seg(&d.Wn[c], L1, A - 1 - c, 0); seg(&d.bn[c], 1, A - 1 - c, 0); }
    d.nseg = n;
    rlc_layout_segs(d);
    for (int i = 0; i < n; i++) *slot[i] = d.seg_dev[i];
    return d;
}

struct RlcNafDev {
    RlcNafDims d;
    RlcReplayDev rep;
    int n_agents;
    int clip_state;
    float tau;
    float *theta, *theta_t, *m, *v;   // [n_agents][Ppad]
    float* pw;                        // [n_agents][2] beta powers
    const float* lr;                  // [n_agents]
    const float *smin, *smax, *amax;
    const float* amin;            // lower clip of the exploration draw (on-device loop)
    float *tap_q, *tap_y, *tap_V, *tap_g;
    float* scratch;
    long long scratch_stride;
};

size_t rlc_naf_scratch_floats(const RlcNafDims& d);
struct RlcNafRollout;   // naf_rollout_device.h: {RlcNafDev, RlcEnvDev, noise} in device memory
// rollout (device pointer, may be null): every iteration first takes one environment step of the on-device loop
int rlc_launch_naf_update(const RlcNafDev& dv, int first_agent, int n_agents, int n_updates, int source,
                          const long long* idx_dev, int grad_taps, hipStream_t st, const RlcNafRollout* rollout = nullptr);
// MFMA-tiled fused update (dims must satisfy rlc_naf_mfma_supported; tile-blocked layout)
bool rlc_naf_mfma_supported(const RlcNafDims& d);
int rlc_launch_naf_update_mfma(const RlcNafDev& dv, int first_agent, int n_agents, int n_updates, int source,
                               const long long* idx_dev, int grad_taps, hipStream_t st, const RlcNafRollout* rollout = nullptr);
int rlc_launch_naf_eval(const RlcNafDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st);
// greedy action mu [n][A] and the L columns [n][A(A+1)/2] (column c = diag_c, then its below-diagonal entries)
int rlc_launch_naf_act(const RlcNafDev& dv, int first_agent, int n, const float* states_dev, float* mu_dev,
                       float* lcols_dev, hipStream_t st, int* done_flag = nullptr, int done_val = 0);
