// sac_generic.hip -- fused SoftActorCritic (SAC-v1) update + acting kernels (any-shape fp32 VALU path).
//
// One workgroup per agent, n_updates sequential updates per launch; each update = sample_batch
// (utils/replaybuffer.py:32-37) + SoftActorCritic_Network_Manager.update_network
// (agents/SoftActorCritic.py:113-126): ONE Session.run(train_ops) of agents/network/sac_network.py:107-136
// + update_target_network.  Everything the reference evaluates in that run is computed from the
// PRE-update weights (pi, Q(s,a), Q(s,pi), V(s), V'(s')), then pi-Adam, then value-Adam, then Polyak.
// Reference quirks kept (see oracle/sac_oracle.c for the derivation):
//   Q9   logp_pi is [B] while q_pi, v are [B,1]: v regresses onto q_pi[i] - alpha*mean_j(logp[j]);
//   the state clip of pi and V uses the SCALARS state_min[0]/state_max[0]; Q sees the raw state;
//   actions are scaled by action_max[0]; gaussian_likelihood divides by std + 1e-6; the tanh-squash
//   correction is log(clip(1 - pi^2, 0, 1) + 1e-6) with the clip passing gradients.
// r and gamma enter fp32 placeholders in this agent (sac_network.py:51-52), so the replay's float64
// values are cast to fp32 at gather time.
#include "generic_blocks.h"
#include "sac_rollout_device.h"
#include "sac_common.h"

namespace {

using namespace gen;

struct SLds {
    float *x, *xc, *x2c, *a, *api, *eps, *mu, *lsp, *t, *sd, *pit, *dmu, *dls, *r, *g;
    float *q, *qpi, *v, *vt, *logp, *dout, *red;
    long long* idx;
    int* pool;
    int* dups;
    float* pol;       // scratch of the on-device training step (sac_rollout_device.h)
};

__host__ __device__ inline size_t slds_carve(const RlcSacDims& d, unsigned char* base, SLds* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int B = d.B, S = d.S, A = d.A;
    SLds L;
    L.idx = (long long*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.x = (float*)take(sizeof(float) * B * S);
    L.xc = (float*)take(sizeof(float) * B * S);
    L.x2c = (float*)take(sizeof(float) * B * S);
    float** pa[] = {&L.a, &L.api, &L.eps, &L.mu, &L.lsp, &L.t, &L.sd, &L.pit, &L.dmu, &L.dls};
    for (auto p : pa) *p = (float*)take(sizeof(float) * B * A);
    float** pb[] = {&L.r, &L.g, &L.q, &L.qpi, &L.v, &L.vt, &L.logp, &L.dout};
    for (auto p : pb) *p = (float*)take(sizeof(float) * B);
    L.red = (float*)take(sizeof(float) * 16);
    L.pool = (int*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (int*)take(sizeof(int) * 4);
    L.pol = (float*)take(sizeof(float) * (sac_policy_lds_floats(d) + 4));
    if (out) *out = L;
    return off;
}

__global__ __launch_bounds__(kThreads) void rlc_sac_update_kernel(RlcSacDev dv_arg, int first_agent, int n_updates,
                                                                  int source, const long long* host_idx,
                                                                  const float* eps_in, int grad_taps,
                                                                  const RlcSacRollout* rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the population view is read through gen::kernarg_view (generic_blocks.h), made opaque again by SAC_PHASE() at the
    // start of every phase (dv_arg is the first argument: offset 0)
    const RlcSacDev* dvp;
#define SAC_PHASE() (dvp = kernarg_view<RlcSacDev>())
#define dv (*dvp)
#define d (dvp->d)
    SAC_PHASE();
    const int S = d.S, A = d.A, L1A = d.L1A, L2A = d.L2A, L1C = d.L1C, L2C = d.L2C, B = d.B;
    const int agent = first_agent + blockIdx.x;
    const int tid = threadIdx.x;
    SLds L;
    slds_carve(d, smem, &L);
    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* mm = dv.m + (size_t)agent * d.Ppad;
    float* vv = dv.v + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 4;
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;
    const float alpha_ent = dv.alpha[agent], amax0 = dv.amax0;
    // scratch carve
    float* sc = dv.scratch + (size_t)agent * dv.scratch_stride;
    float* ph1 = sc;                sc += (size_t)B * L1A;
    float* ph2 = sc;                sc += (size_t)B * L2A;
    float* qh1 = sc;                sc += (size_t)B * L1C;
    float* qh2 = sc;                sc += (size_t)B * L2C;
    float* qh2p = sc;               sc += (size_t)B * L2C;
    float* vh1 = sc;                sc += (size_t)B * L1C;
    float* vh2 = sc;                sc += (size_t)B * L2C;
    float* th1 = sc;                sc += (size_t)B * L1C;
    float* th2 = sc;                sc += (size_t)B * L2C;
    float* dp2 = sc;                sc += (size_t)B * L2A;
    float* dp1 = sc;                sc += (size_t)B * L1A;
    float* dq2 = sc;                sc += (size_t)B * L2C;
    float* dq1 = sc;                sc += (size_t)B * L1C;
    float* dv2 = sc;                sc += (size_t)B * L2C;
    float* dv1 = sc;                sc += (size_t)B * L1C;
    // layer norm (norm_type 'layer'): normalised pre-activations and 1/std of every hidden layer that is differentiated,
    // one throw-away pair for the target network
    const int NORM = d.norm;
    const size_t lnw = NORM ? 1 : 0;
    float* pn1 = sc;                sc += lnw * B * L1A;
    float* pn2 = sc;                sc += lnw * B * L2A;
    float* qn1 = sc;                sc += lnw * B * L1C;
    float* qn2 = sc;                sc += lnw * B * L2C;
    float* qn2p = sc;               sc += lnw * B * L2C;
    float* vn1 = sc;                sc += lnw * B * L1C;
    float* vn2 = sc;                sc += lnw * B * L2C;
    float* tn = sc;                 sc += lnw * B * (L1C > L2C ? L1C : L2C);
    float* rsd = sc;                sc += lnw * 8 * RLC_MAX_BATCH;      // rstd rows: pi1 pi2 q1 q2 q2pi v1 v2 target
    float* gtmp = sc;               sc += lnw * B * L2C;                // d Q(s,pi) / d (layer-2 linear output)
    // Y = act(X W + [E We] + b) with layer norm before the relu when NORM (hidden layers only)
    auto hidden = [&](const float* X, int K, const float* E, int Ke, const float* P, int oW, int ob, int olb, int olg, int N,
                      float* Y, float* nh, float* rs) {
        blk_dense(X, K, K, E, Ke, P + oW, P + ob, N, Y, N, B, NORM ? 0 : 1);
        if (NORM) {
            __syncthreads();
            blk_layernorm_relu(Y, N, B, P + olb, P + olg, nh, rs);
        }
    };
    // layer-norm backward of a hidden layer whose masked output gradient is dY (in place -> gradient of the linear
    // output, with the PRE-step gamma); returns this thread's (gamma, beta) gradient column sums
    auto ln_bwd = [&](float* dY, const float* nh, const float* rs, int olg, int N, float& gg, float& gb) {
        if (!NORM) return;
        __syncthreads();
        blk_layernorm_param_grads(dY, nh, N, B, gg, gb);
        __syncthreads();
        blk_layernorm_bwd_rows(dY, nh, rs, th + olg, N, B);
        __syncthreads();
    };
    auto ln_adam = [&](const AdamCtx& c, int olb, int olg, int N, float gg, float gb) {
        if (NORM && tid < N) { adam_apply(c, olg + tid, gg); adam_apply(c, olb + tid, gb); }
    };
    const float EPS = 1e-6f, LOG2PI = 1.8378770664093453f, HALF_RANGE = 0.5f * (2.0f - (-20.0f));

    for (int u = 0; u < n_updates; u++) {
        if (rollout) {
            // on-device experiment loop: one environment step first; update when learn() would run
            if (!rlc_sac_train_step_device(rollout, agent, L.pol)) continue;
        }
        // ---- sample + gather ----
        SAC_PHASE();
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads) L.idx[b] = host_idx[((size_t)blockIdx.x * n_updates + u) * B + b];
        }
        __syncthreads();
        const unsigned long long nctr = dv.noise_ctr[agent];
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * A; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = (float)dv.rep.gr[slot]; L.g[b] = (float)dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * A; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = (float)dv.rep.rr[slot]; L.g[b] = (float)dv.rep.rg[slot];
            }
            for (int i = 0; i < S; i++) {
                L.x[b * S + i] = ps[i];
                L.xc[b * S + i] = rlc_clip_scalar(ps[i], dv.clip_state, dv.smin0, dv.smax0);
                L.x2c[b * S + i] = rlc_clip_scalar(ps2[i], dv.clip_state, dv.smin0, dv.smax0);
            }
            for (int j = 0; j < A; j++) {
                L.a[b * A + j] = pa[j];
                float e;
                if (eps_in) {
                    e = eps_in[(((size_t)blockIdx.x * n_updates + u) * B + b) * A + j];
                } else {
                    const Philox4 p = philox4x32_10(dv.rep.seed[agent] ^ 0x9E3779B97F4A7C15ull, nctr,
                                                    (unsigned long long)(b * A + j) >> 1);
                    float n0, n1;
                    philox_normal2(p, n0, n1);
                    e = ((b * A + j) & 1) ? n1 : n0;
                }
                L.eps[b * A + j] = e;
            }
        }
        __syncthreads();
        if (tid == 0 && !eps_in) dv.noise_ctr[agent] = nctr + 1;

        // ---- forward: pi ----
        SAC_PHASE();
        hidden(L.xc, S, nullptr, 0, th, d.pW1, d.pb1, d.pL1b, d.pL1g, L1A, ph1, pn1, rsd + 0 * RLC_MAX_BATCH);
        __syncthreads();
        hidden(ph1, L1A, nullptr, 0, th, d.pW2, d.pb2, d.pL2b, d.pL2g, L2A, ph2, pn2, rsd + 1 * RLC_MAX_BATCH);
        __syncthreads();
        blk_dense(ph2, L2A, L2A, nullptr, 0, th + d.pWm, th + d.pbm, A, L.mu, A, B, 0);
        blk_dense(ph2, L2A, L2A, nullptr, 0, th + d.pWs, th + d.pbs, A, L.lsp, A, B, 0);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            float lp = 0.0f;
            for (int j = 0; j < A; j++) {
                const int k = b * A + j;
                const float t = tanhf(L.lsp[k]);
                const float log_std = -20.0f + HALF_RANGE * (t + 1.0f);
                const float sd = expf(log_std);
                const float uu = L.mu[k] + L.eps[k] * sd;
                const float z = (uu - L.mu[k]) / (sd + EPS);
                lp += -0.5f * (z * z + 2.0f * log_std + LOG2PI);
                const float pt = tanhf(uu);
                L.t[k] = t; L.sd[k] = sd; L.pit[k] = pt; L.api[k] = pt * amax0;
            }
            for (int j = 0; j < A; j++) {
                const float pt = L.pit[b * A + j];
                lp -= logf(fminf(fmaxf(1.0f - pt * pt, 0.0f), 1.0f) + 1e-6f);
            }
            L.logp[b] = lp;
            dv.tap_logp[(size_t)agent * RLC_MAX_BATCH + b] = lp;
        }
        // ---- forward: Q(s,a), Q(s,pi) (raw state), V(s), V'(s') ----
        SAC_PHASE();
        if (!NORM) {
            blk_dense(L.x, S, S, nullptr, 0, th + d.qW1, th + d.qb1, L1C, qh1, L1C, B, 1);
            blk_dense(L.xc, S, S, nullptr, 0, th + d.vW1, th + d.vb1, L1C, vh1, L1C, B, 1);
            blk_dense(L.x2c, S, S, nullptr, 0, tt + d.vW1, tt + d.vb1, L1C, th1, L1C, B, 1);
            __syncthreads();
            blk_dense(qh1, L1C, L1C, L.a, A, th + d.qW2, th + d.qb2, L2C, qh2, L2C, B, 1);
            blk_dense(qh1, L1C, L1C, L.api, A, th + d.qW2, th + d.qb2, L2C, qh2p, L2C, B, 1);
            blk_dense(vh1, L1C, L1C, nullptr, 0, th + d.vW2, th + d.vb2, L2C, vh2, L2C, B, 1);
            blk_dense(th1, L1C, L1C, nullptr, 0, tt + d.vW2, tt + d.vb2, L2C, th2, L2C, B, 1);
            __syncthreads();
        } else {
            // the layer-norm pass of a layer needs a barrier after its dense pass: one layer at a time
            hidden(L.x, S, nullptr, 0, th, d.qW1, d.qb1, d.qL1b, d.qL1g, L1C, qh1, qn1, rsd + 2 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(L.xc, S, nullptr, 0, th, d.vW1, d.vb1, d.vL1b, d.vL1g, L1C, vh1, vn1, rsd + 5 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(L.x2c, S, nullptr, 0, tt, d.vW1, d.vb1, d.vL1b, d.vL1g, L1C, th1, tn, rsd + 7 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(qh1, L1C, L.a, A, th, d.qW2, d.qb2, d.qL2b, d.qL2g, L2C, qh2, qn2, rsd + 3 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(qh1, L1C, L.api, A, th, d.qW2, d.qb2, d.qL2b, d.qL2g, L2C, qh2p, qn2p, rsd + 4 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(vh1, L1C, nullptr, 0, th, d.vW2, d.vb2, d.vL2b, d.vL2g, L2C, vh2, vn2, rsd + 6 * RLC_MAX_BATCH);
            __syncthreads();
            hidden(th1, L1C, nullptr, 0, tt, d.vW2, d.vb2, d.vL2b, d.vL2g, L2C, th2, tn, rsd + 7 * RLC_MAX_BATCH);
            __syncthreads();
        }
        blk_dense(qh2, L2C, L2C, nullptr, 0, th + d.qW3, th + d.qb3, 1, L.q, 1, B, 0);
        blk_dense(qh2p, L2C, L2C, nullptr, 0, th + d.qW3, th + d.qb3, 1, L.qpi, 1, B, 0);
        blk_dense(vh2, L2C, L2C, nullptr, 0, th + d.vW3, th + d.vb3, 1, L.v, 1, B, 0);
        blk_dense(th2, L2C, L2C, nullptr, 0, tt + d.vW3, tt + d.vb3, 1, L.vt, 1, B, 0);
        __syncthreads();
        float part_lp = 0.0f, part_qp = 0.0f;
        for (int b = tid; b < B; b += kThreads) { part_lp += L.logp[b]; part_qp += L.qpi[b]; }
        const float mean_logp = blk_sum(part_lp, L.red) / (float)B;
        const float mean_qpi = blk_sum(part_qp, L.red) / (float)B;
        {   // taps + losses (the reference fetches pi_loss, q_loss, v_loss, q, v, logp_pi: sac_network.py:135-136)
            float ql = 0.0f, vl = 0.0f;
            for (int b = tid; b < B; b += kThreads) {
                dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = L.q[b];
                dv.tap_v[(size_t)agent * RLC_MAX_BATCH + b] = L.v[b];
                dv.tap_qpi[(size_t)agent * RLC_MAX_BATCH + b] = L.qpi[b];
                const float e = (L.r[b] + L.g[b] * L.vt[b]) - L.q[b];
                ql += e * e;
                for (int j = 0; j < B; j++) {
                    const float f = L.qpi[b] - alpha_ent * L.logp[j] - L.v[b];
                    vl += f * f;
                }
            }
            ql = blk_sum(ql, L.red);
            vl = blk_sum(vl, L.red);
            if (tid == 0) {
                dv.tap_loss[agent * 4 + 0] = alpha_ent * mean_logp - mean_qpi;
                dv.tap_loss[agent * 4 + 1] = 0.5f * ql / (float)B;
                dv.tap_loss[agent * 4 + 2] = 0.5f * vl / ((float)B * (float)B);
            }
        }

        // ---- pi backward seeds: d(alpha*mean(logp) - mean(Q(s,pi))) / d(mu_raw, ls_pre) ----
        SAC_PHASE();
        if (NORM) {
            // d Q(s,pi) / d (layer-2 linear output): the head's weights through the relu mask and the layer norm
            for (int it = tid; it < B * L2C; it += kThreads) gtmp[it] = qh2p[it] > 0.0f ? th[d.qW3 + it % L2C] : 0.0f;
            __syncthreads();
            blk_layernorm_bwd_rows(gtmp, qn2p, rsd + 4 * RLC_MAX_BATCH, th + d.qL2g, L2C, B);
            __syncthreads();
        }
        for (int it = tid; it < B * A; it += kThreads) {
            const int b = it / A, j = it % A;
            float ga = 0.0f;
            if (NORM) {
                for (int n = 0; n < L2C; n++) ga += gtmp[(size_t)b * L2C + n] * th[d.qW2 + (size_t)(L1C + j) * L2C + n];
            } else
            for (int n = 0; n < L2C; n++)
                if (qh2p[(size_t)b * L2C + n] > 0.0f) ga += th[d.qW3 + n] * th[d.qW2 + (size_t)(L1C + j) * L2C + n];
            const float pt = L.pit[it], om = 1.0f - pt * pt;
            const float dlogp_dpit = 2.0f * pt / (fminf(fmaxf(om, 0.0f), 1.0f) + 1e-6f);
            const float dL_dpit = (-1.0f / (float)B) * ga * amax0 + (alpha_ent / (float)B) * dlogp_dpit;
            const float dL_du = dL_dpit * om;
            const float sd = L.sd[it], e = L.eps[it];
            const float z = e * sd / (sd + EPS);
            const float dz_dls = e * sd * EPS / ((sd + EPS) * (sd + EPS));
            const float dL_dlogstd = dL_du * e * sd + (alpha_ent / (float)B) * (-z * dz_dls - 1.0f);
            L.dmu[it] = dL_du;
            L.dls[it] = dL_dlogstd * HALF_RANGE * (1.0f - L.t[it] * L.t[it]);
        }
        // ---- value backward seeds ----
        SAC_PHASE();
        for (int b = tid; b < B; b += kThreads) {
            L.dout[b] = -((L.r[b] + L.g[b] * L.vt[b]) - L.q[b]) / (float)B;                  // dq_loss/dq
            L.vt[b] = -(L.qpi[b] - alpha_ent * mean_logp - L.v[b]) / (float)B;               // dv_loss/dv (Q9); vt reused
        }
        __syncthreads();
        // hidden-layer gradients, all with the pre-update weights
        SAC_PHASE();
        blk_dense_bwd_input_ex(L.dmu, A, th + d.pWm, ph2, L2A, dp2, B, false);
        for (int it = tid; it < B * L2C; it += kThreads) {
            const int b = it / L2C, n = it % L2C;
            dq2[it] = qh2[it] > 0.0f ? L.dout[b] * th[d.qW3 + n] : 0.0f;
            dv2[it] = vh2[it] > 0.0f ? L.vt[b] * th[d.vW3 + n] : 0.0f;
        }
        __syncthreads();
        blk_dense_bwd_input_ex(L.dls, A, th + d.pWs, ph2, L2A, dp2, B, true);
        float gg[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // p1 p2 q1 q2 v1 v2
        ln_bwd(dq2, qn2, rsd + 3 * RLC_MAX_BATCH, d.qL2g, L2C, gg[3], gb[3]);
        ln_bwd(dv2, vn2, rsd + 6 * RLC_MAX_BATCH, d.vL2g, L2C, gg[5], gb[5]);
        ln_bwd(dp2, pn2, rsd + 1 * RLC_MAX_BATCH, d.pL2g, L2A, gg[1], gb[1]);
        blk_dense_bwd_input(dq2, L2C, th + d.qW2, qh1, L1C, dq1, B);
        blk_dense_bwd_input(dv2, L2C, th + d.vW2, vh1, L1C, dv1, B);
        __syncthreads();
        blk_dense_bwd_input(dp2, L2A, th + d.pW2, ph1, L1A, dp1, B);
        __syncthreads();
        ln_bwd(dq1, qn1, rsd + 2 * RLC_MAX_BATCH, d.qL1g, L1C, gg[2], gb[2]);
        ln_bwd(dv1, vn1, rsd + 5 * RLC_MAX_BATCH, d.vL1g, L1C, gg[4], gb[4]);
        ln_bwd(dp1, pn1, rsd + 0 * RLC_MAX_BATCH, d.pL1g, L1A, gg[0], gb[0]);
        // ---- gradients + Adam: pi optimizer, then value optimizer (disjoint parameters) ----
        SAC_PHASE();
        {
            const AdamCtx cp = {th, mm, vv, adam_alpha(dv.pi_lr[agent], pw[0], pw[1]), tapg};
            blk_dense_grad_adam(ph2, L2A, L2A, nullptr, 0, L.dmu, A, B, cp, d.pWm, d.pbm);
            blk_dense_grad_adam(ph2, L2A, L2A, nullptr, 0, L.dls, A, B, cp, d.pWs, d.pbs);
            blk_dense_grad_adam(ph1, L1A, L1A, nullptr, 0, dp2, L2A, B, cp, d.pW2, d.pb2);
            blk_dense_grad_adam(L.xc, S, S, nullptr, 0, dp1, L1A, B, cp, d.pW1, d.pb1);
            ln_adam(cp, d.pL1b, d.pL1g, L1A, gg[0], gb[0]);
            ln_adam(cp, d.pL2b, d.pL2g, L2A, gg[1], gb[1]);
            const AdamCtx cq = {th, mm, vv, adam_alpha(dv.qv_lr[agent], pw[2], pw[3]), tapg};
            blk_dense_grad_adam(qh2, L2C, L2C, nullptr, 0, L.dout, 1, B, cq, d.qW3, d.qb3);
            blk_dense_grad_adam(qh1, L1C, L1C, L.a, A, dq2, L2C, B, cq, d.qW2, d.qb2);
            blk_dense_grad_adam(L.x, S, S, nullptr, 0, dq1, L1C, B, cq, d.qW1, d.qb1);
            blk_dense_grad_adam(vh2, L2C, L2C, nullptr, 0, L.vt, 1, B, cq, d.vW3, d.vb3);
            blk_dense_grad_adam(vh1, L1C, L1C, nullptr, 0, dv2, L2C, B, cq, d.vW2, d.vb2);
            blk_dense_grad_adam(L.xc, S, S, nullptr, 0, dv1, L1C, B, cq, d.vW1, d.vb1);
            ln_adam(cq, d.qL1b, d.qL1g, L1C, gg[2], gb[2]);
            ln_adam(cq, d.qL2b, d.qL2g, L2C, gg[3], gb[3]);
            ln_adam(cq, d.vL1b, d.vL1g, L1C, gg[4], gb[4]);
            ln_adam(cq, d.vL2b, d.vL2g, L2C, gg[5], gb[5]);
        }
        __syncthreads();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; pw[2] *= 0.9f; pw[3] *= 0.999f; }
        // ---- Polyak over every main/target pair (sac_network.py:72-73): (1-tau)*target + tau*main ----
        SAC_PHASE();
        for (int p = tid; p < d.Pdev; p += kThreads) tt[p] = (1.0f - dv.tau) * tt[p] + dv.tau * th[p];
        __syncthreads();
    }
#undef SAC_PHASE
#undef dv
#undef d
}

// mean / sampled action for one state per agent (sac_network.py:327-343): one workgroup per agent
// done_flag (or null): a word in host-visible memory that receives done_val once the launch's actions are stored
// (rlc_sac_act_queue; see rlc_ddpg_act_kernel)
__global__ __launch_bounds__(kThreads) void rlc_sac_act_kernel(RlcSacDev dv, int first_agent, const float* states,
                                                               const float* eps_in, int sample, float* out,
                                                               int* done_flag, int done_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcSacDims d = dv.d;
    const int S = d.S, A = d.A;
    const int agent = first_agent + blockIdx.x, tid = threadIdx.x;
    const SacPolicyLds L = sac_policy_carve(d, (float*)smem);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    for (int i = tid; i < S; i += kThreads)
        L.x[i] = rlc_clip_scalar(states[(size_t)blockIdx.x * S + i], dv.clip_state, dv.smin0, dv.smax0);
    if (sample && tid < A)
        L.eps[tid] = eps_in ? eps_in[(size_t)blockIdx.x * A + tid] : sac_act_eps(dv.rep.seed[agent], dv.noise_ctr[agent], tid);
    sac_policy_forward(d, th, L, dv.amax0, sample);
    if (tid < A) out[(size_t)blockIdx.x * A + tid] = L.out[tid];
    if (sample && !eps_in && tid == 0) dv.noise_ctr[agent] += 1;
    if (done_flag) {                                // (queued forward of a drop-in agent: one workgroup)
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();                     // the output stores first
            __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// evaluation of the on-device loop: one greedy (mean-action) test episode per workgroup
// (run_episode_eval, experiment.py:163-196; agents/SoftActorCritic.py:102)
__global__ __launch_bounds__(kThreads) void rlc_sac_eval_kernel(RlcSacDev dv, RlcEnvDev env, int eval_round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double sim[RLC_ENV_STATE];
    __shared__ double obs[8];
    __shared__ int s_done;
    const RlcSacDims d = dv.d;
    const int S = d.S;
    const int agent = blockIdx.x / env.eval_episodes, ep = blockIdx.x % env.eval_episodes;
    const int tid = threadIdx.x;
    const SacPolicyLds L = sac_policy_carve(d, (float*)smem);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    if (tid == 0) {
        env_reset(env.env_id, sim, obs, dv.rep.seed[agent] ^ RLC_KEY_ENV_TEST,
                  (unsigned long long)eval_round * env.eval_episodes + ep);
        s_done = 0;
    }
    __syncthreads();
    double ret = 0.0;
    int steps = 0;
    while (steps < env.episode_limit) {
        for (int i = tid; i < S; i += kThreads) L.x[i] = rlc_clip_scalar((float)obs[i], dv.clip_state, dv.smin0, dv.smax0);
        sac_policy_forward(d, th, L, dv.amax0, 0);
        if (tid == 0) {
            double reward;
            s_done = env_step(env.env_id, sim, L.out, obs, &reward, steps + 1, env.episode_limit);
            ret += reward;
        }
        steps++;
        __syncthreads();
        if (s_done) break;
    }
    if (tid == 0 && eval_round < env.max_evals) {
        const size_t at = ((size_t)agent * env.max_evals + eval_round) * env.eval_episodes + ep;
        env.eval_ret[at] = ret;
        env.eval_len[at] = steps;
    }
}

}  // namespace

size_t rlc_sac_scratch_floats(const RlcSacDims& d) {
    const size_t B = d.B;
    const size_t ln = d.norm ? B * ((size_t)d.L1A + d.L2A + 2 * d.L1C + 4 * d.L2C + (d.L1C > d.L2C ? d.L1C : d.L2C)) + 8 * RLC_MAX_BATCH
                             : 0;
    return B * ((size_t)2 * d.L1A + 2 * d.L2A + 5 * d.L1C + 6 * d.L2C) + ln;
}

int rlc_launch_sac_update(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                          const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                          const RlcSacRollout* rollout) {
    const size_t lds = slds_carve(dv.d, nullptr, nullptr);
    RLC_REQUIRE(!(rollout && eps_dev), "the on-device loop draws its own eps");
    RLC_REQUIRE(lds <= 160 * 1024, "SAC kernel needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr = false;
    if (!attr) {
        RLC_HIP(hipFuncSetAttribute((const void*)rlc_sac_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024));
        attr = true;
    }
    hipLaunchKernelGGL(rlc_sac_update_kernel, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates,
                       source, idx_dev, eps_dev, grad_taps, rollout);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_sac_act(const RlcSacDev& dv, int first_agent, int n, const float* states_dev, const float* eps_dev,
                       int sample, float* out_dev, hipStream_t st, int* done_flag, int done_val) {
    const size_t lds = sizeof(float) * sac_policy_lds_floats(dv.d);
    RLC_REQUIRE(done_flag == nullptr || n == 1, "a completion flag needs a one-workgroup acting launch");
    hipLaunchKernelGGL(rlc_sac_act_kernel, dim3(n), dim3(kThreads), lds, st, dv, first_agent, states_dev, eps_dev,
                       sample, out_dev, done_flag, done_val);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_sac_eval(const RlcSacDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st) {
    const size_t lds = sizeof(float) * sac_policy_lds_floats(dv.d);
    hipLaunchKernelGGL(rlc_sac_eval_kernel, dim3(dv.n_agents * env.eval_episodes), dim3(kThreads), lds, st, dv, env,
                       eval_round);
    RLC_HIP(hipGetLastError());
    return 0;
}
