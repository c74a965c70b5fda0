/*
 * oracle/sac_oracle.c -- CPU restatement of the reference's SoftActorCritic (SAC-v1: one Q, one V, target V).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates (reference file:line under /root/reference):
 *   agents/SoftActorCritic.py:113-126          update_network(): ONE Session.run(train_ops) + Polyak
 *   agents/network/sac_network.py:47-136       placeholders (r, gamma are fp32 [B,1] here), losses, the two Adams
 *   agents/network/sac_network.py:152-172      build_networks(): pi, squash, *action_max[0], Q(s,a), Q(s,pi), V
 *   agents/network/sac_network.py:174-232      qf (state NOT clipped) and vf (state clipped to state_min[0], state_max[0])
 *   agents/network/sac_network.py:234-307      policy net, log_std = -20 + 11*(tanh+1), reparameterised sample,
 *                                              gaussian_likelihood with +1e-6 in the divisor, tanh squash with
 *                                              clip_but_pass_gradient(1-pi^2,0,1)+1e-6
 * Quirks kept: Q9 -- logp_pi has shape [B] while q_pi, v are [B,1], so v_backup = q_pi - alpha*logp_pi broadcasts
 * to [B,B] and v_loss = 0.5*mean_{i,j}(q_pi[i] - alpha*logp[j] - v[i])^2 (V regresses onto q_pi[i] - alpha*mean(logp));
 * pi_loss = mean over the same [B,B] = alpha*mean(logp) - mean(q_pi); the state clip uses the SCALARS
 * state_min[0] / state_max[0] for every state dimension; mu and pi are scaled by action_max[0] with no log-det term;
 * all forward values come from the pre-update weights, pi-Adam then value-Adam (control dependency, :129-133);
 * Polyak is (1-tau)*target + tau*main over every main/target variable pair (:72-73).
 * The N(0,1) draw of tf.random_normal (:286) is an INPUT here (eps[B,A]) so that runs are comparable.
 * TensorFlow 1.15 arithmetic: "parity unpinned" (see ddpg_oracle.c header); cross-checked by tests/torch_ref_sac.py.
 *
 * Parameter blob (variable creation order under 'main'): pi: W1[S,L1a] b1 W2[L1a,L2a] b2 Wm[L2a,A] bm Ws[L2a,A] bs |
 *   qf: W1[S,L1c] b1 W2[L1c+A,L2c] b2 W3[L2c] b3 | vf: W1[S,L1c] b1 W2[L1c,L2c] b2 W3[L2c] b3
 */
#include "ftz.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int S, A, L1A, L2A, L1C, L2C;
} sdims_t;

typedef struct {
    int pW1, pb1, pW2, pb2, pWm, pbm, pWs, pbs;
    int qW1, qb1, qW2, qb2, qW3, qb3;
    int vW1, vb1, vW2, vb2, vW3, vb3;
    int Ppi, P;
} soffs_t;

static soffs_t soffsets(sdims_t d) {
    soffs_t o;
    int p = 0;
    o.pW1 = p; p += d.S * d.L1A;   o.pb1 = p; p += d.L1A;
    o.pW2 = p; p += d.L1A * d.L2A; o.pb2 = p; p += d.L2A;
    o.pWm = p; p += d.L2A * d.A;   o.pbm = p; p += d.A;
    o.pWs = p; p += d.L2A * d.A;   o.pbs = p; p += d.A;
    o.Ppi = p;
    o.qW1 = p; p += d.S * d.L1C;   o.qb1 = p; p += d.L1C;
    o.qW2 = p; p += (d.L1C + d.A) * d.L2C; o.qb2 = p; p += d.L2C;
    o.qW3 = p; p += d.L2C;         o.qb3 = p; p += 1;
    o.vW1 = p; p += d.S * d.L1C;   o.vb1 = p; p += d.L1C;
    o.vW2 = p; p += d.L1C * d.L2C; o.vb2 = p; p += d.L2C;
    o.vW3 = p; p += d.L2C;         o.vb3 = p; p += 1;
    o.P = p;
    return o;
}

int sac_oracle_param_count(int S, int A, int L1A, int L2A, int L1C, int L2C) {
    sdims_t d = {S, A, L1A, L2A, L1C, L2C};
    return soffsets(d).P;
}

static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

static void dense(const float* x, int B, int K, const float* W, const float* b, int N, float* y, int relu) {
    for (int i = 0; i < B; i++)
        for (int n = 0; n < N; n++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) acc += x[i * K + k] * W[k * N + n];
            acc += b[n];
            y[i * N + n] = relu ? (acc > 0.0f ? acc : 0.0f) : acc;
        }
}

/* dX = (H > 0) * (dY . W^T) for W[K,N]; accumulate gW += X^T dY, gb += sum dY */
static void dense_bwd(const float* x, const float* dy, int B, int K, const float* W, int N, float* gW, float* gb,
                      float* dx /* may be NULL */, const float* hmask /* relu output of the layer feeding x, or NULL */) {
    for (int i = 0; i < B; i++)
        for (int n = 0; n < N; n++) {
            const float t = dy[i * N + n];
            gb[n] += t;
            for (int k = 0; k < K; k++) gW[k * N + n] += x[i * K + k] * t;
        }
    if (dx)
        for (int i = 0; i < B; i++)
            for (int k = 0; k < K; k++) {
                float acc = 0.0f;
                for (int n = 0; n < N; n++) acc += dy[i * N + n] * W[k * N + n];
                dx[i * K + k] = (hmask == NULL || hmask[i * K + k] > 0.0f) ? acc : 0.0f;
            }
}

static void adam_range(float* var, float* m, float* v, const float* g, int n, float lr, float b1p, float b2p) {
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    for (int i = 0; i < n; i++) {
        m[i] += (g[i] - m[i]) * (1.0f - 0.9f);
        v[i] += (g[i] * g[i] - v[i]) * (1.0f - 0.999f);
        var[i] -= (m[i] * alpha) / (sqrtf(v[i]) + 1e-8f);
    }
}

/* Q(s,a) with the qf block of theta; returns hidden activations for backprop */
static void qf_forward(const float* th, soffs_t o, sdims_t d, const float* s, const float* a, int B, float* h1,
                       float* h2, float* q) {
    dense(s, B, d.S, th + o.qW1, th + o.qb1, d.L1C, h1, 1);
    for (int i = 0; i < B; i++)
        for (int n = 0; n < d.L2C; n++) {
            float acc = 0.0f;
            for (int k = 0; k < d.L1C; k++) acc += h1[i * d.L1C + k] * th[o.qW2 + k * d.L2C + n];
            for (int j = 0; j < d.A; j++) acc += a[i * d.A + j] * th[o.qW2 + (d.L1C + j) * d.L2C + n];
            acc += th[o.qb2 + n];
            h2[i * d.L2C + n] = acc > 0.0f ? acc : 0.0f;
        }
    dense(h2, B, d.L2C, th + o.qW3, th + o.qb3, 1, q, 0);
}

static void vf_forward(const float* th, soffs_t o, sdims_t d, const float* xc, int B, float* h1, float* h2, float* v) {
    dense(xc, B, d.S, th + o.vW1, th + o.vb1, d.L1C, h1, 1);
    dense(h1, B, d.L1C, th + o.vW2, th + o.vb2, d.L2C, h2, 1);
    dense(h2, B, d.L2C, th + o.vW3, th + o.vb3, 1, v, 0);
}

/* policy forward: mean action (mode 0) or sample with injected eps (mode 1); outputs scaled by amax0 */
void sac_oracle_act(int S, int A, int L1A, int L2A, int L1C, int L2C, const float* theta, const float* states, int B,
                    int do_clip, float smin0, float smax0, float amax0, const float* eps, float* out) {
    sdims_t d = {S, A, L1A, L2A, L1C, L2C};
    soffs_t o = soffsets(d);
    float* xc = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * L1A);
    float* h2 = malloc(sizeof(float) * B * L2A);
    float* mu = malloc(sizeof(float) * B * A);
    float* ls = malloc(sizeof(float) * B * A);
    for (int i = 0; i < B * S; i++) xc[i] = do_clip ? clipf(states[i], smin0, smax0) : states[i];
    dense(xc, B, S, theta + o.pW1, theta + o.pb1, L1A, h1, 1);
    dense(h1, B, L1A, theta + o.pW2, theta + o.pb2, L2A, h2, 1);
    dense(h2, B, L2A, theta + o.pWm, theta + o.pbm, A, mu, 0);
    dense(h2, B, L2A, theta + o.pWs, theta + o.pbs, A, ls, 0);
    for (int i = 0; i < B * A; i++) {
        float u = mu[i];
        if (eps) {
            const float log_std = -20.0f + 0.5f * (2.0f - (-20.0f)) * (tanhf(ls[i]) + 1.0f);
            u = mu[i] + eps[i] * expf(log_std);
        }
        out[i] = tanhf(u) * amax0;
    }
    free(xc); free(h1); free(h2); free(mu); free(ls);
}

/*
 * One update_network() + update_target_network() (agents/SoftActorCritic.py:113-126).
 *   theta, theta_t [P]; m, v [P] Adam slots (pi optimizer owns [0,Ppi), value optimizer [Ppi,P));
 *   pw[4] = {pi b1^t, pi b2^t, value b1^t, value b2^t};  r, gam fp32 (placeholders are fp32 here)
 *   taps (may be NULL): q[B], v[B], logp[B], q_pi[B], losses[3] = {pi_loss, q_loss, v_loss}, grads[P]
 */
static void sac_oracle_update_impl(int S, int A, int L1A, int L2A, int L1C, int L2C, int B, float* theta, float* theta_t, float* m,
                       float* v, float* pw, const float* s, const float* a, const float* r, const float* s2,
                       const float* gam, const float* eps, float pi_lr, float qv_lr, float alpha_ent, float tau,
                       int do_clip, float smin0, float smax0, float amax0, float* tap_q, float* tap_v,
                       float* tap_logp, float* tap_qpi, float* tap_loss, float* tap_g) {
    sdims_t d = {S, A, L1A, L2A, L1C, L2C};
    soffs_t o = soffsets(d);
    const int P = o.P;
    const float EPS = 1e-6f, LOG2PI = (float)log(2.0 * 3.14159265358979323846);
    float* xc = malloc(sizeof(float) * B * S);
    float* x2c = malloc(sizeof(float) * B * S);
    float* ph1 = malloc(sizeof(float) * B * L1A);
    float* ph2 = malloc(sizeof(float) * B * L2A);
    float* mu = malloc(sizeof(float) * B * A);
    float* lsp = malloc(sizeof(float) * B * A);
    float* t_ = malloc(sizeof(float) * B * A);
    float* std_ = malloc(sizeof(float) * B * A);
    float* pit = malloc(sizeof(float) * B * A);
    float* api = malloc(sizeof(float) * B * A);
    float* logp = malloc(sizeof(float) * B);
    float* qh1 = malloc(sizeof(float) * B * L1C);
    float* qh2 = malloc(sizeof(float) * B * L2C);
    float* qh2p = malloc(sizeof(float) * B * L2C);
    float* q = malloc(sizeof(float) * B);
    float* qpi = malloc(sizeof(float) * B);
    float* vh1 = malloc(sizeof(float) * B * L1C);
    float* vh2 = malloc(sizeof(float) * B * L2C);
    float* vv = malloc(sizeof(float) * B);
    float* th1 = malloc(sizeof(float) * B * L1C);
    float* th2 = malloc(sizeof(float) * B * L2C);
    float* vt = malloc(sizeof(float) * B);
    float* g = calloc(P, sizeof(float));
    float* d1 = malloc(sizeof(float) * B * (L2A > L2C ? L2A : L2C));
    float* d0 = malloc(sizeof(float) * B * (L1A > L1C ? L1A : L1C));
    float* dmu = malloc(sizeof(float) * B * A);
    float* dls = malloc(sizeof(float) * B * A);
    float* dout = malloc(sizeof(float) * B);

    for (int i = 0; i < B * S; i++) {
        xc[i] = do_clip ? clipf(s[i], smin0, smax0) : s[i];
        x2c[i] = do_clip ? clipf(s2[i], smin0, smax0) : s2[i];
    }
    /* ---- forward, all with the pre-update weights ---- */
    dense(xc, B, S, theta + o.pW1, theta + o.pb1, L1A, ph1, 1);
    dense(ph1, B, L1A, theta + o.pW2, theta + o.pb2, L2A, ph2, 1);
    dense(ph2, B, L2A, theta + o.pWm, theta + o.pbm, A, mu, 0);
    dense(ph2, B, L2A, theta + o.pWs, theta + o.pbs, A, lsp, 0);
    for (int i = 0; i < B; i++) {
        float lp = 0.0f;
        for (int j = 0; j < A; j++) {
            const int k = i * A + j;
            t_[k] = tanhf(lsp[k]);
            const float log_std = -20.0f + 0.5f * (2.0f - (-20.0f)) * (t_[k] + 1.0f);
            std_[k] = expf(log_std);
            const float u = mu[k] + eps[k] * std_[k];
            const float z = (u - mu[k]) / (std_[k] + EPS);
            lp += -0.5f * (z * z + 2.0f * log_std + LOG2PI);
            pit[k] = tanhf(u);
            api[k] = pit[k] * amax0;
        }
        for (int j = 0; j < A; j++) {
            const float om = 1.0f - pit[i * A + j] * pit[i * A + j];
            lp -= logf(clipf(om, 0.0f, 1.0f) + 1e-6f);
        }
        logp[i] = lp;
    }
    qf_forward(theta, o, d, s, a, B, qh1, qh2, q);          /* Q sees the RAW state (sac_network.py:176) */
    qf_forward(theta, o, d, s, api, B, qh1, qh2p, qpi);
    vf_forward(theta, o, d, xc, B, vh1, vh2, vv);
    vf_forward(theta_t, o, d, x2c, B, th1, th2, vt);         /* only the target V is ever read */
    float mean_logp = 0.0f, mean_qpi = 0.0f;
    for (int i = 0; i < B; i++) { mean_logp += logp[i]; mean_qpi += qpi[i]; }
    mean_logp /= (float)B; mean_qpi /= (float)B;
    if (tap_q) memcpy(tap_q, q, sizeof(float) * B);
    if (tap_v) memcpy(tap_v, vv, sizeof(float) * B);
    if (tap_logp) memcpy(tap_logp, logp, sizeof(float) * B);
    if (tap_qpi) memcpy(tap_qpi, qpi, sizeof(float) * B);
    if (tap_loss) {
        float ql = 0.0f, vl = 0.0f;
        for (int i = 0; i < B; i++) {
            const float e = (r[i] + gam[i] * vt[i]) - q[i];
            ql += e * e;
            for (int j = 0; j < B; j++) {
                const float f = qpi[i] - alpha_ent * logp[j] - vv[i];
                vl += f * f;
            }
        }
        tap_loss[0] = alpha_ent * mean_logp - mean_qpi;
        tap_loss[1] = 0.5f * ql / (float)B;
        tap_loss[2] = 0.5f * vl / ((float)B * (float)B);
    }

    /* ---- pi gradient: d/d(pi params) of alpha*mean(logp) - mean(Q(s, pi)) ---- */
    /* dQ/da at a = api through the qf block (weights fixed) */
    for (int i = 0; i < B; i++)
        for (int j = 0; j < A; j++) {
            float ga = 0.0f;
            for (int n = 0; n < L2C; n++)
                if (qh2p[i * L2C + n] > 0.0f) ga += theta[o.qW3 + n] * theta[o.qW2 + (L1C + j) * L2C + n];
            const int k = i * A + j;
            const float om = 1.0f - pit[k] * pit[k];
            const float dlogp_dpit = 2.0f * pit[k] / (clipf(om, 0.0f, 1.0f) + 1e-6f);     /* -d corr / d pi_t */
            const float dL_dpit = (-1.0f / (float)B) * ga * amax0 + (alpha_ent / (float)B) * dlogp_dpit;
            const float dL_du = dL_dpit * om;
            const float sd = std_[k], e = eps[k];
            const float z = e * sd / (sd + EPS);
            const float dz_dls = e * sd * EPS / ((sd + EPS) * (sd + EPS));
            const float dL_dlogstd = dL_du * e * sd + (alpha_ent / (float)B) * (-z * dz_dls - 1.0f);
            dmu[k] = dL_du;
            dls[k] = dL_dlogstd * (0.5f * (2.0f - (-20.0f))) * (1.0f - t_[k] * t_[k]);
        }
    /* heads -> ph2 */
    for (int i = 0; i < B * L2A; i++) d1[i] = 0.0f;
    {
        float* tmp = malloc(sizeof(float) * B * L2A);
        dense_bwd(ph2, dmu, B, L2A, theta + o.pWm, A, g + o.pWm, g + o.pbm, tmp, ph2);
        for (int i = 0; i < B * L2A; i++) d1[i] += tmp[i];
        dense_bwd(ph2, dls, B, L2A, theta + o.pWs, A, g + o.pWs, g + o.pbs, tmp, ph2);
        for (int i = 0; i < B * L2A; i++) d1[i] += tmp[i];
        free(tmp);
    }
    dense_bwd(ph1, d1, B, L1A, theta + o.pW2, L2A, g + o.pW2, g + o.pb2, d0, ph1);
    dense_bwd(xc, d0, B, S, theta + o.pW1, L1A, g + o.pW1, g + o.pb1, NULL, NULL);

    /* ---- value gradients (forward values from the pre-update weights) ---- */
    /* q_loss = 0.5 mean((r + g*v_targ - q)^2) */
    for (int i = 0; i < B; i++) dout[i] = -((r[i] + gam[i] * vt[i]) - q[i]) / (float)B;
    for (int i = 0; i < B; i++) {
        g[o.qb3] += dout[i];
        for (int n = 0; n < L2C; n++) {
            g[o.qW3 + n] += qh2[i * L2C + n] * dout[i];
            d1[i * L2C + n] = qh2[i * L2C + n] > 0.0f ? dout[i] * theta[o.qW3 + n] : 0.0f;
        }
    }
    for (int i = 0; i < B; i++)
        for (int n = 0; n < L2C; n++) {
            const float t = d1[i * L2C + n];
            g[o.qb2 + n] += t;
            for (int k = 0; k < L1C; k++) g[o.qW2 + k * L2C + n] += qh1[i * L1C + k] * t;
            for (int j = 0; j < A; j++) g[o.qW2 + (L1C + j) * L2C + n] += a[i * A + j] * t;
        }
    for (int i = 0; i < B; i++)
        for (int k = 0; k < L1C; k++) {
            float acc = 0.0f;
            for (int n = 0; n < L2C; n++) acc += d1[i * L2C + n] * theta[o.qW2 + k * L2C + n];
            d0[i * L1C + k] = qh1[i * L1C + k] > 0.0f ? acc : 0.0f;
        }
    dense_bwd(s, d0, B, S, theta + o.qW1, L1C, g + o.qW1, g + o.qb1, NULL, NULL);
    /* v_loss = 0.5 mean_{i,j}(q_pi[i] - alpha*logp[j] - v[i])^2   (Q9) */
    for (int i = 0; i < B; i++) dout[i] = -(qpi[i] - alpha_ent * mean_logp - vv[i]) / (float)B;
    for (int i = 0; i < B; i++) {
        g[o.vb3] += dout[i];
        for (int n = 0; n < L2C; n++) {
            g[o.vW3 + n] += vh2[i * L2C + n] * dout[i];
            d1[i * L2C + n] = vh2[i * L2C + n] > 0.0f ? dout[i] * theta[o.vW3 + n] : 0.0f;
        }
    }
    dense_bwd(vh1, d1, B, L1C, theta + o.vW2, L2C, g + o.vW2, g + o.vb2, d0, vh1);
    dense_bwd(xc, d0, B, S, theta + o.vW1, L1C, g + o.vW1, g + o.vb1, NULL, NULL);
    if (tap_g) memcpy(tap_g, g, sizeof(float) * P);

    /* pi-Adam, then value-Adam (sac_network.py:124-133) */
    adam_range(theta, m, v, g, o.Ppi, pi_lr, pw[0], pw[1]);
    pw[0] *= 0.9f; pw[1] *= 0.999f;
    adam_range(theta + o.Ppi, m + o.Ppi, v + o.Ppi, g + o.Ppi, P - o.Ppi, qv_lr, pw[2], pw[3]);
    pw[2] *= 0.9f; pw[3] *= 0.999f;
    /* Polyak over every main/target pair (sac_network.py:72-73): (1-tau)*target + tau*main */
    for (int i = 0; i < P; i++) theta_t[i] = (1.0f - tau) * theta_t[i] + tau * theta[i];

    free(xc); free(x2c); free(ph1); free(ph2); free(mu); free(lsp); free(t_); free(std_); free(pit); free(api);
    free(logp); free(qh1); free(qh2); free(qh2p); free(q); free(qpi); free(vh1); free(vh2); free(vv); free(th1);
    free(th2); free(vt); free(g); free(d1); free(d0); free(dmu); free(dls); free(dout);
}

void sac_oracle_update(int S, int A, int L1A, int L2A, int L1C, int L2C, int B, float* theta, float* theta_t, float* m,
                       float* v, float* pw, const float* s, const float* a, const float* r, const float* s2,
                       const float* gam, const float* eps, float pi_lr, float qv_lr, float alpha_ent, float tau,
                       int do_clip, float smin0, float smax0, float amax0, float* tap_q, float* tap_v,
                       float* tap_logp, float* tap_qpi, float* tap_loss, float* tap_g) {
    const unsigned csr = oracle_ftz_on();       /* TF-1.15 CPU arithmetic: denormals flushed (oracle/ftz.h) */
    sac_oracle_update_impl(S, A, L1A, L2A, L1C, L2C, B, theta, theta_t, m, v, pw, s, a, r, s2, gam, eps, pi_lr, qv_lr, alpha_ent, tau, do_clip, smin0, smax0, amax0, tap_q, tap_v, tap_logp, tap_qpi, tap_loss, tap_g);
    oracle_ftz_restore(csr);
}
