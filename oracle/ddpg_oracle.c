/*
 * oracle/ddpg_oracle.c -- CPU restatement of the reference's DDPG ("hydra") hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rlcontrol_amd/ (the product) may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported baseline.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   agents/DDPG.py:74-95                      update_network(): the 7-step order
 *   agents/network/hydra_ddpg_network.py:97-142   network(): shared 3->H1 trunk, actor and critic heads
 *   agents/network/hydra_ddpg_network.py:78-95    build_network(): clip(normalize(x)) (quirk Q6), a_max scale
 *   agents/network/hydra_ddpg_network.py:29,36-37,68,71-72,75  Polyak, actor grads (batch SUM, unscaled
 *                                                 tanh output, quirk Q3), two Adams (Q1), MSE loss, dQ/da
 *   agents/DDPG.py:79-84                      TD target formed in float64 then cast to fp32 (Q5)
 *
 * The arithmetic itself lives in a third-party dependency that is absent from /root/reference:
 * tensorflow_cpu==1.15.0 (requirements.txt:2).  Its published semantics restated here:
 *   - tf.contrib.layers.fully_connected: y = x.W + b with W[in,out]
 *   - tf.train.AdamOptimizer (core/kernels/training_ops.cc ApplyAdam, non-Nesterov):
 *        alpha = lr*sqrt(1-b2p)/(1-b1p);  m += (g-m)*(1-b1);  v += (g*g-v)*(1-b2);
 *        var -= (m*alpha)/(sqrt(v)+eps);   afterwards b1p*=b1, b2p*=b2        (quirk Q2)
 *   - tf.gradients(ys, xs, grad_ys): sum over the batch of grad_ys * d ys/d xs
 *   - minimize()/apply_gradients() skip variables whose gradient is None
 * PARITY STATUS: the numpy-side behaviour (sampling, OU noise, gating) is pinned by golden vectors
 * generated from the reference itself (tests/golden/make_golden.py); the network arithmetic is
 * "parity unpinned" at the TensorFlow boundary (no reference test or fixture holds Q-values or
 * gradients) and is pinned instead by a second, independent float64 autograd restatement
 * (tests/torch_ref.py) that this file must agree with.
 *
 * Parameter blob layout (variable creation order, hydra_ddpg_network.py:100-140):
 *   W1[S,H1] b1[H1] | Wa2[H1,HA] ba2[HA] Wa3[HA,A] ba3[A] | Wc2[H1+A,HC] bc2[HC] Wc3[HC,1] bc3[1]
 * The critic's concat puts the action LAST (hydra_ddpg_network.py:128), so rows H1..H1+A-1 of Wc2
 * multiply the action.
 */
#include "ftz.h"
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int S, A, H1, HA, HC;
} dims_t;

typedef struct {
    int oW1, ob1, oWa2, oba2, oWa3, oba3, oWc2, obc2, oWc3, obc3, P;
} offs_t;

static offs_t offsets(dims_t d) {
    offs_t o;
    int p = 0;
    o.oW1 = p;  p += d.S * d.H1;
    o.ob1 = p;  p += d.H1;
    o.oWa2 = p; p += d.H1 * d.HA;
    o.oba2 = p; p += d.HA;
    o.oWa3 = p; p += d.HA * d.A;
    o.oba3 = p; p += d.A;
    o.oWc2 = p; p += (d.H1 + d.A) * d.HC;
    o.obc2 = p; p += d.HC;
    o.oWc3 = p; p += d.HC;
    o.obc3 = p; p += 1;
    o.P = p;
    return o;
}

int ddpg_oracle_param_count(int S, int A, int H1, int HA, int HC) {
    dims_t d = {S, A, H1, HA, HC};
    return offsets(d).P;
}

static float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* x[B,S] -> clipped copy (hydra_ddpg_network.py:86-87; RunningMeanStd is mean 0 / var 1: Q6) */
static void clip_state(const float* s, float* x, int B, int S, int do_clip, const float* smin, const float* smax) {
    for (int b = 0; b < B; b++)
        for (int i = 0; i < S; i++) {
            float v = s[b * S + i];
            if (do_clip) v = clipf((v - 0.0f) / 1.0f, smin[i], smax[i]);
            x[b * S + i] = v;
        }
}

/* y[B,N] = act(x[B,K].W[K,N] + b[N]);  act: 0 none, 1 relu, 2 tanh */
static void dense(const float* x, int B, int K, const float* W, const float* bias, int N, float* y, int act) {
    for (int b = 0; b < B; b++)
        for (int n = 0; n < N; n++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) acc += x[b * K + k] * W[k * N + n];
            acc += bias[n];
            if (act == 1) acc = acc > 0.0f ? acc : 0.0f;
            else if (act == 2) acc = tanhf(acc);
            y[b * N + n] = acc;
        }
}

/* actor: mu = tanh(relu(h1.Wa2+ba2).Wa3+ba3) (unscaled); h2 returned for backprop */
static void actor_head(const float* th, offs_t o, dims_t d, const float* h1, int B, float* h2, float* mu) {
    dense(h1, B, d.H1, th + o.oWa2, th + o.oba2, d.HA, h2, 1);
    dense(h2, B, d.HA, th + o.oWa3, th + o.oba3, d.A, mu, 2);
}

/* critic: q = relu([h1,a].Wc2+bc2).Wc3+bc3 ; g2 returned for backprop */
static void critic_head(const float* th, offs_t o, dims_t d, const float* h1, const float* a, int B,
                        float* g2, float* q) {
    const float* Wc2 = th + o.oWc2;
    for (int b = 0; b < B; b++)
        for (int n = 0; n < d.HC; n++) {
            float acc = 0.0f;
            for (int k = 0; k < d.H1; k++) acc += h1[b * d.H1 + k] * Wc2[k * d.HC + n];
            for (int j = 0; j < d.A; j++) acc += a[b * d.A + j] * Wc2[(d.H1 + j) * d.HC + n];
            acc += th[o.obc2 + n];
            g2[b * d.HC + n] = acc > 0.0f ? acc : 0.0f;
        }
    dense(g2, B, d.HC, th + o.oWc3, th + o.obc3, 1, q, 0);
}

/* TF-1.15 ApplyAdam on a contiguous range, with the optimizer's current beta powers */
static void adam_range(float* var, float* m, float* v, const float* g, int n, float lr, float b1p, float b2p) {
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    for (int i = 0; i < n; i++) {
        m[i] += (g[i] - m[i]) * (1.0f - beta1);
        v[i] += (g[i] * g[i] - v[i]) * (1.0f - beta2);
        var[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
    }
}

/* Greedy scaled action for a batch of states (agents/DDPG.py:36; hydra_ddpg_network.py:162-171) */
void ddpg_oracle_act(int S, int A, int H1, int HA, int HC, const float* theta, const float* states, int B,
                     int do_clip, const float* smin, const float* smax, const float* amax, float* out) {
    dims_t d = {S, A, H1, HA, HC};
    offs_t o = offsets(d);
    float* x = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * H1);
    float* h2 = malloc(sizeof(float) * B * HA);
    float* mu = malloc(sizeof(float) * B * A);
    clip_state(states, x, B, S, do_clip, smin, smax);
    dense(x, B, S, theta + o.oW1, theta + o.ob1, H1, h1, 1);
    actor_head(theta, o, d, h1, B, h2, mu);
    for (int b = 0; b < B; b++)
        for (int j = 0; j < A; j++) out[b * A + j] = mu[b * A + j] * amax[j];
    free(x); free(h1); free(h2); free(mu);
}

/* Q(s,a) for a batch (hydra_ddpg_network.py:183-193) */
void ddpg_oracle_qval(int S, int A, int H1, int HA, int HC, const float* theta, const float* states,
                      const float* actions, int B, int do_clip, const float* smin, const float* smax, float* out) {
    dims_t d = {S, A, H1, HA, HC};
    offs_t o = offsets(d);
    float* x = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * H1);
    float* g2 = malloc(sizeof(float) * B * HC);
    clip_state(states, x, B, S, do_clip, smin, smax);
    dense(x, B, S, theta + o.oW1, theta + o.ob1, H1, h1, 1);
    critic_head(theta, o, d, h1, actions, B, g2, out);
    free(x); free(h1); free(g2);
}

/*
 * One update_network() call (agents/DDPG.py:74-95).
 *   theta, theta_t            [P]   online / target parameters          (in/out)
 *   m_a, v_a, m_c, v_c        [P]   Adam slots of the actor / critic optimizers, blob-aligned (in/out)
 *   pw                        [4]   beta1^t, beta2^t of actor opt, then of critic opt           (in/out)
 *   s[B,S] a[B,A] s2[B,S]     fp32 (what the fp32 placeholders receive); r[B], gam[B] float64 (Q5)
 *   taps (may be NULL): q_pre[B] critic output before its step (train_critic's first fetch),
 *                       y[B] TD target, a_out[B,A] scaled actor output, dqda[B,A], grads_c[P], grads_a[P]
 */
static void ddpg_oracle_update_impl(int S, int A, int H1, int HA, int HC, int B,
                        float* theta, float* theta_t, float* m_a, float* v_a, float* m_c, float* v_c, float* pw,
                        const float* s, const float* a, const double* r, const float* s2, const double* gam,
                        float actor_lr, float critic_lr, float tau,
                        int do_clip, const float* smin, const float* smax, const float* amax,
                        float* tap_q, float* tap_y, float* tap_aout, float* tap_dqda,
                        float* tap_gc, float* tap_ga) {
    dims_t d = {S, A, H1, HA, HC};
    offs_t o = offsets(d);
    const int P = o.P;
    float* x = malloc(sizeof(float) * B * S);
    float* x2 = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * H1);
    float* h2 = malloc(sizeof(float) * B * HA);
    float* g2 = malloc(sizeof(float) * B * HC);
    float* mu = malloc(sizeof(float) * B * A);
    float* aout = malloc(sizeof(float) * B * A);
    float* q = malloc(sizeof(float) * B);
    float* y = malloc(sizeof(float) * B);
    float* dq = malloc(sizeof(float) * B);
    float* dg2 = malloc(sizeof(float) * B * HC);
    float* dh2 = malloc(sizeof(float) * B * HA);
    float* dh1 = malloc(sizeof(float) * B * H1);
    float* dqda = malloc(sizeof(float) * B * A);
    float* dz = malloc(sizeof(float) * B * A);
    float* g = calloc(P, sizeof(float));

    clip_state(s, x, B, S, do_clip, smin, smax);
    clip_state(s2, x2, B, S, do_clip, smin, smax);

    /* steps 1-2: target actor then target critic on s' (DDPG.py:77) */
    dense(x2, B, S, theta_t + o.oW1, theta_t + o.ob1, H1, h1, 1);
    actor_head(theta_t, o, d, h1, B, h2, mu);
    for (int i = 0; i < B * A; i++) aout[i] = mu[i] * amax[i % A];
    critic_head(theta_t, o, d, h1, aout, B, g2, q);
    /* TD target in float64, then the fp32 placeholder cast (DDPG.py:80-84) */
    for (int b = 0; b < B; b++) y[b] = (float)(r[b] + gam[b] * (double)q[b]);
    if (tap_y) memcpy(tap_y, y, sizeof(float) * B);

    /* step 3: critic step (hydra_ddpg_network.py:71-72,153-160) */
    dense(x, B, S, theta + o.oW1, theta + o.ob1, H1, h1, 1);
    critic_head(theta, o, d, h1, a, B, g2, q);
    if (tap_q) memcpy(tap_q, q, sizeof(float) * B);
    /* L = mean (y-q)^2  ->  dL/dq = 2(q-y)/B */
    for (int b = 0; b < B; b++) dq[b] = 2.0f * (q[b] - y[b]) / (float)B;
    memset(g, 0, sizeof(float) * P);
    for (int b = 0; b < B; b++) {
        g[o.obc3] += dq[b];
        for (int n = 0; n < HC; n++) {
            g[o.oWc3 + n] += g2[b * HC + n] * dq[b];
            dg2[b * HC + n] = g2[b * HC + n] > 0.0f ? dq[b] * theta[o.oWc3 + n] : 0.0f;
        }
    }
    for (int b = 0; b < B; b++)
        for (int n = 0; n < HC; n++) {
            float t = dg2[b * HC + n];
            g[o.obc2 + n] += t;
            for (int k = 0; k < H1; k++) g[o.oWc2 + k * HC + n] += h1[b * H1 + k] * t;
            for (int j = 0; j < A; j++) g[o.oWc2 + (H1 + j) * HC + n] += a[b * A + j] * t;
        }
    for (int b = 0; b < B; b++)
        for (int k = 0; k < H1; k++) {
            float acc = 0.0f;
            for (int n = 0; n < HC; n++) acc += dg2[b * HC + n] * theta[o.oWc2 + k * HC + n];
            acc = h1[b * H1 + k] > 0.0f ? acc : 0.0f;
            dh1[b * H1 + k] = acc;
            g[o.ob1 + k] += acc;
            for (int i = 0; i < S; i++) g[o.oW1 + i * H1 + k] += x[b * S + i] * acc;
        }
    if (tap_gc) memcpy(tap_gc, g, sizeof(float) * P);
    /* Adam_c over the variables with a non-None gradient: W1,b1 and the critic branch (Q1) */
    adam_range(theta + o.oW1, m_c + o.oW1, v_c + o.oW1, g + o.oW1, S * H1 + H1, critic_lr, pw[2], pw[3]);
    adam_range(theta + o.oWc2, m_c + o.oWc2, v_c + o.oWc2, g + o.oWc2, P - o.oWc2, critic_lr, pw[2], pw[3]);
    pw[2] *= 0.9f;
    pw[3] *= 0.999f;

    /* step 4: actor forward with the UPDATED trunk (DDPG.py:90) */
    dense(x, B, S, theta + o.oW1, theta + o.ob1, H1, h1, 1);
    actor_head(theta, o, d, h1, B, h2, mu);
    for (int i = 0; i < B * A; i++) aout[i] = mu[i] * amax[i % A];
    if (tap_aout) memcpy(tap_aout, aout, sizeof(float) * B * A);
    /* step 5: dQ/da at the scaled action, updated critic (DDPG.py:91; hydra_ddpg_network.py:75) */
    critic_head(theta, o, d, h1, aout, B, g2, q);
    for (int b = 0; b < B; b++)
        for (int j = 0; j < A; j++) {
            float acc = 0.0f;
            for (int n = 0; n < HC; n++)
                if (g2[b * HC + n] > 0.0f) acc += theta[o.oWc3 + n] * theta[o.oWc2 + (H1 + j) * HC + n];
            dqda[b * A + j] = acc;
        }
    if (tap_dqda) memcpy(tap_dqda, dqda, sizeof(float) * B * A);
    /* step 6: actor step.  grad_ys = -dQ/da on the UNSCALED tanh output, summed over the batch (Q3) */
    memset(g, 0, sizeof(float) * P);
    for (int i = 0; i < B * A; i++) dz[i] = -dqda[i] * (1.0f - mu[i] * mu[i]);
    for (int b = 0; b < B; b++) {
        for (int j = 0; j < A; j++) g[o.oba3 + j] += dz[b * A + j];
        for (int n = 0; n < HA; n++) {
            float acc = 0.0f;
            for (int j = 0; j < A; j++) {
                g[o.oWa3 + n * A + j] += h2[b * HA + n] * dz[b * A + j];
                acc += dz[b * A + j] * theta[o.oWa3 + n * A + j];
            }
            dh2[b * HA + n] = h2[b * HA + n] > 0.0f ? acc : 0.0f;
        }
    }
    for (int b = 0; b < B; b++)
        for (int n = 0; n < HA; n++) {
            float t = dh2[b * HA + n];
            g[o.oba2 + n] += t;
            for (int k = 0; k < H1; k++) g[o.oWa2 + k * HA + n] += h1[b * H1 + k] * t;
        }
    for (int b = 0; b < B; b++)
        for (int k = 0; k < H1; k++) {
            float acc = 0.0f;
            for (int n = 0; n < HA; n++) acc += dh2[b * HA + n] * theta[o.oWa2 + k * HA + n];
            acc = h1[b * H1 + k] > 0.0f ? acc : 0.0f;
            g[o.ob1 + k] += acc;
            for (int i = 0; i < S; i++) g[o.oW1 + i * H1 + k] += x[b * S + i] * acc;
        }
    if (tap_ga) memcpy(tap_ga, g, sizeof(float) * P);
    adam_range(theta, m_a, v_a, g, o.oWc2, actor_lr, pw[0], pw[1]);
    pw[0] *= 0.9f;
    pw[1] *= 0.999f;

    /* step 7: Polyak over all ten tensors (hydra_ddpg_network.py:29) */
    for (int i = 0; i < P; i++) theta_t[i] += tau * (theta[i] - theta_t[i]);

    free(x); free(x2); free(h1); free(h2); free(g2); free(mu); free(aout); free(q); free(y); free(dq);
    free(dg2); free(dh2); free(dh1); free(dqda); free(dz); free(g);
}

void ddpg_oracle_update(int S, int A, int H1, int HA, int HC, int B,
                        float* theta, float* theta_t, float* m_a, float* v_a, float* m_c, float* v_c, float* pw,
                        const float* s, const float* a, const double* r, const float* s2, const double* gam,
                        float actor_lr, float critic_lr, float tau,
                        int do_clip, const float* smin, const float* smax, const float* amax,
                        float* tap_q, float* tap_y, float* tap_aout, float* tap_dqda,
                        float* tap_gc, float* tap_ga) {
    const unsigned csr = oracle_ftz_on();       /* TF-1.15 CPU arithmetic: denormals flushed (oracle/ftz.h) */
    ddpg_oracle_update_impl(S, A, H1, HA, HC, B, theta, theta_t, m_a, v_a, m_c, v_c, pw, s, a, r, s2, gam, actor_lr, critic_lr, tau, do_clip, smin, smax, amax, tap_q, tap_y, tap_aout, tap_dqda, tap_gc, tap_ga);
    oracle_ftz_restore(csr);
}
