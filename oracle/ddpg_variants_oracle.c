/*
 * oracle/ddpg_variants_oracle.c -- CPU restatement of the DDPG variants of SURVEY.md section 8(f) item 3:
 *   norm_type 'layer'  (agents/network/base_network.py:53-56: tf.contrib.layers.layer_norm(net, center=True,
 *                       scale=True, activation_fn=relu) after every hidden fully_connected), and
 *   separate actor / critic networks (agents/network/actor_network.py:73-96, critic_network.py:77-99: the same
 *                       layers as the hydra network without the shared first layer; commented out in agents/DDPG.py:8-9).
 *
 * TEST INFRASTRUCTURE ONLY (see ddpg_oracle.c).  With norm = 0 and separate = 0 this file must reproduce
 * ddpg_oracle.c bit for bit (tests/test_ddpg_variants.py), which ties it to the pinned restatement.
 *
 * Third-party semantics restated (tensorflow_cpu==1.15.0, absent -- parity unpinned at that boundary):
 *   tf.contrib.layers.layer_norm on [B, N]: mean / variance over the N features of each row (tf.nn.moments: the
 *   biased variance mean((x - mean)^2)), y = (x - mean) * rsqrt(var + 1e-12) * gamma + beta, variables created in the
 *   order beta (zeros) then gamma (ones), both trainable and Polyak-averaged like every other variable of the scope.
 *
 * Parameter blob, variable creation order ([..] only with layer norm):
 *   hydra     W1 b1 [l1b l1g] | Wa2 ba2 [l2b l2g] Wa3 ba3 | Wc2 bc2 [l3b l3g] Wc3 bc3
 *   separate  W1 b1 [l1b l1g]   Wa2 ba2 [l2b l2g] Wa3 ba3 | Wc1 bc1 [lcb lcg] Wc2 bc2 [l3b l3g] Wc3 bc3
 * The actor optimizer owns every tensor before the critic block (hydra: everything before Wc2, where the shared first
 * layer also gets the critic optimizer's step, quirk Q1); the critic optimizer owns the critic block.
 */
#include "ftz.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LN_EPS 1e-12f

typedef struct { int S, A, H1, HA, HC, norm, sep; } vd_t;
typedef struct {
    int W1, b1, l1b, l1g, Wa2, ba2, l2b, l2g, Wa3, ba3;
    int Wc1, bc1, lcb, lcg;            /* critic's first layer: aliases of W1.. when the first layer is shared */
    int Wc2, bc2, l3b, l3g, Wc3, bc3;
    int critic0;                       /* first offset of the critic optimizer's own block */
    int P;
} vo_t;

static vo_t voffsets(vd_t d) {
    vo_t o;
    int p = 0;
    o.W1 = p; p += d.S * d.H1;
    o.b1 = p; p += d.H1;
    o.l1b = p; if (d.norm) p += d.H1;
    o.l1g = p; if (d.norm) p += d.H1;
    o.Wa2 = p; p += d.H1 * d.HA;
    o.ba2 = p; p += d.HA;
    o.l2b = p; if (d.norm) p += d.HA;
    o.l2g = p; if (d.norm) p += d.HA;
    o.Wa3 = p; p += d.HA * d.A;
    o.ba3 = p; p += d.A;
    o.critic0 = p;
    if (d.sep) {
        o.Wc1 = p; p += d.S * d.H1;
        o.bc1 = p; p += d.H1;
        o.lcb = p; if (d.norm) p += d.H1;
        o.lcg = p; if (d.norm) p += d.H1;
    } else {
        o.Wc1 = o.W1; o.bc1 = o.b1; o.lcb = o.l1b; o.lcg = o.l1g;
    }
    o.Wc2 = p; p += (d.H1 + d.A) * d.HC;
    o.bc2 = p; p += d.HC;
    o.l3b = p; if (d.norm) p += d.HC;
    o.l3g = p; if (d.norm) p += d.HC;
    o.Wc3 = p; p += d.HC;
    o.bc3 = p; p += 1;
    o.P = p;
    return o;
}

int ddpg_variant_param_count(int S, int A, int H1, int HA, int HC, int norm, int sep) {
    vd_t d = {S, A, H1, HA, HC, norm, sep};
    return voffsets(d).P;
}

/* hidden layer: h = relu(LN(x1.W[0:K1] + x2.W[K1:K1+K2] + b)).  nhat / rstd (may be NULL without norm) keep what the
 * backward pass needs.  Same summation order as ddpg_oracle.c's dense(): k ascending, then the extra inputs, then b. */
static void layer_fwd(const float* x1, int K1, const float* x2, int K2, int B, const float* W, const float* b, int N,
                      int norm, const float* beta, const float* gamma, float* h, float* nhat, float* rstd) {
    for (int r = 0; r < B; r++) {
        float* z = h + (size_t)r * N;
        for (int n = 0; n < N; n++) {
            float acc = 0.0f;
            for (int k = 0; k < K1; k++) acc += x1[r * K1 + k] * W[k * N + n];
            for (int j = 0; j < K2; j++) acc += x2[r * K2 + j] * W[(K1 + j) * N + n];
            z[n] = acc + b[n];
        }
        if (norm) {
            float mean = 0.0f, var = 0.0f;
            for (int n = 0; n < N; n++) mean += z[n];
            mean /= (float)N;
            for (int n = 0; n < N; n++) var += (z[n] - mean) * (z[n] - mean);
            var /= (float)N;
            const float rs = 1.0f / sqrtf(var + LN_EPS);
            rstd[r] = rs;
            for (int n = 0; n < N; n++) {
                const float nh = (z[n] - mean) * rs;
                nhat[(size_t)r * N + n] = nh;
                z[n] = nh * gamma[n] + beta[n];
            }
        }
        for (int n = 0; n < N; n++) z[n] = z[n] > 0.0f ? z[n] : 0.0f;
    }
}

/* backward of layer_fwd.  dh: gradient w.r.t. the layer's output h (overwritten with dz, the gradient w.r.t. the
 * pre-norm activation).  Accumulates the gradients of W, b, beta, gamma into g; writes dx1 (may be NULL) and dx2. */
static void layer_bwd(const float* x1, int K1, const float* x2, int K2, int B, const float* W, int N, int norm,
                      const float* gamma, const float* h, const float* nhat, const float* rstd, float* dh,
                      float* g, int oW, int ob, int olb, int olg, float* dx1, float* dx2) {
    for (int r = 0; r < B; r++) {
        float* dz = dh + (size_t)r * N;
        for (int n = 0; n < N; n++) dz[n] = h[(size_t)r * N + n] > 0.0f ? dz[n] : 0.0f;
        if (norm) {
            float m1 = 0.0f, m2 = 0.0f;
            for (int n = 0; n < N; n++) {
                const float nh = nhat[(size_t)r * N + n];
                if (g) { g[olg + n] += dz[n] * nh; g[olb + n] += dz[n]; }
                dz[n] *= gamma[n];
                m1 += dz[n];
                m2 += dz[n] * nh;
            }
            m1 /= (float)N; m2 /= (float)N;
            for (int n = 0; n < N; n++) dz[n] = rstd[r] * (dz[n] - m1 - nhat[(size_t)r * N + n] * m2);
        }
        for (int n = 0; n < N; n++) {
            const float t = dz[n];
            if (g) {
                g[ob + n] += t;
                for (int k = 0; k < K1; k++) g[oW + k * N + n] += x1[r * K1 + k] * t;
                for (int j = 0; j < K2; j++) g[oW + (K1 + j) * N + n] += x2[r * K2 + j] * t;
            }
        }
        if (dx1)
            for (int k = 0; k < K1; k++) {
                float acc = 0.0f;
                for (int n = 0; n < N; n++) acc += dz[n] * W[k * N + n];
                dx1[r * K1 + k] = acc;
            }
        if (dx2)
            for (int j = 0; j < K2; j++) {
                float acc = 0.0f;
                for (int n = 0; n < N; n++) acc += dz[n] * W[(K1 + j) * N + n];
                dx2[r * K2 + j] = acc;
            }
    }
}

static void adam_range(float* var, float* m, float* v, const float* g, int n, float lr, float b1p, float b2p) {
    const float beta1 = 0.9f, beta2 = 0.999f, eps = 1e-8f;
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    for (int i = 0; i < n; i++) {
        m[i] += (g[i] - m[i]) * (1.0f - beta1);
        v[i] += (g[i] * g[i] - v[i]) * (1.0f - beta2);
        var[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
    }
}

typedef struct {                       /* activations of one forward pass */
    float *h1, *n1, *r1, *h2, *n2, *r2, *c1, *nc, *rc, *g2, *n3, *r3, *mu, *q;
} act_t;

static void actor_fwd(const float* th, vo_t o, vd_t d, const float* x, int B, act_t* a) {
    layer_fwd(x, d.S, NULL, 0, B, th + o.W1, th + o.b1, d.H1, d.norm, th + o.l1b, th + o.l1g, a->h1, a->n1, a->r1);
    layer_fwd(a->h1, d.H1, NULL, 0, B, th + o.Wa2, th + o.ba2, d.HA, d.norm, th + o.l2b, th + o.l2g, a->h2, a->n2, a->r2);
    for (int r = 0; r < B; r++)
        for (int j = 0; j < d.A; j++) {
            float acc = 0.0f;
            for (int n = 0; n < d.HA; n++) acc += a->h2[r * d.HA + n] * th[o.Wa3 + n * d.A + j];
            a->mu[r * d.A + j] = tanhf(acc + th[o.ba3 + j]);
        }
}

/* critic on (x, act); first = the critic's first-layer activation (recomputed here when the networks are separate or
 * when shared_h1 is NULL) */
static void critic_fwd(const float* th, vo_t o, vd_t d, const float* x, const float* act, int B, act_t* a,
                       const float* shared_h1) {
    const float* c1 = shared_h1;
    if (d.sep || !shared_h1) {
        layer_fwd(x, d.S, NULL, 0, B, th + o.Wc1, th + o.bc1, d.H1, d.norm, th + o.lcb, th + o.lcg, a->c1, a->nc, a->rc);
        c1 = a->c1;
    }
    layer_fwd(c1, d.H1, act, d.A, B, th + o.Wc2, th + o.bc2, d.HC, d.norm, th + o.l3b, th + o.l3g, a->g2, a->n3, a->r3);
    for (int r = 0; r < B; r++) {
        float acc = 0.0f;
        for (int n = 0; n < d.HC; n++) acc += a->g2[r * d.HC + n] * th[o.Wc3 + n];
        a->q[r] = acc + th[o.bc3];
    }
}

static void ddpg_variant_update_impl(int S, int A, int H1, int HA, int HC, int norm, int sep, int B,
                                     float* theta, float* theta_t, float* m_a, float* v_a, float* m_c, float* v_c, float* pw,
                                     const float* s, const float* a, const double* r, const float* s2, const double* gam,
                                     float actor_lr, float critic_lr, float tau, int do_clip, const float* smin,
                                     const float* smax, const float* amax, float* tap_q, float* tap_y, float* tap_aout,
                                     float* tap_dqda, float* tap_gc, float* tap_ga) {
    vd_t d = {S, A, H1, HA, HC, norm, sep};
    vo_t o = voffsets(d);
    const int P = o.P;
    const int HM = HA > HC ? HA : HC;
    float* buf = calloc((size_t)B * (2 * S + 8 * H1 + 4 * HM + 3 * HA + 3 * HC + 6 * A + 16) + P, sizeof(float));
    float* p = buf;
#define TAKE(n) (p += (n), p - (n))
    float* x = TAKE(B * S); float* x2 = TAKE(B * S);
    act_t t;
    t.h1 = TAKE(B * H1); t.n1 = TAKE(B * H1); t.r1 = TAKE(B);
    t.h2 = TAKE(B * HA); t.n2 = TAKE(B * HA); t.r2 = TAKE(B);
    t.c1 = TAKE(B * H1); t.nc = TAKE(B * H1); t.rc = TAKE(B);
    t.g2 = TAKE(B * HC); t.n3 = TAKE(B * HC); t.r3 = TAKE(B);
    t.mu = TAKE(B * A); t.q = TAKE(B);
    float* aout = TAKE(B * A); float* y = TAKE(B); float* dg2 = TAKE(B * HM); float* dh1 = TAKE(B * H1);
    float* dh2 = TAKE(B * HA); float* dqda = TAKE(B * A); float* dz = TAKE(B * A); float* dx1 = TAKE(B * H1);
    float* g = TAKE(P);
#undef TAKE
    for (int i = 0; i < B * S; i++) {
        const int k = i % S;
        x[i] = do_clip ? fminf(fmaxf((s[i] - 0.0f) / 1.0f, smin[k]), smax[k]) : s[i];
        x2[i] = do_clip ? fminf(fmaxf((s2[i] - 0.0f) / 1.0f, smin[k]), smax[k]) : s2[i];
    }
    /* steps 1-2: target networks on s' (DDPG.py:77) */
    actor_fwd(theta_t, o, d, x2, B, &t);
    for (int i = 0; i < B * A; i++) aout[i] = t.mu[i] * amax[i % A];
    critic_fwd(theta_t, o, d, x2, aout, B, &t, t.h1);
    for (int b = 0; b < B; b++) y[b] = (float)(r[b] + gam[b] * (double)t.q[b]);      /* float64 glue (DDPG.py:80-84) */
    if (tap_y) memcpy(tap_y, y, sizeof(float) * B);

    /* step 3: critic step */
    critic_fwd(theta, o, d, x, a, B, &t, NULL);                 /* first layer lands in t.c1 / t.nc / t.rc */
    if (tap_q) memcpy(tap_q, t.q, sizeof(float) * B);
    memset(g, 0, sizeof(float) * P);
    for (int b = 0; b < B; b++) {
        const float dq = 2.0f * (t.q[b] - y[b]) / (float)B;     /* d mean((y-q)^2) / dq */
        g[o.bc3] += dq;
        for (int n = 0; n < HC; n++) {
            g[o.Wc3 + n] += t.g2[b * HC + n] * dq;
            dg2[b * HC + n] = dq * theta[o.Wc3 + n];
        }
    }
    layer_bwd(t.c1, H1, a, A, B, theta + o.Wc2, HC, norm, theta + o.l3g, t.g2, t.n3, t.r3, dg2, g, o.Wc2, o.bc2, o.l3b,
              o.l3g, dh1, NULL);
    layer_bwd(x, S, NULL, 0, B, theta + o.Wc1, H1, norm, theta + o.lcg, t.c1, t.nc, t.rc, dh1, g, o.Wc1, o.bc1, o.lcb,
              o.lcg, NULL, NULL);
    if (tap_gc) memcpy(tap_gc, g, sizeof(float) * P);
    if (!sep) adam_range(theta + o.W1, m_c + o.W1, v_c + o.W1, g + o.W1, o.Wa2 - o.W1, critic_lr, pw[2], pw[3]);
    adam_range(theta + o.critic0, m_c + o.critic0, v_c + o.critic0, g + o.critic0, P - o.critic0, critic_lr, pw[2], pw[3]);
    pw[2] *= 0.9f; pw[3] *= 0.999f;

    /* step 4: actor forward with the updated first layer (DDPG.py:90) */
    actor_fwd(theta, o, d, x, B, &t);
    for (int i = 0; i < B * A; i++) aout[i] = t.mu[i] * amax[i % A];
    if (tap_aout) memcpy(tap_aout, aout, sizeof(float) * B * A);
    /* step 5: dQ/da at the scaled action with the updated critic (DDPG.py:91) */
    critic_fwd(theta, o, d, x, aout, B, &t, t.h1);
    for (int b = 0; b < B; b++)
        for (int n = 0; n < HC; n++) dg2[b * HC + n] = theta[o.Wc3 + n];
    layer_bwd(sep ? t.c1 : t.h1, H1, aout, A, B, theta + o.Wc2, HC, norm, theta + o.l3g, t.g2, t.n3, t.r3, dg2, NULL, 0, 0,
              0, 0, NULL, dqda);
    if (tap_dqda) memcpy(tap_dqda, dqda, sizeof(float) * B * A);
    /* step 6: actor step; grad_ys = -dQ/da on the UNSCALED tanh output, batch SUM (quirk Q3) */
    memset(g, 0, sizeof(float) * P);
    for (int i = 0; i < B * A; i++) dz[i] = -dqda[i] * (1.0f - t.mu[i] * t.mu[i]);
    for (int b = 0; b < B; b++) {
        for (int j = 0; j < A; j++) g[o.ba3 + j] += dz[b * A + j];
        for (int n = 0; n < HA; n++) {
            float acc = 0.0f;
            for (int j = 0; j < A; j++) {
                g[o.Wa3 + n * A + j] += t.h2[b * HA + n] * dz[b * A + j];
                acc += dz[b * A + j] * theta[o.Wa3 + n * A + j];
            }
            dh2[b * HA + n] = acc;
        }
    }
    layer_bwd(t.h1, H1, NULL, 0, B, theta + o.Wa2, HA, norm, theta + o.l2g, t.h2, t.n2, t.r2, dh2, g, o.Wa2, o.ba2, o.l2b,
              o.l2g, dx1, NULL);
    layer_bwd(x, S, NULL, 0, B, theta + o.W1, H1, norm, theta + o.l1g, t.h1, t.n1, t.r1, dx1, g, o.W1, o.b1, o.l1b, o.l1g,
              NULL, NULL);
    if (tap_ga) memcpy(tap_ga, g, sizeof(float) * P);
    adam_range(theta, m_a, v_a, g, o.critic0, actor_lr, pw[0], pw[1]);
    pw[0] *= 0.9f; pw[1] *= 0.999f;
    /* step 7: Polyak over every tensor */
    for (int i = 0; i < P; i++) theta_t[i] += tau * (theta[i] - theta_t[i]);
    free(buf);
}

void ddpg_variant_update(int S, int A, int H1, int HA, int HC, int norm, int sep, int B,
                         float* theta, float* theta_t, float* m_a, float* v_a, float* m_c, float* v_c, float* pw,
                         const float* s, const float* a, const double* r, const float* s2, const double* gam,
                         float actor_lr, float critic_lr, float tau, int do_clip, const float* smin,
                         const float* smax, const float* amax, float* tap_q, float* tap_y, float* tap_aout,
                         float* tap_dqda, float* tap_gc, float* tap_ga) {
    const unsigned csr = oracle_ftz_on();       /* TF-1.15 CPU arithmetic: denormals flushed (oracle/ftz.h) */
    ddpg_variant_update_impl(S, A, H1, HA, HC, norm, sep, B, theta, theta_t, m_a, v_a, m_c, v_c, pw, s, a, r, s2, gam,
                             actor_lr, critic_lr, tau, do_clip, smin, smax, amax, tap_q, tap_y, tap_aout, tap_dqda,
                             tap_gc, tap_ga);
    oracle_ftz_restore(csr);
}

/* greedy scaled action (predict_action) and Q(s, a) (predict_qval) for B rows */
void ddpg_variant_act(int S, int A, int H1, int HA, int HC, int norm, int sep, const float* theta, const float* states,
                      int B, int do_clip, const float* smin, const float* smax, const float* amax, float* out) {
    vd_t d = {S, A, H1, HA, HC, norm, sep};
    vo_t o = voffsets(d);
    float* buf = calloc((size_t)B * (S + 2 * H1 + 2 * HA + A + 4), sizeof(float));
    float* x = buf;
    act_t t;
    memset(&t, 0, sizeof(t));
    t.h1 = x + B * S; t.n1 = t.h1 + B * H1; t.r1 = t.n1 + B * H1; t.h2 = t.r1 + B; t.n2 = t.h2 + B * HA; t.r2 = t.n2 + B * HA;
    t.mu = t.r2 + B;
    for (int i = 0; i < B * S; i++) x[i] = do_clip ? fminf(fmaxf(states[i], smin[i % S]), smax[i % S]) : states[i];
    actor_fwd(theta, o, d, x, B, &t);
    for (int i = 0; i < B * A; i++) out[i] = t.mu[i] * amax[i % A];
    free(buf);
}
