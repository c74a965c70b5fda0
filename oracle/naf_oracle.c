/*
 * oracle/naf_oracle.c -- CPU restatement of the reference's NAF agent (normalized advantage functions).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Restates (reference file:line under /root/reference):
 *   agents/NAF.py:69-75                         update_network(): y = r + gamma*V'(s') in float64, train, Polyak
 *   agents/network/naf_network.py:49-63         loss = SUM (y - Q)^2 (not the mean), one Adam, Polyak by assign_add
 *   agents/network/naf_network.py:65-123        clip(x, state_min, state_max); trunk S->L1; mu branch L1->L2->A tanh*a_max;
 *                                               V branch L1->L2->1; L columns: diag = exp(clip(fc,-5,5)) (one fc per
 *                                               action dim), below-diagonal fc of widths A-1, A-2, ..., 1;
 *                                               p_c = sum_k (a-mu)[c+k]*Lcol_c[k]; Adv = -0.5 sum_c p_c^2; Q = V + Adv
 * Blob = variable creation order: W1[S,L1] b1 | Wa2[L1,L2] ba2 | Wa3[L2,A] ba3 | Wv2[L1,L2] bv2 | Wv3[L2] bv3 |
 *        for c < A: Wd_c[L1] bd_c | for c < A-1: Wn_c[L1,A-1-c] bn_c
 * tf.clip_by_value passes the gradient where lo <= x <= hi.  TensorFlow 1.15 arithmetic: "parity unpinned";
 * cross-checked by tests/torch_ref_naf.py.
 */
#include "ftz.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NAF_MAX_A 8

typedef struct {
    int S, A, L1, L2;
    int W1, b1, Wa2, ba2, Wa3, ba3, Wv2, bv2, Wv3, bv3;
    int Wd[NAF_MAX_A], bd[NAF_MAX_A], Wn[NAF_MAX_A], bn[NAF_MAX_A];
    int P;
} naf_t;

static naf_t naf_layout(int S, int A, int L1, int L2) {
    naf_t o;
    int p = 0;
    o.S = S; o.A = A; o.L1 = L1; o.L2 = L2;
    o.W1 = p; p += S * L1;   o.b1 = p; p += L1;
    o.Wa2 = p; p += L1 * L2; o.ba2 = p; p += L2;
    o.Wa3 = p; p += L2 * A;  o.ba3 = p; p += A;
    o.Wv2 = p; p += L1 * L2; o.bv2 = p; p += L2;
    o.Wv3 = p; p += L2;      o.bv3 = p; p += 1;
    for (int c = 0; c < A; c++) { o.Wd[c] = p; p += L1; o.bd[c] = p; p += 1; }
    for (int c = 0; c < A - 1; c++) { o.Wn[c] = p; p += L1 * (A - 1 - c); o.bn[c] = p; p += A - 1 - c; }
    o.P = p;
    return o;
}

int naf_oracle_param_count(int S, int A, int L1, int L2) { return naf_layout(S, A, L1, L2).P; }

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

/* gW += X^T dY, gb += sum dY; dX (+)= dY.W^T (no mask) */
static void dense_bwd_acc(const float* x, const float* dy, int B, int K, const float* W, int N, float* gW, float* gb,
                          float* dx_acc) {
    for (int i = 0; i < B; i++)
        for (int n = 0; n < N; n++) {
            const float t = dy[i * N + n];
            gb[n] += t;
            for (int k = 0; k < K; k++) gW[k * N + n] += x[i * K + k] * t;
        }
    if (dx_acc)
        for (int i = 0; i < B; i++)
            for (int k = 0; k < K; k++) {
                float acc = 0.0f;
                for (int n = 0; n < N; n++) acc += dy[i * N + n] * W[k * N + n];
                dx_acc[i * K + k] += acc;
            }
}

/* forward of one network: trunk h1, branches; outputs mu (scaled), V, Lcol (flattened per column), pre-activations */
static void naf_forward(const float* th, naf_t o, const float* xc, int B, const float* amax, float* h1, float* ha,
                        float* hv, float* mu_t /* tanh */, float* V, float* dpre /* [B,A] */, float* npre /* [B,NN] */) {
    const int A = o.A, L1 = o.L1, L2 = o.L2, NN = A * (A - 1) / 2;
    float* z = malloc(sizeof(float) * B * A);
    dense(xc, B, o.S, th + o.W1, th + o.b1, L1, h1, 1);
    dense(h1, B, L1, th + o.Wa2, th + o.ba2, L2, ha, 1);
    dense(ha, B, L2, th + o.Wa3, th + o.ba3, A, z, 0);
    for (int i = 0; i < B * A; i++) mu_t[i] = tanhf(z[i]);
    dense(h1, B, L1, th + o.Wv2, th + o.bv2, L2, hv, 1);
    dense(hv, B, L2, th + o.Wv3, th + o.bv3, 1, V, 0);
    if (dpre) {
        for (int c = 0; c < A; c++) {
            float* col = malloc(sizeof(float) * B);
            dense(h1, B, L1, th + o.Wd[c], th + o.bd[c], 1, col, 0);
            for (int i = 0; i < B; i++) dpre[i * A + c] = col[i];
            free(col);
        }
        int off = 0;
        for (int c = 0; c < A - 1; c++) {
            const int w = A - 1 - c;
            float* blk = malloc(sizeof(float) * B * w);
            dense(h1, B, L1, th + o.Wn[c], th + o.bn[c], w, blk, 0);
            for (int i = 0; i < B; i++)
                for (int k = 0; k < w; k++) npre[i * NN + off + k] = blk[i * w + k];
            free(blk);
            off += w;
        }
    }
    (void)amax;
    free(z);
}

/* greedy action (predict_action, naf_network.py:144-149) and the L columns (sample_action's fetch, :157-158):
 * lcols [B][A*(A+1)/2]: column c = [diag_c, below-diagonal entries ...] */
void naf_oracle_act(int S, int A, int L1, int L2, const float* theta, const float* states, int B, int do_clip,
                    const float* smin, const float* smax, const float* amax, float* out_mu, float* out_lcols) {
    naf_t o = naf_layout(S, A, L1, L2);
    const int NN = A * (A - 1) / 2, NL = A * (A + 1) / 2;
    float* xc = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * L1);
    float* ha = malloc(sizeof(float) * B * L2);
    float* hv = malloc(sizeof(float) * B * L2);
    float* mt = malloc(sizeof(float) * B * A);
    float* V = malloc(sizeof(float) * B);
    float* dpre = malloc(sizeof(float) * B * A);
    float* npre = malloc(sizeof(float) * B * (NN > 0 ? NN : 1));
    for (int i = 0; i < B; i++)
        for (int k = 0; k < S; k++) xc[i * S + k] = do_clip ? clipf(states[i * S + k], smin[k], smax[k]) : states[i * S + k];
    naf_forward(theta, o, xc, B, amax, h1, ha, hv, mt, V, dpre, npre);
    for (int i = 0; i < B; i++) {
        for (int j = 0; j < A; j++) out_mu[i * A + j] = mt[i * A + j] * amax[j];
        if (out_lcols) {
            int p = 0, off = 0;
            for (int c = 0; c < A; c++) {
                out_lcols[i * NL + p++] = expf(clipf(dpre[i * A + c], -5.0f, 5.0f));
                for (int k = 0; k < A - 1 - c; k++) out_lcols[i * NL + p++] = npre[i * NN + off + k];
                off += A - 1 - c;
            }
        }
    }
    free(xc); free(h1); free(ha); free(hv); free(mt); free(V); free(dpre); free(npre);
}

/*
 * One update_network (agents/NAF.py:69-75).  theta, theta_t, m, v [P]; pw[2] = {b1^t, b2^t};
 * s,a,s2 fp32 [B,*]; r, gam float64 [B].  taps (may be NULL): q[B], y[B], V[B], grads[P]
 */
static void naf_oracle_update_impl(int S, int A, int L1, int L2, int B, float* theta, float* theta_t, float* m, float* v, float* pw,
                       const float* s, const float* a, const double* r, const float* s2, const double* gam, float lr,
                       float tau, int do_clip, const float* smin, const float* smax, const float* amax, float* tap_q,
                       float* tap_y, float* tap_V, float* tap_g) {
    naf_t o = naf_layout(S, A, L1, L2);
    const int P = o.P, NN = A * (A - 1) / 2;
    float* xc = malloc(sizeof(float) * B * S);
    float* x2c = malloc(sizeof(float) * B * S);
    float* h1 = malloc(sizeof(float) * B * L1);
    float* ha = malloc(sizeof(float) * B * L2);
    float* hv = malloc(sizeof(float) * B * L2);
    float* mt = malloc(sizeof(float) * B * A);
    float* V = malloc(sizeof(float) * B);
    float* dpre = malloc(sizeof(float) * B * A);
    float* npre = malloc(sizeof(float) * B * (NN > 0 ? NN : 1));
    float* y = malloc(sizeof(float) * B);
    float* q = malloc(sizeof(float) * B);
    float* dz = malloc(sizeof(float) * B * A);
    float* dd = malloc(sizeof(float) * B * A);
    float* dn = calloc(B * (NN > 0 ? NN : 1), sizeof(float));
    float* dV = malloc(sizeof(float) * B);
    float* dha = calloc(B * L2, sizeof(float));
    float* dhv = calloc(B * L2, sizeof(float));
    float* dh1 = calloc(B * L1, sizeof(float));
    float* g = calloc(P, sizeof(float));
    for (int i = 0; i < B; i++)
        for (int k = 0; k < S; k++) {
            xc[i * S + k] = do_clip ? clipf(s[i * S + k], smin[k], smax[k]) : s[i * S + k];
            x2c[i * S + k] = do_clip ? clipf(s2[i * S + k], smin[k], smax[k]) : s2[i * S + k];
        }
    /* target V'(s') and the float64 TD glue (NAF.py:70) */
    naf_forward(theta_t, o, x2c, B, amax, h1, ha, hv, mt, V, NULL, NULL);
    for (int i = 0; i < B; i++) y[i] = (float)(r[i] + gam[i] * (double)V[i]);
    if (tap_y) memcpy(tap_y, y, sizeof(float) * B);
    /* online forward */
    naf_forward(theta, o, xc, B, amax, h1, ha, hv, mt, V, dpre, npre);
    if (tap_V) memcpy(tap_V, V, sizeof(float) * B);
    for (int i = 0; i < B; i++) {
        float diff[NAF_MAX_A], Lc[NAF_MAX_A][NAF_MAX_A], p[NAF_MAX_A], ddiff[NAF_MAX_A];
        int off = 0;
        for (int j = 0; j < A; j++) { diff[j] = a[i * A + j] - mt[i * A + j] * amax[j]; ddiff[j] = 0.0f; }
        float adv = 0.0f;
        for (int c = 0; c < A; c++) {
            Lc[c][0] = expf(clipf(dpre[i * A + c], -5.0f, 5.0f));
            for (int k = 1; k < A - c; k++) Lc[c][k] = npre[i * NN + off + k - 1];
            off += A - 1 - c;
            float pc = 0.0f;
            for (int k = 0; k < A - c; k++) pc += diff[c + k] * Lc[c][k];
            p[c] = pc;
            adv += pc * pc;
        }
        q[i] = V[i] + (-0.5f * adv);
        const float dq = 2.0f * (q[i] - y[i]);           /* loss = SUM (y - q)^2 */
        dV[i] = dq;
        off = 0;
        for (int c = 0; c < A; c++) {
            const float dp = -p[c] * dq;
            for (int k = 0; k < A - c; k++) ddiff[c + k] += dp * Lc[c][k];
            const float x = dpre[i * A + c];
            dd[i * A + c] = (x >= -5.0f && x <= 5.0f) ? dp * diff[c] * Lc[c][0] : 0.0f;
            for (int k = 1; k < A - c; k++) dn[i * NN + off + k - 1] = dp * diff[c + k];
            off += A - 1 - c;
        }
        for (int j = 0; j < A; j++) {
            const float t = mt[i * A + j];
            dz[i * A + j] = -ddiff[j] * amax[j] * (1.0f - t * t);
        }
    }
    if (tap_q) memcpy(tap_q, q, sizeof(float) * B);
    /* heads -> hidden */
    dense_bwd_acc(ha, dz, B, L2, theta + o.Wa3, A, g + o.Wa3, g + o.ba3, dha);
    dense_bwd_acc(hv, dV, B, L2, theta + o.Wv3, 1, g + o.Wv3, g + o.bv3, dhv);
    for (int i = 0; i < B * L2; i++) { if (!(ha[i] > 0.0f)) dha[i] = 0.0f; if (!(hv[i] > 0.0f)) dhv[i] = 0.0f; }
    dense_bwd_acc(h1, dha, B, L1, theta + o.Wa2, L2, g + o.Wa2, g + o.ba2, dh1);
    dense_bwd_acc(h1, dhv, B, L1, theta + o.Wv2, L2, g + o.Wv2, g + o.bv2, dh1);
    {
        float* col = malloc(sizeof(float) * B);
        for (int c = 0; c < A; c++) {
            for (int i = 0; i < B; i++) col[i] = dd[i * A + c];
            dense_bwd_acc(h1, col, B, L1, theta + o.Wd[c], 1, g + o.Wd[c], g + o.bd[c], dh1);
        }
        free(col);
        int off = 0;
        for (int c = 0; c < A - 1; c++) {
            const int w = A - 1 - c;
            float* blk = malloc(sizeof(float) * B * w);
            for (int i = 0; i < B; i++)
                for (int k = 0; k < w; k++) blk[i * w + k] = dn[i * NN + off + k];
            dense_bwd_acc(h1, blk, B, L1, theta + o.Wn[c], w, g + o.Wn[c], g + o.bn[c], dh1);
            free(blk);
            off += w;
        }
    }
    for (int i = 0; i < B * L1; i++) if (!(h1[i] > 0.0f)) dh1[i] = 0.0f;
    dense_bwd_acc(xc, dh1, B, S, theta + o.W1, L1, g + o.W1, g + o.b1, NULL);
    if (tap_g) memcpy(tap_g, g, sizeof(float) * P);
    {
        const float alpha = lr * sqrtf(1.0f - pw[1]) / (1.0f - pw[0]);
        for (int i = 0; i < P; i++) {
            m[i] += (g[i] - m[i]) * (1.0f - 0.9f);
            v[i] += (g[i] * g[i] - v[i]) * (1.0f - 0.999f);
            theta[i] -= (m[i] * alpha) / (sqrtf(v[i]) + 1e-8f);
        }
        pw[0] *= 0.9f; pw[1] *= 0.999f;
    }
    for (int i = 0; i < P; i++) theta_t[i] += tau * (theta[i] - theta_t[i]);
    free(xc); free(x2c); free(h1); free(ha); free(hv); free(mt); free(V); free(dpre); free(npre); free(y); free(q);
    free(dz); free(dd); free(dn); free(dV); free(dha); free(dhv); free(dh1); free(g);
}

void naf_oracle_update(int S, int A, int L1, int L2, int B, float* theta, float* theta_t, float* m, float* v, float* pw,
                       const float* s, const float* a, const double* r, const float* s2, const double* gam, float lr,
                       float tau, int do_clip, const float* smin, const float* smax, const float* amax, float* tap_q,
                       float* tap_y, float* tap_V, float* tap_g) {
    const unsigned csr = oracle_ftz_on();       /* TF-1.15 CPU arithmetic: denormals flushed (oracle/ftz.h) */
    naf_oracle_update_impl(S, A, L1, L2, B, theta, theta_t, m, v, pw, s, a, r, s2, gam, lr, tau, do_clip, smin, smax, amax, tap_q, tap_y, tap_V, tap_g);
    oracle_ftz_restore(csr);
}
