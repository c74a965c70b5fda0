// ddpg_generic.hip -- dimension-generic fused DDPG update + acting kernels (fp32 VALU path).
//
// One workgroup per agent; n_updates sequential updates per launch.  Each update fuses what the
// reference does in BaseAgent.learn (agents/base_agent.py:65-70): sample_batch
// (utils/replaybuffer.py:32-37) and DDPG_Network_Manager.update_network (agents/DDPG.py:74-95,
// seven Session.run calls) into one kernel iteration:
//   gather -> target actor/critic on s' -> float64 TD target (Q5) -> critic forward/backward, MSE,
//   TF-Adam (critic optimizer, incl. the shared trunk: Q1) -> actor forward with the UPDATED trunk ->
//   dQ/da at the scaled action -> actor backward (batch SUM, unscaled tanh: Q3) -> TF-Adam (actor
//   optimizer) -> Polyak on all ten tensors.
// This is the any-shape path (arbitrary S, A, H1, HA, HC, B <= 128); the gfx950 matrix-core path for
// the headline shapes is ddpg_mfma.hip.  Activations [B,H] live in a per-agent global scratch that
// stays in the XCD's L2; per-sample vectors live in LDS.
#include "generic_blocks.h"
#include "ddpg_rollout_device.h"

namespace {

using namespace gen;

struct Lds {
    float *x, *x2, *a, *aout, *mu, *dqda, *dz, *q, *y, *dq;
    double *r, *g;
    long long* idx;
    int* pool;
    int* dups;
    float* pol;       // scratch of the on-device training step (ddpg_rollout_device.h)
    float* rstd;      // [4][B] reciprocal standard deviations of the layer-norm rows
};

__host__ __device__ inline size_t lds_carve(const RlcDims& d, unsigned char* base, Lds* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int B = d.B, S = d.S, A = d.A;
    double* r = (double*)take(sizeof(double) * B);
    double* g = (double*)take(sizeof(double) * B);
    long long* idx = (long long*)take(sizeof(long long) * RLC_MAX_BATCH);
    float* x = (float*)take(sizeof(float) * B * S);
    float* x2 = (float*)take(sizeof(float) * B * S);
    float* a = (float*)take(sizeof(float) * B * A);
    float* aout = (float*)take(sizeof(float) * B * A);
    float* mu = (float*)take(sizeof(float) * B * A);
    float* dqda = (float*)take(sizeof(float) * B * A);
    float* dz = (float*)take(sizeof(float) * B * A);
    float* q = (float*)take(sizeof(float) * B);
    float* y = (float*)take(sizeof(float) * B);
    float* dq = (float*)take(sizeof(float) * B);
    int* pool = (int*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    int* dups = (int*)take(sizeof(int) * 4);
    float* pol = (float*)take(sizeof(float) * (ddpg_policy_lds_floats(d) + 4));
    float* rstd = (float*)take(sizeof(float) * 4 * B);
    if (out) {
        out->rstd = rstd;
        out->r = r; out->g = g; out->idx = idx; out->x = x; out->x2 = x2; out->a = a; out->aout = aout;
        out->mu = mu; out->dqda = dqda; out->dz = dz; out->q = q; out->y = y; out->dq = dq;
        out->pool = pool; out->dups = dups; out->pol = pol;
    }
    return off;
}

__global__ __launch_bounds__(kThreads) void rlc_ddpg_update_generic_kernel(RlcDev dv_arg, int first_agent,
                                                                           int n_updates, int source,
                                                                           const long long* host_idx,
                                                                           int grad_taps, const RlcRollout* rollout,
                                                                           int q8_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the population view is read through gen::kernarg_view (generic_blocks.h), made opaque again by DDPG_PHASE() at the
    // start of every phase (dv_arg is the first argument: offset 0)
    const RlcDev* dvp;
#define DDPG_PHASE() (dvp = kernarg_view<RlcDev>())
#define dv (*dvp)
#define d (dvp->d)
    DDPG_PHASE();
    const int S = d.S, A = d.A, H1 = d.H1, HA = d.HA, HC = d.HC, B = d.B;
    const int agent = first_agent + blockIdx.x;
    const int tid = threadIdx.x;
    Lds L;
    lds_carve(d, smem, &L);

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* sc = dv.scratch + (size_t)agent * dv.scratch_stride;
    const int HM = max(HA, HC), NORM = d.norm, SEP = d.sep;
    float* h1 = sc;                  sc += (size_t)B * H1;
    float* h2 = sc;                  sc += (size_t)B * HA;
    float* g2 = sc;                  sc += (size_t)B * HC;
    float* d2 = sc;                  sc += (size_t)B * HM;
    float* dh1 = sc;                 sc += (size_t)B * H1;
    // layer norm keeps the normalised activations of the three hidden layers for the backward pass; separate
    // networks give the critic a first layer (c1) of its own -- otherwise it reads the shared h1
    float* n1 = sc;                  if (NORM) sc += (size_t)B * H1;
    float* n2 = sc;                  if (NORM) sc += (size_t)B * HA;
    float* n3 = sc;                  if (NORM) sc += (size_t)B * HC;
    float* c1 = SEP ? sc : h1;       if (SEP) sc += (size_t)B * H1;
    float* nc = SEP ? sc : n1;       if (SEP && NORM) sc += (size_t)B * H1;
    float* r1 = L.rstd; float* r2 = r1 + B; float* r3 = r2 + B; float* rc = SEP ? r3 + B : r1;
    float* pw = dv.pw + agent * 4;
    const float lr_a = dv.actor_lr[agent], lr_c = dv.critic_lr[agent];
    float* tap_gc = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;
    float* tap_ga = grad_taps ? dv.tap_ga + (size_t)agent * d.Ppad : nullptr;
    // hidden layer: Y = relu([LN](X.W[0:K] + E.W[K:K+Ke] + b)); nh / rs (layer norm only) keep what its backward needs
    auto hidden = [&](const float* X, int K, const float* E, int Ke, const float* P, int oW, int ob, int olb, int olg, int N,
                      float* Y, float* nh, float* rs) {
        blk_dense(X, K, K, E, Ke, P + oW, P + ob, N, Y, N, B, NORM ? 0 : 1);
        if (NORM) {
            __syncthreads();
            blk_layernorm_relu(Y, N, B, P + olb, P + olg, nh, rs);
        }
        __syncthreads();
    };
    // layer-norm backward of a hidden layer whose masked output gradient is dY: returns this thread's (gamma, beta)
    // gradient column sums and turns dY into the gradient w.r.t. the linear output, with the PRE-step gamma
    auto ln_bwd = [&](float* dY, const float* nh, const float* rs, int olg, int N, float& gg, float& gb) {
        if (!NORM) return;
        blk_layernorm_param_grads(dY, nh, N, B, gg, gb);
        __syncthreads();
        blk_layernorm_bwd_rows(dY, nh, rs, th + olg, N, B);
        __syncthreads();
    };
    auto ln_adam = [&](const AdamCtx& c, int olb, int olg, int N, float gg, float gb) {
        if (NORM && tid < N) { adam_apply(c, olg + tid, gg); adam_apply(c, olb + tid, gb); }
    };

    for (int u = 0; u < n_updates; u++) {
        if (rollout) {
            // on-device experiment loop: one environment step first; update when learn() would run
            if (!rlc_train_step_device(rollout, agent, L.pol, u == 0 ? q8_first : 0)) continue;
        }
        // ---- sample + gather (utils/replaybuffer.py:32-37) ----
        DDPG_PHASE();
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads)
                L.idx[b] = host_idx[((size_t)blockIdx.x * n_updates + u) * B + b];
        }
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * A; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = dv.rep.gr[slot]; L.g[b] = dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * A; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = dv.rep.rr[slot]; L.g[b] = dv.rep.rg[slot];
            }
            for (int i = 0; i < S; i++) {
                L.x[b * S + i] = clip_state_val(ps[i], dv.clip_state, dv.smin[i], dv.smax[i]);
                L.x2[b * S + i] = clip_state_val(ps2[i], dv.clip_state, dv.smin[i], dv.smax[i]);
            }
            for (int j = 0; j < A; j++) L.a[b * A + j] = pa[j];
        }
        __syncthreads();

        // ---- steps 1-2: target actor / critic on s' (DDPG.py:77) ----
        DDPG_PHASE();
        hidden(L.x2, S, nullptr, 0, tt, d.oW1, d.ob1, d.oL1b, d.oL1g, H1, h1, nullptr, nullptr);
        hidden(h1, H1, nullptr, 0, tt, d.oWa2, d.oba2, d.oL2b, d.oL2g, HA, h2, nullptr, nullptr);
        blk_dense(h2, HA, HA, nullptr, 0, tt + d.oWa3, tt + d.oba3, A, L.mu, A, B, 2);
        __syncthreads();
        for (int i = tid; i < B * A; i += kThreads) L.aout[i] = L.mu[i] * dv.amax[i % A];
        __syncthreads();
        if (SEP) hidden(L.x2, S, nullptr, 0, tt, d.oWc1, d.obc1, d.oLcb, d.oLcg, H1, c1, nullptr, nullptr);
        hidden(c1, H1, L.aout, A, tt, d.oWc2, d.obc2, d.oL3b, d.oL3g, HC, g2, nullptr, nullptr);
        blk_dense(g2, HC, HC, nullptr, 0, tt + d.oWc3, tt + d.obc3, 1, L.q, 1, B, 0);
        __syncthreads();
        // TD target in float64, then the fp32 placeholder cast (DDPG.py:80-84)
        for (int b = tid; b < B; b += kThreads) {
            const float y = (float)(L.r[b] + L.g[b] * (double)L.q[b]);
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + b] = y;
        }
        __syncthreads();

        // ---- step 3: critic step (hydra_ddpg_network.py:71-72) ----
        DDPG_PHASE();
        hidden(L.x, S, nullptr, 0, th, d.oWc1, d.obc1, d.oLcb, d.oLcg, H1, c1, nc, rc);
        hidden(c1, H1, L.a, A, th, d.oWc2, d.obc2, d.oL3b, d.oL3g, HC, g2, n3, r3);
        blk_dense(g2, HC, HC, nullptr, 0, th + d.oWc3, th + d.obc3, 1, L.q, 1, B, 0);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = L.q[b];
            L.dq[b] = 2.0f * (L.q[b] - L.y[b]) / (float)B;          // d mean((y-q)^2) / dq
        }
        __syncthreads();
        for (int it = tid; it < B * HC; it += kThreads) {
            const int b = it / HC, n = it % HC;
            d2[it] = g2[it] > 0.0f ? L.dq[b] * th[d.oWc3 + n] : 0.0f;
        }
        __syncthreads();
        float gg3 = 0.f, gb3 = 0.f, ggc = 0.f, gbc = 0.f;
        ln_bwd(d2, n3, r3, d.oL3g, HC, gg3, gb3);
        blk_dense_bwd_input(d2, HC, th + d.oWc2, c1, H1, dh1, B);    // uses the pre-step Wc2
        __syncthreads();
        ln_bwd(dh1, nc, rc, d.oLcg, H1, ggc, gbc);
        {
            const AdamCtx c = {th, dv.m_c + (size_t)agent * d.Ppad, dv.v_c + (size_t)agent * d.Ppad,
                               adam_alpha(lr_c, pw[2], pw[3]), tap_gc};
            blk_dense_grad_adam(g2, HC, HC, nullptr, 0, L.dq, 1, B, c, d.oWc3, d.obc3);
            blk_dense_grad_adam(c1, H1, H1, L.a, A, d2, HC, B, c, d.oWc2, d.obc2);
            blk_dense_grad_adam(L.x, S, S, nullptr, 0, dh1, H1, B, c, d.oWc1, d.obc1);
            ln_adam(c, d.oL3b, d.oL3g, HC, gg3, gb3);
            ln_adam(c, d.oLcb, d.oLcg, H1, ggc, gbc);
        }
        __syncthreads();
        if (tid == 0) { pw[2] *= 0.9f; pw[3] *= 0.999f; }

        // ---- step 4: actor forward with the updated first layer (DDPG.py:90) ----
        DDPG_PHASE();
        hidden(L.x, S, nullptr, 0, th, d.oW1, d.ob1, d.oL1b, d.oL1g, H1, h1, n1, r1);
        hidden(h1, H1, nullptr, 0, th, d.oWa2, d.oba2, d.oL2b, d.oL2g, HA, h2, n2, r2);
        blk_dense(h2, HA, HA, nullptr, 0, th + d.oWa3, th + d.oba3, A, L.mu, A, B, 2);
        __syncthreads();
        for (int i = tid; i < B * A; i += kThreads) {
            const float ao = L.mu[i] * dv.amax[i % A];
            L.aout[i] = ao;
            dv.tap_aout[(size_t)agent * RLC_MAX_BATCH * A + i] = ao;
        }
        __syncthreads();
        // ---- step 5: dQ/da at the scaled action with the updated critic (DDPG.py:91) ----
        DDPG_PHASE();
        if (SEP) hidden(L.x, S, nullptr, 0, th, d.oWc1, d.obc1, d.oLcb, d.oLcg, H1, c1, nullptr, nullptr);
        hidden(c1, H1, L.aout, A, th, d.oWc2, d.obc2, d.oL3b, d.oL3g, HC, g2, n3, r3);
        if (NORM) {
            // through the layer norm: dz3 = LN'(relu'(.) * Wc3), then its action rows
            for (int it = tid; it < B * HC; it += kThreads) d2[it] = g2[it] > 0.0f ? th[d.oWc3 + it % HC] : 0.0f;
            __syncthreads();
            blk_layernorm_bwd_rows(d2, n3, r3, th + d.oL3g, HC, B);
            __syncthreads();
        }
        for (int it = tid; it < B * A; it += kThreads) {
            const int b = it / A, j = it % A;
            float acc = 0.0f;
            if (NORM) {
                for (int n = 0; n < HC; n++) acc += d2[(size_t)b * HC + n] * th[d.oWc2 + (size_t)(H1 + j) * HC + n];
            } else {
                for (int n = 0; n < HC; n++)
                    if (g2[(size_t)b * HC + n] > 0.0f) acc += th[d.oWc3 + n] * th[d.oWc2 + (size_t)(H1 + j) * HC + n];
            }
            L.dqda[it] = acc;
            dv.tap_dqda[(size_t)agent * RLC_MAX_BATCH * A + it] = acc;
            L.dz[it] = -acc * (1.0f - L.mu[it] * L.mu[it]);          // grad_ys = -dQ/da on tanh output (Q3)
        }
        __syncthreads();
        // ---- step 6: actor step ----
        DDPG_PHASE();
        for (int it = tid; it < B * HA; it += kThreads) {
            const int b = it / HA, n = it % HA;
            float acc = 0.0f;
            for (int j = 0; j < A; j++) acc += L.dz[b * A + j] * th[d.oWa3 + n * A + j];
            d2[it] = h2[it] > 0.0f ? acc : 0.0f;
        }
        __syncthreads();
        float gg2 = 0.f, gb2 = 0.f, gg1 = 0.f, gb1 = 0.f;
        ln_bwd(d2, n2, r2, d.oL2g, HA, gg2, gb2);
        blk_dense_bwd_input(d2, HA, th + d.oWa2, h1, H1, dh1, B);
        __syncthreads();
        ln_bwd(dh1, n1, r1, d.oL1g, H1, gg1, gb1);
        {
            const AdamCtx c = {th, dv.m_a + (size_t)agent * d.Ppad, dv.v_a + (size_t)agent * d.Ppad,
                               adam_alpha(lr_a, pw[0], pw[1]), tap_ga};
            blk_dense_grad_adam(h2, HA, HA, nullptr, 0, L.dz, A, B, c, d.oWa3, d.oba3);
            blk_dense_grad_adam(h1, H1, H1, nullptr, 0, d2, HA, B, c, d.oWa2, d.oba2);
            blk_dense_grad_adam(L.x, S, S, nullptr, 0, dh1, H1, B, c, d.oW1, d.ob1);
            ln_adam(c, d.oL2b, d.oL2g, HA, gg2, gb2);
            ln_adam(c, d.oL1b, d.oL1g, H1, gg1, gb1);
        }
        __syncthreads();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; }
        // ---- step 7: Polyak on every tensor (hydra_ddpg_network.py:29) ----
        DDPG_PHASE();
        for (int p = tid; p < d.Pdev; p += kThreads) {
            const float t = tt[p];
            tt[p] = t + dv.tau * (th[p] - t);
        }
        __syncthreads();
    }
#undef DDPG_PHASE
#undef dv
#undef d
}

// greedy action (+ optional device OU noise) for one state per agent: predict_action on B=1
// (agents/DDPG.py:36-44; utils/exploration_policy.py:18-21).  One workgroup per agent.
// done_flag (or null): a word in host-visible memory that receives done_val once every action of the launch is stored
// (single-workgroup launches only: the queued forward of a drop-in agent, rlc_ddpg_act_queue)
__global__ __launch_bounds__(kThreads) void rlc_ddpg_act_kernel(RlcDev dv, int first_agent, const float* states,
                                                                float* out, int explore, int* done_flag, int done_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcDims d = dv.d;
    const int S = d.S, A = d.A;
    const int agent = first_agent + blockIdx.x;
    const int tid = threadIdx.x;
    const DdpgPolicyLds L = ddpg_policy_carve(d, (float*)smem);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    for (int i = tid; i < S; i += kThreads)
        L.x[i] = clip_state_val(states[(size_t)blockIdx.x * S + i], dv.clip_state, dv.smin[i], dv.smax[i]);
    ddpg_greedy_forward(d, th, L, dv.amax);
    if (tid < A) {
        float act = L.act[tid];
        if (explore) act = ddpg_ou_explore(dv, agent, tid, act, dv.noise_ctr[agent]);
        out[(size_t)blockIdx.x * A + tid] = act;
    }
    __syncthreads();
    if (explore && tid == 0) dv.noise_ctr[agent] += 1;
    if (done_flag && tid == 0) {
        __threadfence_system();                     // the action stores (tid < A, ordered by the barrier above) first
        __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ void rlc_reset_noise_kernel(RlcDev dv, int first_agent, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * dv.d.A) dv.ou_state[(size_t)first_agent * dv.d.A + i] = dv.ou_mu;
}

// Q(s,a) rows on one agent's online network: predict_qval (hydra_ddpg_network.py:183-193).
// One workgroup per row.
__global__ __launch_bounds__(kThreads) void rlc_ddpg_qval_kernel(RlcDev dv, int agent, const float* states,
                                                                 const float* actions, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcDims d = dv.d;
    const int S = d.S, A = d.A, H1 = d.H1, HC = d.HC;
    const int tid = threadIdx.x, row = blockIdx.x;
    float* x = (float*)smem;
    float* h1 = x + ((S + A + 3) & ~3);
    float* g2 = h1 + ((H1 + 3) & ~3);
    float* red = g2 + ((HC + 3) & ~3);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    for (int i = tid; i < S; i += kThreads)
        x[i] = clip_state_val(states[(size_t)row * S + i], dv.clip_state, dv.smin[i], dv.smax[i]);
    for (int j = tid; j < A; j += kThreads) x[S + j] = actions[(size_t)row * A + j];
    __syncthreads();
    // the critic's first layer: the shared one of the hydra network, its own with separate networks
    for (int k = tid; k < H1; k += kThreads) {
        float acc = 0.0f;
        for (int i = 0; i < S; i++) acc += x[i] * th[d.oWc1 + i * H1 + k];
        acc += th[d.obc1 + k];
        h1[k] = d.norm ? acc : fmaxf(acc, 0.0f);
    }
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(h1, H1, th + d.oLcb, th + d.oLcg, red);
    for (int n = tid; n < HC; n += kThreads) {
        float acc = 0.0f;
        for (int k = 0; k < H1; k++) acc += h1[k] * th[d.oWc2 + rlc_widx(d.blocked, k, n, HC)];
        for (int j = 0; j < A; j++) acc += x[S + j] * th[d.oWc2 + rlc_widx(d.blocked, d.arow0 + j, n, HC)];
        acc += th[d.obc2 + n];
        g2[n] = d.norm ? acc : fmaxf(acc, 0.0f);
    }
    __syncthreads();
    if (d.norm) rlc_row_layernorm_relu(g2, HC, th + d.oL3b, th + d.oL3g, red);
    float part = 0.0f;
    for (int n = tid; n < HC; n += kThreads) part += g2[n] * th[d.oWc3 + n];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, RLC_WAVE);
    __syncthreads();
    if (tid % RLC_WAVE == 0) red[tid / RLC_WAVE] = part;
    __syncthreads();
    if (tid == 0) {
        float q = 0.0f;
        for (int w = 0; w < kThreads / RLC_WAVE; w++) q += red[w];
        out[row] = q + th[d.obc3];
    }
}

}  // namespace

size_t rlc_generic_scratch_floats(const RlcDims& d) {
    const size_t B = d.B;
    size_t n = B * d.H1 * 2 + B * d.HA + B * d.HC + B * (size_t)(d.HA > d.HC ? d.HA : d.HC);
    if (d.norm) n += B * ((size_t)d.H1 + d.HA + d.HC);
    if (d.sep) n += B * (size_t)d.H1 * (d.norm ? 2 : 1);
    return n;
}

int rlc_launch_ddpg_update_generic(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                   const long long* idx_dev, int grad_taps, hipStream_t st, const RlcRollout* rollout,
                                   int q8_first) {
    const size_t lds = lds_carve(dv.d, nullptr, nullptr);
    RLC_REQUIRE(!dv.d.blocked, "the generic kernel reads row-major weights (rlc_ddpg_set_kernel re-packs them)");
    RLC_REQUIRE(lds <= 160 * 1024, "generic DDPG kernel needs %zu B of LDS (> 160 KiB)", lds);
    hipLaunchKernelGGL(rlc_ddpg_update_generic_kernel, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent,
                       n_updates, source, idx_dev, grad_taps, rollout, q8_first);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_act(const RlcDev& dv, int first_agent, int n, const float* states_dev, float* out_dev, int explore,
                   hipStream_t st, int* done_flag, int done_val) {
    const size_t lds = sizeof(float) * ddpg_policy_lds_floats(dv.d);
    RLC_REQUIRE(done_flag == nullptr || n == 1, "a completion flag needs a one-workgroup acting launch");
    hipLaunchKernelGGL(rlc_ddpg_act_kernel, dim3(n), dim3(kThreads), lds, st, dv, first_agent, states_dev, out_dev,
                       explore, done_flag, done_val);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_reset_noise(const RlcDev& dv, int first_agent, int n, hipStream_t st) {
    const int total = n * dv.d.A;
    hipLaunchKernelGGL(rlc_reset_noise_kernel, dim3((total + 255) / 256), dim3(256), 0, st, dv, first_agent, n);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_qval(const RlcDev& dv, int agent, int n, const float* states_dev, const float* actions_dev,
                    float* out_dev, hipStream_t st) {
    const size_t lds = sizeof(float) * (((dv.d.S + dv.d.A + 3) & ~3) + ((dv.d.H1 + 3) & ~3) + ((dv.d.HC + 3) & ~3) + 40);
    hipLaunchKernelGGL(rlc_ddpg_qval_kernel, dim3(n), dim3(kThreads), lds, st, dv, agent, states_dev, actions_dev,
                       out_dev);
    RLC_HIP(hipGetLastError());
    return 0;
}
