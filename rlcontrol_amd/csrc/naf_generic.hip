// naf_generic.hip -- fused NAF update + acting kernels (any-shape fp32 VALU path).
//
// One workgroup per agent, n_updates sequential updates per launch; each update = sample_batch
// (utils/replaybuffer.py:32-37) + NAF_Network_Manager.update_network (agents/NAF.py:69-75):
//   y = r + gamma*V'(s') formed in float64 then cast (NAF.py:70), one Adam step on loss = SUM (y - Q)^2
//   (naf_network.py:53-54), Polyak by assign_add (:62-63).
// Q(s,a) = V(s) - 0.5 * sum_c p_c^2 with p_c = sum_k (a - mu)[c+k] * Lcol_c[k], Lcol_c = [exp(clip(d_c,-5,5)),
// below-diagonal fc outputs] (naf_network.py:98-121); mu = tanh(.) * action_max per dimension.
#include "generic_blocks.h"
#include "naf_common.h"
#include "naf_rollout_device.h"

namespace {

using namespace gen;

struct NLds {
    float *x, *x2, *a, *mt, *dz, *dpre, *dd, *npre, *dn, *V, *y, *q, *dV;
    double *r, *g;
    long long* idx;
    int* pool;
    int* dups;
    float* pol;       // scratch of the on-device training step (naf_rollout_device.h)
};

__host__ __device__ inline size_t nlds_carve(const RlcNafDims& d, unsigned char* base, NLds* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int B = d.B, S = d.S, A = d.A, NN = d.NN > 0 ? d.NN : 1;
    NLds L;
    L.r = (double*)take(sizeof(double) * B);
    L.g = (double*)take(sizeof(double) * B);
    L.idx = (long long*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.x = (float*)take(sizeof(float) * B * S);
    L.x2 = (float*)take(sizeof(float) * B * S);
    float** pa[] = {&L.a, &L.mt, &L.dz, &L.dpre, &L.dd};
    for (auto p : pa) *p = (float*)take(sizeof(float) * B * A);
    L.npre = (float*)take(sizeof(float) * B * NN);
    L.dn = (float*)take(sizeof(float) * B * NN);
    float** pb[] = {&L.V, &L.y, &L.q, &L.dV};
    for (auto p : pb) *p = (float*)take(sizeof(float) * B);
    L.pool = (int*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (int*)take(sizeof(int) * 4);
    L.pol = (float*)take(sizeof(float) * (naf_policy_lds_floats(d) + 4));
    if (out) *out = L;
    return off;
}

// layer-norm side outputs of one forward pass (norm_type 'layer'): normalised pre-activations [B, width] and 1/std [B]
// of the trunk, the action branch and the value branch; all null when the pass is not differentiated
struct NafLnSave { float *n1, *rs1, *na, *rsa, *nv, *rsv; };

// trunk + value branch (+ optionally the action and L heads) for B rows
__device__ inline void naf_forward(const RlcNafDims& d, const float* th, const float* xin, int B, float* h1, float* ha,
                                   float* hv, float* z_out /* [B,A] pre-tanh or null */, float* V,
                                   float* dpre /* or null */, float* npre, const NafLnSave& ln) {
    const int S = d.S, A = d.A, L1 = d.L1, L2 = d.L2, NN = d.NN;
    const int act = d.norm ? 0 : 1;     // with layer norm the relu follows the normalisation
    blk_dense(xin, S, S, nullptr, 0, th + d.W1, th + d.b1, L1, h1, L1, B, act);
    __syncthreads();
    if (d.norm) {
        blk_layernorm_relu(h1, L1, B, th + d.L1b, th + d.L1g, ln.n1, ln.rs1);
        __syncthreads();
    }
    blk_dense(h1, L1, L1, nullptr, 0, th + d.Wv2, th + d.bv2, L2, hv, L2, B, act);
    if (z_out) blk_dense(h1, L1, L1, nullptr, 0, th + d.Wa2, th + d.ba2, L2, ha, L2, B, act);
    if (dpre) {
        for (int c = 0; c < A; c++) blk_dense(h1, L1, L1, nullptr, 0, th + d.Wd[c], th + d.bd[c], 1, dpre + c, A, B, 0);
        int off = 0;
        for (int c = 0; c < A - 1; c++) {
            blk_dense(h1, L1, L1, nullptr, 0, th + d.Wn[c], th + d.bn[c], A - 1 - c, npre + off, NN, B, 0);
            off += A - 1 - c;
        }
    }
    __syncthreads();
    if (d.norm) {
        blk_layernorm_relu(hv, L2, B, th + d.Lv2b, th + d.Lv2g, ln.nv, ln.rsv);
        if (z_out) blk_layernorm_relu(ha, L2, B, th + d.La2b, th + d.La2g, ln.na, ln.rsa);
        __syncthreads();
    }
    blk_dense(hv, L2, L2, nullptr, 0, th + d.Wv3, th + d.bv3, 1, V, 1, B, 0);
    if (z_out) blk_dense(ha, L2, L2, nullptr, 0, th + d.Wa3, th + d.ba3, A, z_out, A, B, 0);
    __syncthreads();
}

__global__ __launch_bounds__(kThreads) void rlc_naf_update_kernel(RlcNafDev dv_arg, int first_agent, int n_updates, int source,
                                                                  const long long* host_idx, int grad_taps,
                                                                  const RlcNafRollout* rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the population view is read through gen::kernarg_view (generic_blocks.h), made opaque again by NAF_PHASE() at the
    // start of every phase (dv_arg is the first argument: offset 0)
    const RlcNafDev* dvp;
#define NAF_PHASE() (dvp = kernarg_view<RlcNafDev>())
#define dv (*dvp)
#define d (dvp->d)
    NAF_PHASE();
    const int S = d.S, A = d.A, L1 = d.L1, L2 = d.L2, B = d.B, NN = d.NN;
    const int agent = first_agent + blockIdx.x, tid = threadIdx.x;
    NLds L;
    nlds_carve(d, smem, &L);
    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 2;
    float* sc = dv.scratch + (size_t)agent * dv.scratch_stride;
    float* h1 = sc;  sc += (size_t)B * L1;
    float* ha = sc;  sc += (size_t)B * L2;
    float* hv = sc;  sc += (size_t)B * L2;
    float* dha = sc; sc += (size_t)B * L2;
    float* dhv = sc; sc += (size_t)B * L2;
    float* dh1 = sc; sc += (size_t)B * L1;
    // layer norm: what the backward pass of the three normalised layers needs (the target pass keeps nothing)
    const int NORM = d.norm;
    const size_t lnw = NORM ? 1 : 0;
    NafLnSave lns, lnone = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    lns.n1 = sc;  sc += lnw * B * L1;
    lns.na = sc;  sc += lnw * B * L2;
    lns.nv = sc;  sc += lnw * B * L2;
    lns.rs1 = sc; sc += lnw * RLC_MAX_BATCH;
    lns.rsa = sc; sc += lnw * RLC_MAX_BATCH;
    lns.rsv = sc; sc += lnw * RLC_MAX_BATCH;
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;

    for (int u = 0; u < n_updates; u++) {
        if (rollout) {
            // on-device experiment loop: one environment step first; update when learn() would run
            if (!rlc_naf_train_step_device(rollout, agent, L.pol)) continue;
        }
        NAF_PHASE();
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
        NAF_PHASE();
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
        // ---- target V'(s') and the float64 TD glue (agents/NAF.py:70) ----
        NAF_PHASE();
        naf_forward(d, tt, L.x2, B, h1, ha, hv, nullptr, L.V, nullptr, nullptr, lnone);
        for (int b = tid; b < B; b += kThreads) {
            const float y = (float)(L.r[b] + L.g[b] * (double)L.V[b]);
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + b] = y;
        }
        __syncthreads();
        // ---- online forward ----
        NAF_PHASE();
        naf_forward(d, th, L.x, B, h1, ha, hv, L.mt, L.V, L.dpre, L.npre, lns);
        // ---- per sample: L columns, advantage, Q, and the seeds of every head's gradient ----
        NAF_PHASE();
        for (int b = tid; b < B; b += kThreads) {
            float diff[RLC_NAF_MAX_A], ddiff[RLC_NAF_MAX_A], tanhv[RLC_NAF_MAX_A];
            for (int j = 0; j < A; j++) {
                tanhv[j] = tanhf(L.mt[b * A + j]);
                diff[j] = L.a[b * A + j] - tanhv[j] * dv.amax[j];
                ddiff[j] = 0.0f;
            }
            float p[RLC_NAF_MAX_A], l0[RLC_NAF_MAX_A];
            float adv = 0.0f;
            int off = 0;
            for (int c = 0; c < A; c++) {
                l0[c] = expf(fminf(fmaxf(L.dpre[b * A + c], -5.0f), 5.0f));
                float pc = diff[c] * l0[c];
                for (int k = 1; k < A - c; k++) pc += diff[c + k] * L.npre[b * NN + off + k - 1];
                off += A - 1 - c;
                p[c] = pc;
                adv += pc * pc;
            }
            const float q = L.V[b] + (-0.5f * adv);
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            dv.tap_V[(size_t)agent * RLC_MAX_BATCH + b] = L.V[b];
            const float dq = 2.0f * (q - L.y[b]);            // loss = SUM (y - q)^2
            L.q[b] = q;
            L.dV[b] = dq;
            off = 0;
            for (int c = 0; c < A; c++) {
                const float dp = -p[c] * dq;
                ddiff[c] += dp * l0[c];
                for (int k = 1; k < A - c; k++) ddiff[c + k] += dp * L.npre[b * NN + off + k - 1];
                const float xpre = L.dpre[b * A + c];
                L.dd[b * A + c] = (xpre >= -5.0f && xpre <= 5.0f) ? dp * diff[c] * l0[c] : 0.0f;
                for (int k = 1; k < A - c; k++) L.dn[b * NN + off + k - 1] = dp * diff[c + k];
                off += A - 1 - c;
            }
            for (int j = 0; j < A; j++) L.dz[b * A + j] = -ddiff[j] * dv.amax[j] * (1.0f - tanhv[j] * tanhv[j]);
        }
        __syncthreads();
        // ---- hidden-layer gradients with the pre-step weights ----
        NAF_PHASE();
        blk_dense_bwd_input_ld(L.dz, A, A, th + d.Wa3, ha, L2, dha, B, false);
        blk_dense_bwd_input_ld(L.dV, 1, 1, th + d.Wv3, hv, L2, dhv, B, false);
        __syncthreads();
        // layer norm: this thread's (gamma, beta) gradient column sums, then dha / dhv / dh1 become the gradients of
        // the layers' linear outputs (pre-step gamma)
        float gga = 0.0f, gba = 0.0f, ggv = 0.0f, gbv = 0.0f, gg1 = 0.0f, gb1 = 0.0f;
        NAF_PHASE();
        if (NORM) {
            blk_layernorm_param_grads(dha, lns.na, L2, B, gga, gba);
            blk_layernorm_param_grads(dhv, lns.nv, L2, B, ggv, gbv);
            __syncthreads();
            blk_layernorm_bwd_rows(dha, lns.na, lns.rsa, th + d.La2g, L2, B);
            blk_layernorm_bwd_rows(dhv, lns.nv, lns.rsv, th + d.Lv2g, L2, B);
            __syncthreads();
        }
        NAF_PHASE();
        blk_dense_bwd_input_ld(dha, L2, L2, th + d.Wa2, h1, L1, dh1, B, false);
        __syncthreads();
        blk_dense_bwd_input_ld(dhv, L2, L2, th + d.Wv2, h1, L1, dh1, B, true);
        __syncthreads();
        NAF_PHASE();
        for (int c = 0; c < A; c++) {
            blk_dense_bwd_input_ld(L.dd + c, A, 1, th + d.Wd[c], h1, L1, dh1, B, true);
            __syncthreads();
        }
        {
            int off = 0;
            for (int c = 0; c < A - 1; c++) {
                blk_dense_bwd_input_ld(L.dn + off, NN, A - 1 - c, th + d.Wn[c], h1, L1, dh1, B, true);
                __syncthreads();
                off += A - 1 - c;
            }
        }
        NAF_PHASE();
        if (NORM) {
            blk_layernorm_param_grads(dh1, lns.n1, L1, B, gg1, gb1);
            __syncthreads();
            blk_layernorm_bwd_rows(dh1, lns.n1, lns.rs1, th + d.L1g, L1, B);
            __syncthreads();
        }
        // ---- gradients + one Adam over every tensor ----
        NAF_PHASE();
        {
            const AdamCtx c = {th, dv.m + (size_t)agent * d.Ppad, dv.v + (size_t)agent * d.Ppad,
                               adam_alpha(dv.lr[agent], pw[0], pw[1]), tapg};
            blk_dense_grad_adam_ld(ha, L2, L2, L.dz, A, A, B, c, d.Wa3, d.ba3);
            blk_dense_grad_adam_ld(hv, L2, L2, L.dV, 1, 1, B, c, d.Wv3, d.bv3);
            NAF_PHASE();
            blk_dense_grad_adam_ld(h1, L1, L1, dha, L2, L2, B, c, d.Wa2, d.ba2);
            blk_dense_grad_adam_ld(h1, L1, L1, dhv, L2, L2, B, c, d.Wv2, d.bv2);
            NAF_PHASE();
            for (int cc = 0; cc < A; cc++) blk_dense_grad_adam_ld(h1, L1, L1, L.dd + cc, A, 1, B, c, d.Wd[cc], d.bd[cc]);
            int off = 0;
            for (int cc = 0; cc < A - 1; cc++) {
                blk_dense_grad_adam_ld(h1, L1, L1, L.dn + off, NN, A - 1 - cc, B, c, d.Wn[cc], d.bn[cc]);
                off += A - 1 - cc;
            }
            NAF_PHASE();
            blk_dense_grad_adam_ld(L.x, S, S, dh1, L1, L1, B, c, d.W1, d.b1);
            if (NORM) {
                if (tid < L2) {
                    adam_apply(c, d.La2g + tid, gga); adam_apply(c, d.La2b + tid, gba);
                    adam_apply(c, d.Lv2g + tid, ggv); adam_apply(c, d.Lv2b + tid, gbv);
                }
                if (tid < L1) { adam_apply(c, d.L1g + tid, gg1); adam_apply(c, d.L1b + tid, gb1); }
            }
        }
        __syncthreads();
        NAF_PHASE();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; }
        for (int p = tid; p < d.Pdev; p += kThreads) {
            const float t = tt[p];
            tt[p] = t + dv.tau * (th[p] - t);
        }
        __syncthreads();
    }
#undef NAF_PHASE
#undef dv
#undef d
}

// greedy action + L columns for one state per agent (predict_action / sample_action's fetch, naf_network.py:144-158)
// done_flag (or null): a word in host-visible memory that receives done_val once the launch's outputs are stored
// (rlc_naf_act_queue; see rlc_ddpg_act_kernel)
__global__ __launch_bounds__(kThreads) void rlc_naf_act_kernel(RlcNafDev dv, int first_agent, const float* states,
                                                               float* mu_out, float* lcols_out, int* done_flag,
                                                               int done_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcNafDims& d = dv.d;
    const int S = d.S, A = d.A, NL = A * (A + 1) / 2;
    const int agent = first_agent + blockIdx.x, tid = threadIdx.x;
    const NafPolicyLds L = naf_policy_carve(d, (float*)smem);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    for (int i = tid; i < S; i += kThreads)
        L.x[i] = clip_state_val(states[(size_t)blockIdx.x * S + i], dv.clip_state, dv.smin[i], dv.smax[i]);
    naf_policy_forward(d, th, L, dv.amax);
    if (tid < A) mu_out[(size_t)blockIdx.x * A + tid] = L.out[tid];
    if (tid == 0 && lcols_out) {
        int p = 0;
        for (int c = 0; c < A; c++)
            for (int i = c; i < A; i++) lcols_out[(size_t)blockIdx.x * NL + p++] = naf_l_entry(d, L, i, c);
    }
    if (done_flag) {                                // (queued forward of a drop-in agent: one workgroup)
        __syncthreads();
        if (tid == 0) {
            __threadfence_system();                     // the output stores first
            __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// evaluation of the on-device loop: one greedy test episode per workgroup (run_episode_eval, experiment.py:163-196)
__global__ __launch_bounds__(kThreads) void rlc_naf_eval_kernel(RlcNafDev dv, RlcEnvDev env, int eval_round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double sim[RLC_ENV_STATE];
    __shared__ double obs[8];
    __shared__ int s_done;
    const RlcNafDims& d = dv.d;
    const int S = d.S;
    const int agent = blockIdx.x / env.eval_episodes, ep = blockIdx.x % env.eval_episodes;
    const int tid = threadIdx.x;
    const NafPolicyLds L = naf_policy_carve(d, (float*)smem);
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
        for (int i = tid; i < S; i += kThreads) L.x[i] = clip_state_val((float)obs[i], dv.clip_state, dv.smin[i], dv.smax[i]);
        naf_policy_forward(d, th, L, dv.amax);
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

size_t rlc_naf_scratch_floats(const RlcNafDims& d) {
    const size_t act = (size_t)d.B * (2 * (size_t)d.L1 + 4 * (size_t)d.L2);
    return d.norm ? act + (size_t)d.B * ((size_t)d.L1 + 2 * (size_t)d.L2) + 3 * RLC_MAX_BATCH : act;
}

int rlc_launch_naf_update(const RlcNafDev& dv, int first_agent, int n_agents, int n_updates, int source,
                          const long long* idx_dev, int grad_taps, hipStream_t st, const RlcNafRollout* rollout) {
    const size_t lds = nlds_carve(dv.d, nullptr, nullptr);
    RLC_REQUIRE(lds <= 64 * 1024, "NAF kernel needs %zu B of LDS (> 64 KiB)", lds);
    hipLaunchKernelGGL(rlc_naf_update_kernel, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source,
                       idx_dev, grad_taps, rollout);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_naf_act(const RlcNafDev& dv, int first_agent, int n, const float* states_dev, float* mu_dev,
                       float* lcols_dev, hipStream_t st, int* done_flag, int done_val) {
    const size_t lds = sizeof(float) * naf_policy_lds_floats(dv.d);
    RLC_REQUIRE(done_flag == nullptr || n == 1, "a completion flag needs a one-workgroup acting launch");
    hipLaunchKernelGGL(rlc_naf_act_kernel, dim3(n), dim3(kThreads), lds, st, dv, first_agent, states_dev, mu_dev,
                       lcols_dev, done_flag, done_val);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_naf_eval(const RlcNafDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st) {
    const size_t lds = sizeof(float) * naf_policy_lds_floats(dv.d);
    hipLaunchKernelGGL(rlc_naf_eval_kernel, dim3(dv.n_agents * env.eval_episodes), dim3(kThreads), lds, st, dv, env,
                       eval_round);
    RLC_HIP(hipGetLastError());
    return 0;
}
