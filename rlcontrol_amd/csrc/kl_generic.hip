// kl_generic.hip -- fused ReverseKL / ForwardKL update + acting kernels (any-shape fp32 VALU path).
//
// One workgroup per agent, n_updates sequential updates per launch; each update = sample_batch
// (utils/replaybuffer.py:32-37) + ReverseKL_Network_Manager.update_network (agents/ReverseKL.py:83-93, the same in
// agents/ForwardKL.py): network.update_network (reversekl_network.py:130-218 / forwardkl_network.py:123-207) followed by
// update_target_network (reversekl_network.py:220-225, the V network only).
// Everything is evaluated from the PRE-update weights (the three losses are built before the first optimizer.step()),
// then Q-Adam, V-Adam, pi-Adam (torch semantics) and the Polyak step of V.
//   networks      pi: s -> L1a -> L2a -> {mean, log_std clamped to [-20, 2]}; Q: [s, a] -> L1c -> L2c -> 1 (action at the
//                 INPUT); V, V': s -> L1c -> L2c -> 1                                  (reversekl_network.py:238-330)
//   sampled z     mean + std * eps, no gradient through the draw (normal.sample()), log pi = N(z) - sum log(1 - tanh(z)^2 + 1e-6)
//   action_dim>1  get_distribution builds MultivariateNormal(mean, diag_embed(std)) (reversekl_network.py:383-389): the
//                 COVARIANCE is diag(std), so component j has variance std_j (not std_j^2): z_j = mean_j + sqrt(std_j) eps_j,
//                 log N = -sum (z-mean)^2 / (2 std) - sum log sqrt(std) - A log sqrt(2 pi).  Reproduced as written.
//   action integral (optim_type intg / hard_intg): Q at the B x K pairs (state_b, node_k) -- the one large contraction of
//                 the update, [B*K, L1c] x [L1c, L2c] -- and log pi(node_k | state_b) through atanh (get_logprob,
//                 reversekl_network.py:360-381).  The nodes [K, A] and weights [K] come from the host: the Clenshaw-Curtis
//                 line rule for action_dim 1, the sparse grid of reversekl_network.py:78-108 above it (weights of either sign)
//     reverse     loss_b = sum_k w_k * -exp(lp) * ((Q_bk - V_b) - alpha * lp)     (hard: without the alpha * lp term)
//     forward     loss_b = -sum_k w_k * softmax_k(Q_bk / alpha) * lp, softmax normalised with the quadrature weights
//   ll / hard_ll  loss_b = -lp_b * (Q(s, a_new) - V - alpha * lp_b)                 (hard: without the alpha * lp term)
// The input normaliser handed to the networks is never applied by them (reversekl_network.py:43): states enter raw.
// r and gamma are cast to fp32 (torch.FloatTensor(reward_batch), reversekl_network.py:135-136).
#include "generic_blocks.h"
#include "sac_rollout_device.h"
#include "../../include/rlcontrol_hip.h"

namespace {

using namespace gen;

#define RLC_KL_MAX_A 6

struct KLds {
    // [B, A]: a, eps, mu, lsr (raw log_std head), vr (variance), lsq (log of its square root), z, newa, dmu, dls; [B]: the rest
    float *x, *x2, *a, *eps, *mu, *lsr, *vr, *lsq, *z, *newa, *lp, *r, *g, *q, *v, *vt, *qn, *dq, *dvs, *dmu, *dls, *red, *adam;
    long long* idx;
    int* pool;
    int* dups;
    float* pol;       // scratch of the on-device training step (sac_rollout_device.h)
};

__host__ __device__ inline size_t klds_carve(const RlcSacDims& d, unsigned char* base, KLds* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        unsigned char* p = base ? base + off : nullptr;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int B = d.B, S = d.S, A = d.A;
    KLds L;
    L.idx = (long long*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.x = (float*)take(sizeof(float) * B * S);
    L.x2 = (float*)take(sizeof(float) * B * S);
    float** pa[] = {&L.a, &L.eps, &L.mu, &L.lsr, &L.vr, &L.lsq, &L.z, &L.newa, &L.dmu, &L.dls};
    for (auto p : pa) *p = (float*)take(sizeof(float) * B * A);
    float** pb[] = {&L.lp, &L.r, &L.g, &L.q, &L.v, &L.vt, &L.qn, &L.dq, &L.dvs};
    for (auto p : pb) *p = (float*)take(sizeof(float) * B);
    L.red = (float*)take(sizeof(float) * 16);
    L.adam = (float*)take(sizeof(float) * 4);
    L.pool = (int*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (int*)take(sizeof(int) * 4);
    L.pol = (float*)take(sizeof(float) * (sac_policy_lds_floats(d) + 4));
    if (out) *out = L;
    return off;
}

// rows of the B x K action integral processed per pass (bounds the scratch when N_param is large)
__host__ __device__ inline int kl_chunk_rows(int rows) { return rows < 2048 ? rows : 2048; }

__global__ __launch_bounds__(kThreads) void rlc_kl_update_kernel(RlcSacDev dv_arg, int first_agent, int n_updates,
                                                                 int source, const long long* host_idx,
                                                                 const float* eps_in, int grad_taps,
                                                                 const RlcSacRollout* rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the population view is read through gen::kernarg_view (generic_blocks.h), made opaque again by KL_PHASE() at the
    // start of every phase (dv_arg is the first argument: offset 0)
    const RlcSacDev* dvp;
#define KL_PHASE() (dvp = kernarg_view<RlcSacDev>())
#define dv (*dvp)
#define d (dvp->d)
    KL_PHASE();
    const int S = d.S, A = d.A, L1A = d.L1A, L2A = d.L2A, L1C = d.L1C, L2C = d.L2C, B = d.B, K = dv.kl_nodes;
    const int agent = first_agent + blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    KLds L;
    klds_carve(d, smem, &L);
    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* mm = dv.m + (size_t)agent * d.Ppad;
    float* vv = dv.v + (size_t)agent * d.Ppad;
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;
    const float alpha_ent = dv.alpha[agent], amax0 = dv.amax0;
    const bool integral = dv.kl_optim == RLC_KL_OPTIM_INTG || dv.kl_optim == RLC_KL_OPTIM_HARD_INTG;
    const int rows = B * K, chunk = kl_chunk_rows(rows);
    // scratch carve
    float* sc = dv.scratch + (size_t)agent * dv.scratch_stride;
    auto carve = [&](size_t n) { float* p = sc; sc += (n + 3) & ~(size_t)3; return p; };
    float* ph1 = carve((size_t)B * L1A);  float* ph2 = carve((size_t)B * L2A);
    float* qh1 = carve((size_t)B * L1C);  float* qh2 = carve((size_t)B * L2C);
    float* nh1 = carve((size_t)B * L1C);  float* nh2 = carve((size_t)B * L2C);
    float* vh1 = carve((size_t)B * L1C);  float* vh2 = carve((size_t)B * L2C);
    float* th1 = carve((size_t)B * L1C);  float* th2 = carve((size_t)B * L2C);
    float* dp2 = carve((size_t)B * L2A);  float* dp1 = carve((size_t)B * L1A);
    float* dq2 = carve((size_t)B * L2C);  float* dq1 = carve((size_t)B * L1C);
    float* dv2 = carve((size_t)B * L2C);  float* dv1 = carve((size_t)B * L1C);
    float* z1s = carve((size_t)B * L1C);
    float* iq = carve((size_t)rows);
    float* ih1 = carve((size_t)chunk * L1C);
    float* ih2 = carve((size_t)chunk * L2C);
    const float LOG_SQRT_2PI = 0.9189385332046727f, EPS = 1e-6f;
    const float invB = 1.0f / (float)B;

    for (int u = 0; u < n_updates; u++) {
        if (rollout) {
            // on-device experiment loop: one environment step first; update when learn() would run
            if (!rlc_sac_train_step_device(rollout, agent, L.pol)) continue;
        }
        // ---- sample + gather ----
        KL_PHASE();
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
        const int step = dv.kl_step[agent] + 1;
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
            for (int i = 0; i < S; i++) { L.x[b * S + i] = ps[i]; L.x2[b * S + i] = ps2[i]; }
            for (int j = 0; j < A; j++) {
                const int e_at = b * A + j;
                L.a[e_at] = pa[j];
                float e;
                if (eps_in) {
                    e = eps_in[((size_t)blockIdx.x * n_updates + u) * B * A + e_at];
                } else {
                    const Philox4 p = philox4x32_10(dv.rep.seed[agent] ^ RLC_KEY_SAC_EPS, nctr, (unsigned long long)e_at >> 1);
                    float n0, n1;
                    philox_normal2(p, n0, n1);
                    e = (e_at & 1) ? n1 : n0;
                }
                L.eps[e_at] = e;
            }
        }
        if (tid == 0) {
            // torch's Adam: theta -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + 1e-8); with c = sqrt(1 - b2^t)
            // that is the TF-form step of generic_blocks.h with alpha = lr * c / (1 - b1^t) and epsilon = 1e-8 * c
            const double c = sqrt(1.0 - pow(0.999, (double)step)), bc1 = 1.0 - pow(0.9, (double)step);
            L.adam[0] = (float)((double)dv.pi_lr[agent] * c / bc1);
            L.adam[1] = (float)((double)dv.qv_lr[agent] * c / bc1);
            L.adam[2] = (float)(1e-8 * c);
        }
        __syncthreads();
        if (tid == 0 && !eps_in) dv.noise_ctr[agent] = nctr + 1;

        // ---- forward: pi, Q(s,a), V(s), V'(s') ----
        KL_PHASE();
        blk_dense(L.x, S, S, nullptr, 0, th + d.pW1, th + d.pb1, L1A, ph1, L1A, B, 1);
        blk_dense(L.x, S, S, L.a, A, th + d.qW1, th + d.qb1, L1C, qh1, L1C, B, 1);
        blk_dense(L.x, S, S, nullptr, 0, th + d.vW1, th + d.vb1, L1C, vh1, L1C, B, 1);
        blk_dense(L.x2, S, S, nullptr, 0, tt + d.vW1, tt + d.vb1, L1C, th1, L1C, B, 1);
        if (integral) blk_dense(L.x, S, S, nullptr, 0, th + d.qW1, th + d.qb1, L1C, z1s, L1C, B, 0);
        __syncthreads();
        blk_dense(ph1, L1A, L1A, nullptr, 0, th + d.pW2, th + d.pb2, L2A, ph2, L2A, B, 1);
        blk_dense(qh1, L1C, L1C, nullptr, 0, th + d.qW2, th + d.qb2, L2C, qh2, L2C, B, 1);
        blk_dense(vh1, L1C, L1C, nullptr, 0, th + d.vW2, th + d.vb2, L2C, vh2, L2C, B, 1);
        blk_dense(th1, L1C, L1C, nullptr, 0, tt + d.vW2, tt + d.vb2, L2C, th2, L2C, B, 1);
        __syncthreads();
        blk_dense(ph2, L2A, L2A, nullptr, 0, th + d.pWm, th + d.pbm, A, L.mu, A, B, 0);
        blk_dense(ph2, L2A, L2A, nullptr, 0, th + d.pWs, th + d.pbs, A, L.lsr, A, B, 0);
        blk_dense(qh2, L2C, L2C, nullptr, 0, th + d.qW3, th + d.qb3, 1, L.q, 1, B, 0);
        blk_dense(vh2, L2C, L2C, nullptr, 0, th + d.vW3, th + d.vb3, 1, L.v, 1, B, 0);
        blk_dense(th2, L2C, L2C, nullptr, 0, tt + d.vW3, tt + d.vb3, 1, L.vt, 1, B, 0);
        __syncthreads();
        // evaluate(): draw, squash, log-density of the draw (reversekl_network.py:332-357)
        KL_PHASE();
        for (int b = tid; b < B; b += kThreads) {
            float quad = 0.0f, corr = 0.0f;
            for (int j = 0; j < A; j++) {
                const int at = b * A + j;
                const float ls = fminf(fmaxf(L.lsr[at], -20.0f), 2.0f);
                const float sd = expf(ls), mu = L.mu[at];
                // Normal(mean, std) for one action dimension; MultivariateNormal with covariance diag(std) above it
                const float var = A == 1 ? sd * sd : sd, sq = A == 1 ? sd : sqrtf(sd);
                const float z = mu + sq * L.eps[at];
                const float t = tanhf(z);
                const float dz = z - mu;
                const float lsq = logf(sq);
                quad += -(dz * dz) / (2.0f * var) - lsq;
                corr += logf(1.0f - t * t + EPS);
                L.vr[at] = var; L.lsq[at] = lsq; L.z[at] = z; L.newa[at] = t * amax0;
            }
            const float lp = quad - (float)A * LOG_SQRT_2PI - corr;
            L.lp[b] = lp;
            dv.tap_logp[(size_t)agent * RLC_MAX_BATCH + b] = lp;
        }
        __syncthreads();
        // Q(s, a_new): the V target of q_update_type 'sac' and the advantage of the ll updates
        KL_PHASE();
        blk_dense(L.x, S, S, L.newa, A, th + d.qW1, th + d.qb1, L1C, nh1, L1C, B, 1);
        __syncthreads();
        blk_dense(nh1, L1C, L1C, nullptr, 0, th + d.qW2, th + d.qb2, L2C, nh2, L2C, B, 1);
        __syncthreads();
        blk_dense(nh2, L2C, L2C, nullptr, 0, th + d.qW3, th + d.qb3, 1, L.qn, 1, B, 0);
        __syncthreads();

        // ---- value seeds (MSELoss: mean over the B x 1 outputs) and their losses ----
        KL_PHASE();
        float ql = 0.0f, vl = 0.0f, pl = 0.0f;
        for (int b = tid; b < B; b += kThreads) {
            const float tq = L.r[b] + L.g[b] * L.vt[b];
            const float tv = dv.kl_qupdate == RLC_KL_Q_SAC ? L.qn[b] - alpha_ent * L.lp[b]
                                                           : (L.r[b] - alpha_ent * L.lp[b]) + L.g[b] * L.vt[b];
            const float eq = L.q[b] - tq, ev = L.v[b] - tv;
            ql += eq * eq; vl += ev * ev;
            L.dq[b] = 2.0f * eq * invB;
            L.dvs[b] = 2.0f * ev * invB;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = L.q[b];
            dv.tap_v[(size_t)agent * RLC_MAX_BATCH + b] = L.v[b];
            dv.tap_qpi[(size_t)agent * RLC_MAX_BATCH + b] = L.qn[b];
            if (!integral) {
                // -log_prob * (advantage).detach(): the gradient reaches mean / log_std through N(z; mean, std) only
                const float adv = (L.qn[b] - L.v[b]) - (dv.kl_optim == RLC_KL_OPTIM_LL ? alpha_ent * L.lp[b] : 0.0f);
                const float coef = -adv * invB;
                pl += -L.lp[b] * adv;
                for (int j = 0; j < A; j++) {
                    const int at = b * A + j;
                    const float dz = L.z[at] - L.mu[at], var = L.vr[at];
                    const bool inside = L.lsr[at] >= -20.0f && L.lsr[at] <= 2.0f;
                    L.dmu[at] = coef * dz / var;
                    // d log N / d log_std: variance std^2 (one dimension) or std (the diag(std) covariance)
                    L.dls[at] = !inside ? 0.0f : A == 1 ? coef * (dz * dz / var - 1.0f) : coef * (dz * dz / (2.0f * var) - 0.5f);
                }
            }
        }
        __syncthreads();

        if (integral) {
            // ---- Q at the quadrature nodes: rows rho = b*K + k, layer 1 = relu(z1s[b] + a_k . W1[action rows]) ----
            for (int r0 = 0; r0 < rows; r0 += chunk) {
                KL_PHASE();
                const float* w1a = th + d.qW1 + (size_t)S * L1C;
                const int nr = min(chunk, rows - r0);
                for (int it = tid; it < nr * L1C; it += kThreads) {
                    const int rho = r0 + it / L1C, n = it % L1C;
                    float acc = z1s[(size_t)(rho / K) * L1C + n];
                    for (int j = 0; j < A; j++) acc += dv.kl_node_a[(rho % K) * A + j] * w1a[(size_t)j * L1C + n];
                    ih1[it] = fmaxf(acc, 0.0f);
                }
                __syncthreads();
                blk_dense(ih1, L1C, L1C, nullptr, 0, th + d.qW2, th + d.qb2, L2C, ih2, L2C, nr, 1);
                __syncthreads();
                const float b3 = th[d.qb3];
                for (int r = wave; r < nr; r += kThreads / 64) {
                    float acc = 0.0f;
                    for (int n = lane; n < L2C; n += 64) acc += ih2[(size_t)r * L2C + n] * th[d.qW3 + n];
                    acc = wave_sum64(acc);
                    if (lane == 0) {
                        iq[r0 + r] = acc + b3;
                        dv.kl_tap_iq[(size_t)agent * rows + r0 + r] = acc + b3;
                    }
                }
                __syncthreads();
            }
            // ---- one wave per state: log pi at the nodes, the integrand's derivative, seeds of mean and log_std ----
            KL_PHASE();
            for (int b = wave; b < B; b += kThreads / 64) {
                const float vb = L.v[b];
                float shift = -INFINITY, zsum = 0.0f;
                if (dv.kl_kind == RLC_KL_FORWARD) {
                    for (int k = lane; k < K; k += 64) shift = fmaxf(shift, iq[b * K + k] / alpha_ent);
                    for (int off = 32; off > 0; off >>= 1) shift = fmaxf(shift, __shfl_xor(shift, off, 64));
                    for (int k = lane; k < K; k += 64) zsum += expf(iq[b * K + k] / alpha_ent - shift) * dv.kl_node_w[k];
                    zsum = wave_sum64(zsum);
                    zsum = __shfl(zsum, 0, 64);
                }
                float gm[RLC_KL_MAX_A], gs[RLC_KL_MAX_A], loss = 0.0f;
#pragma unroll
                for (int j = 0; j < RLC_KL_MAX_A; j++) { gm[j] = 0.0f; gs[j] = 0.0f; }
                for (int k = lane; k < K; k += 64) {
                    const float w = dv.kl_node_w[k];
                    float quad = 0.0f, corr = 0.0f, du[RLC_KL_MAX_A];
#pragma unroll
                    for (int j = 0; j < RLC_KL_MAX_A; j++) {
                        if (j < A) {
                            const float an = dv.kl_node_a[k * A + j] / amax0;
                            const float uu = (logf(1.0f + an) - logf(1.0f - an)) / 2.0f;
                            du[j] = uu - L.mu[b * A + j];
                            quad += -(du[j] * du[j]) / (2.0f * L.vr[b * A + j]) - L.lsq[b * A + j];
                            corr += logf(1.0f - an * an + EPS);
                        }
                    }
                    const float lp = quad - (float)A * LOG_SQRT_2PI - corr;
                    float coef;   // d loss_b / d lp_k
                    if (dv.kl_kind == RLC_KL_FORWARD) {
                        const float bp = expf(iq[b * K + k] / alpha_ent - shift) / zsum;
                        loss += -(bp * lp) * w;
                        coef = -(bp * w);
                    } else {
                        const float adv = iq[b * K + k] - vb, e = expf(lp);
                        if (dv.kl_optim == RLC_KL_OPTIM_INTG) {
                            loss += (-e * (adv - alpha_ent * lp)) * w;
                            coef = -e * (adv - alpha_ent * lp - alpha_ent) * w;
                        } else {
                            loss += (-e * adv) * w;
                            coef = -e * adv * w;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RLC_KL_MAX_A; j++) {
                        if (j < A) {
                            const float var = L.vr[b * A + j];
                            gm[j] += coef * (du[j] / var);
                            gs[j] += A == 1 ? coef * (du[j] * du[j] / var - 1.0f) : coef * (du[j] * du[j] / (2.0f * var) - 0.5f);
                        }
                    }
                }
                loss = wave_sum64(loss);
#pragma unroll
                for (int j = 0; j < RLC_KL_MAX_A; j++) {
                    if (j < A) {
                        const float m = wave_sum64(gm[j]), sg = wave_sum64(gs[j]);
                        if (lane == 0) {
                            const bool inside = L.lsr[b * A + j] >= -20.0f && L.lsr[b * A + j] <= 2.0f;
                            L.dmu[b * A + j] = m * invB;
                            L.dls[b * A + j] = inside ? sg * invB : 0.0f;
                        }
                    }
                }
                if (lane == 0) L.lp[b] = loss;          // per-state loss; lp itself is already tapped and consumed
            }
            __syncthreads();
            for (int b = tid; b < B; b += kThreads) pl += L.lp[b];
        }
        ql = blk_sum(ql, L.red); vl = blk_sum(vl, L.red); pl = blk_sum(pl, L.red);
        if (tid == 0) {
            dv.tap_loss[agent * 4 + 0] = pl * invB;
            dv.tap_loss[agent * 4 + 1] = ql * invB;
            dv.tap_loss[agent * 4 + 2] = vl * invB;
        }

        // ---- hidden-layer gradients, all with the pre-update weights ----
        KL_PHASE();
        blk_dense_bwd_input_ex(L.dmu, A, th + d.pWm, ph2, L2A, dp2, B, false);
        for (int it = tid; it < B * L2C; it += kThreads) {
            const int b = it / L2C, n = it % L2C;
            dq2[it] = qh2[it] > 0.0f ? L.dq[b] * th[d.qW3 + n] : 0.0f;
            dv2[it] = vh2[it] > 0.0f ? L.dvs[b] * th[d.vW3 + n] : 0.0f;
        }
        __syncthreads();
        blk_dense_bwd_input_ex(L.dls, A, th + d.pWs, ph2, L2A, dp2, B, true);
        blk_dense_bwd_input(dq2, L2C, th + d.qW2, qh1, L1C, dq1, B);
        blk_dense_bwd_input(dv2, L2C, th + d.vW2, vh1, L1C, dv1, B);
        __syncthreads();
        blk_dense_bwd_input(dp2, L2A, th + d.pW2, ph1, L1A, dp1, B);
        __syncthreads();
        // ---- gradients + Adam: q_optimizer, v_optimizer, pi_optimizer (disjoint parameters) ----
        KL_PHASE();
        {
            const AdamCtx cq = {th, mm, vv, L.adam[1], tapg, L.adam[2]};
            blk_dense_grad_adam(qh2, L2C, L2C, nullptr, 0, L.dq, 1, B, cq, d.qW3, d.qb3);
            blk_dense_grad_adam(qh1, L1C, L1C, nullptr, 0, dq2, L2C, B, cq, d.qW2, d.qb2);
            blk_dense_grad_adam(L.x, S, S, L.a, A, dq1, L1C, B, cq, d.qW1, d.qb1);
            blk_dense_grad_adam(vh2, L2C, L2C, nullptr, 0, L.dvs, 1, B, cq, d.vW3, d.vb3);
            blk_dense_grad_adam(vh1, L1C, L1C, nullptr, 0, dv2, L2C, B, cq, d.vW2, d.vb2);
            blk_dense_grad_adam(L.x, S, S, nullptr, 0, dv1, L1C, B, cq, d.vW1, d.vb1);
            KL_PHASE();
            const AdamCtx cp = {th, mm, vv, L.adam[0], tapg, L.adam[2]};
            blk_dense_grad_adam(ph2, L2A, L2A, nullptr, 0, L.dmu, A, B, cp, d.pWm, d.pbm);
            blk_dense_grad_adam(ph2, L2A, L2A, nullptr, 0, L.dls, A, B, cp, d.pWs, d.pbs);
            blk_dense_grad_adam(ph1, L1A, L1A, nullptr, 0, dp2, L2A, B, cp, d.pW2, d.pb2);
            blk_dense_grad_adam(L.x, S, S, nullptr, 0, dp1, L1A, B, cp, d.pW1, d.pb1);
        }
        __syncthreads();
        if (tid == 0) dv.kl_step[agent] = step;
        // ---- update_target_network: the V network only, target*(1-tau) + param*tau ----
        KL_PHASE();
        // target_param * (1 - tau) + param * tau as torch evaluates it (reversekl_network.py:222-225): two separately
        // rounded products and one sum -- no fused multiply-add
        for (int p = d.vW1 + tid; p < d.Pdev; p += kThreads)
            tt[p] = __fadd_rn(__fmul_rn(tt[p], 1.0f - dv.tau), __fmul_rn(th[p], dv.tau));
        __syncthreads();
    }
#undef KL_PHASE
#undef dv
#undef d
}

// predict_action (tanh(mean) * action_max) / sample_action (tanh(mean + std*eps) * action_max) for one state per
// agent (reversekl_network.py:111-128): one workgroup per agent
__global__ __launch_bounds__(kThreads) void rlc_kl_act_kernel(RlcSacDev dv, int first_agent, const float* states,
                                                              const float* eps_in, int sample, float* out,
                                                              int* done_flag, int done_val) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcSacDims d = dv.d;
    const int S = d.S, A = d.A;
    const int agent = first_agent + blockIdx.x, tid = threadIdx.x;
    const SacPolicyLds L = sac_policy_carve(d, (float*)smem);
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    for (int i = tid; i < S; i += kThreads) L.x[i] = states[(size_t)blockIdx.x * S + i];
    if (sample && tid < A)
        L.eps[tid] = eps_in ? eps_in[(size_t)blockIdx.x * A + tid] : sac_act_eps(dv.rep.seed[agent], dv.noise_ctr[agent], tid);
    sac_policy_forward(d, th, L, dv.amax0, sample, 1);
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

}  // namespace

size_t rlc_kl_scratch_floats(const RlcSacDims& d, int nodes) {
    const size_t B = d.B, rows = B * (size_t)nodes, chunk = kl_chunk_rows((int)rows);
    auto r4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
    return 2 * r4(B * d.L1A) + 2 * r4(B * d.L2A) + 6 * r4(B * d.L1C) + 6 * r4(B * d.L2C) + r4(B * d.L1C) + r4(rows) +
           r4(chunk * d.L1C) + r4(chunk * d.L2C);
}

int rlc_launch_kl_update(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                         const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                         const RlcSacRollout* rollout) {
    const size_t lds = klds_carve(dv.d, nullptr, nullptr);
    RLC_REQUIRE(dv.d.A >= 1 && dv.d.A <= RLC_KL_MAX_A && dv.d.qcat == 1,
                "the KL update kernel needs action_dim in [1,%d] and the input-concatenated Q layout", RLC_KL_MAX_A);
    const bool integral = dv.kl_optim == RLC_KL_OPTIM_INTG || dv.kl_optim == RLC_KL_OPTIM_HARD_INTG;
    RLC_REQUIRE(!integral || dv.kl_nodes >= 1, "no quadrature nodes");
    RLC_REQUIRE(lds <= 64 * 1024, "KL kernel needs %zu B of LDS", lds);
    RLC_REQUIRE(!(rollout && eps_dev), "the on-device loop draws its own eps");
    hipLaunchKernelGGL(rlc_kl_update_kernel, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates,
                       source, idx_dev, eps_dev, grad_taps, rollout);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_kl_act(const RlcSacDev& dv, int first_agent, int n, const float* states_dev, const float* eps_dev,
                      int sample, float* out_dev, hipStream_t st, int* done_flag, int done_val) {
    const size_t lds = sizeof(float) * sac_policy_lds_floats(dv.d);
    RLC_REQUIRE(done_flag == nullptr || n == 1, "a completion flag needs a one-workgroup acting launch");
    hipLaunchKernelGGL(rlc_kl_act_kernel, dim3(n), dim3(kThreads), lds, st, dv, first_agent, states_dev, eps_dev,
                       sample, out_dev, done_flag, done_val);
    RLC_HIP(hipGetLastError());
    return 0;
}
