// sac_mfma_kernel.h -- fused SoftActorCritic (SAC-v1) update on gfx950 fp32 matrix cores.
//
// Same contract as sac_generic.hip (one workgroup per agent, n_updates sequential updates per launch, every update
// = sample_batch + ONE Session.run(train_ops) of agents/network/sac_network.py:107-136 + update_target_network,
// agents/SoftActorCritic.py:113-126), built from the MFMA blocks of mfma_blocks.h.  Everything the reference
// evaluates in that run comes from the PRE-update weights; the three optimizer scopes (pi | qf | vf) own disjoint
// parameters, so each network is stepped as soon as nothing else needs its old weights:
//
//   1  V'(s')                trunk(vf', clip s') -> GEMM vW2' -> v_targ
//   2  Q hidden contraction  trunk(qf, raw s)    -> GEMM qW2[:L1C]  (shared by Q(s,a) and Q(s,pi): they differ only
//                                                   in the action rows of the concat layer); accumulators parked
//   3  pi forward            trunk(pi, clip s)   -> GEMM pW2 -> mu / log_std heads -> sample, logp, tanh squash
//   4  Q heads               Q(s,pi), dQ/da (from the parked accumulators, untouched), then Q(s,a) in place
//   5  pi step               GEMM (d h1) -> pW1/pb1, GEMM (d pW2) with Adam + Polyak in the epilogue, heads
//   6  Q step                trunk(qf) again -> GEMM (d h1) -> qW1/qb1, GEMM (d qW2) + Adam + Polyak, heads
//   7  V step                trunk(vf) -> GEMM vW2 -> v -> GEMM (d h1), GEMM (d vW2) + Adam + Polyak, heads
//
// 10 contractions of [B, L1] x [L1, L2] per update.  Reference quirks kept (oracle/sac_oracle.c derives them):
//   Q9   logp_pi is [B] while q_pi, v are [B,1]: v regresses onto q_pi[i] - alpha*mean_j(logp[j]);
//   the state clip of pi and V uses the SCALARS state_min[0]/state_max[0]; Q sees the raw state;
//   actions are scaled by action_max[0]; gaussian_likelihood divides by std + 1e-6; the tanh-squash
//   correction is log(clip(1 - pi^2, 0, 1) + 1e-6) with the clip passing gradients;
//   r and gamma enter fp32 placeholders (sac_network.py:51-52): the replay's float64 values are cast at gather time.
//
// Supported shapes: S <= 8, A in {1,2}, layer widths multiples of 4 in [16, 128*NTW], B <= 128, LDS <= 160 KiB.
#pragma once
#include "mfma_blocks.h"
#include "sac_rollout_device.h"
#include "sac_common.h"

namespace {

using namespace mfb;

struct SSmem {
    lds_f32* hbuf;
    lds_u8* mask;        // bit 0: pi hidden (later V hidden), bit 1: Q(s,a) hidden
    lds_f32* part_h;     // [kWaves][MB][2A]   mu | log_std head partials (later: the V head, stride 1)
    lds_f32* part_p;     // [kWaves][MB][1+A]  Q(s,pi) | dQ/da partials
    lds_f32* part_q;     // [kWaves][MB][1]    Q(s,a) (also V'(s'))
    lds_f32* wvec;       // [2A][256] staged output-layer weights
    lds_f32 *x, *xc, *x2c;                          // [MB][SMAX] raw s, clipped s, clipped s'
    lds_f32 *a, *api, *eps, *pit, *sd, *t;          // [MB][A]
    lds_f32* dml;                                   // [MB][2A] seeds of the pi heads: d mu_raw | d log_std_pre
    lds_f32 *r, *g, *vt, *q, *qpi, *v, *logp, *dout, *dvs;    // [MB]
    lds_f32* red;        // 16
    lds_i64* idx;
    lds_i32* pool;
    lds_i32* dups;
    lds_f32* pol;        // scratch of the on-device training step (sac_rollout_device.h)
    lds_f32 *w1vt, *w1q, *w1p, *w1v;     // first layers staged in LDS: vf target, qf, pi, vf (mfma_blocks.h stage_*)
};

__host__ __device__ inline int sac_mfma_ldh(const RlcSacDims& d) { return ldh_for(d.L1A > d.L1C ? d.L1A : d.L1C); }

template <int MSTRIDE>
__host__ __device__ inline size_t ssmem_carve(const RlcSacDims& d, int MT, lds_u8* base, SSmem* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        lds_u8* p = base + off;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int MB = MT * 16, A = d.A, LDH = sac_mfma_ldh(d);
    SSmem L;
    L.hbuf = (lds_f32*)take(sizeof(float) * (MB * LDH + 16));
    L.idx = (lds_i64*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.mask = take((size_t)MB * MSTRIDE);
    L.part_h = (lds_f32*)take(sizeof(float) * kWaves * MB * 2 * A);
    L.part_p = (lds_f32*)take(sizeof(float) * kWaves * MB * (1 + A));
    L.part_q = (lds_f32*)take(sizeof(float) * kWaves * MB);
    L.wvec = (lds_f32*)take(sizeof(float) * 2 * A * 256);
    lds_f32** ps[] = {&L.x, &L.xc, &L.x2c};
    for (auto p : ps) *p = (lds_f32*)take(sizeof(float) * MB * SMAX);
    lds_f32** pa[] = {&L.a, &L.api, &L.eps, &L.pit, &L.sd, &L.t};
    for (auto p : pa) *p = (lds_f32*)take(sizeof(float) * MB * A);
    L.dml = (lds_f32*)take(sizeof(float) * MB * 2 * A);
    lds_f32** pb[] = {&L.r, &L.g, &L.vt, &L.q, &L.qpi, &L.v, &L.logp, &L.dout, &L.dvs};
    for (auto p : pb) *p = (lds_f32*)take(sizeof(float) * MB);
    L.red = (lds_f32*)take(sizeof(float) * 16);
    L.pool = (lds_i32*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (lds_i32*)take(sizeof(int) * 4);
    L.pol = (lds_f32*)take(sizeof(float) * (sac_policy_lds_floats(d) + 4));
#ifdef RLC_W1_STAGE
    L.w1vt = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.L1C);
    L.w1q = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.L1C);
    L.w1p = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.L1A);
    L.w1v = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.L1C);
#else
    L.w1vt = L.w1q = L.w1p = L.w1v = nullptr;
#endif
    if (out) *out = L;
    return off;
}

// block-wide sum of v over threads (fixed order: deterministic); result broadcast to all threads
__device__ inline float sac_blk_sum(float v, lds_f32* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    lds_barrier();
    float s = 0.0f;
    for (int w = 0; w < kWaves; w++) s += red[w];
    lds_barrier();
    return s;
}

// two block-wide sums in one pass (one exchange, three barriers instead of six)
__device__ inline void sac_blk_sum2(float a, float b, lds_f32* red, float& sa, float& sb) {
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    lds_barrier();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[8 + (threadIdx.x >> 6)] = b; }
    lds_barrier();
    sa = 0.0f; sb = 0.0f;
    for (int w = 0; w < kWaves; w++) { sa += red[w]; sb += red[8 + w]; }
    lds_barrier();
}

// T4: the minibatch ends within the first four rows of its last tile (mfma_blocks.h, Blk's T4; the launcher checks it)
template <int MT, int NTW, int AD, bool T4>
__global__ __launch_bounds__(kThreads) void rlc_sac_update_mfma_kernel(RlcSacDev dv, int first_agent, int n_updates,
                                                                       int source, const long long* host_idx,
                                                                       const float* eps_in, int grad_taps,
                                                                       const RlcSacRollout* rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MSTRIDE = mask_stride(8 * NTW);
    constexpr int NS = 2 * AD;
    using U = Blk<MT, NTW, MSTRIDE, true, false, T4>;
    constexpr int MB = U::MB;
    const RlcSacDims d = dv.d;
    U u;
    u.init_geometry();
    u.S = d.S; u.H1 = d.L1A; u.B = d.B; u.LDH = sac_mfma_ldh(d);
    SSmem L;
    ssmem_carve<MSTRIDE>(d, MT, (lds_u8*)smem, &L);
    u.L.hbuf = L.hbuf; u.L.mask = L.mask;
    const int tid = u.tid, S = d.S, L1A = d.L1A, L2A = d.L2A, L1C = d.L1C, L2C = d.L2C, B = d.B;
    const int agent = first_agent + blockIdx.x;

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* mm = dv.m + (size_t)agent * d.Ppad;
    float* vv = dv.v + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 4;
#ifdef RLC_STAMPS
    float* stamp_buf = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;   // diagnostic build: no gradient taps
    float* tapg = nullptr;
#else
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;
#endif
    const float alpha_ent = dv.alpha[agent], amax0 = dv.amax0, tau = dv.tau;
    const float EPS = 1e-6f, LOG2PI = 1.8378770664093453f, HALF_RANGE = 0.5f * (2.0f - (-20.0f));
    const float invB = 1.0f / (float)B;

    // zero the padded tails of the per-sample vectors once (rows >= B never change afterwards)
    for (int i = tid; i < MB * AD; i += kThreads) { L.a[i] = 0.f; L.api[i] = 0.f; L.eps[i] = 0.f; L.pit[i] = 0.f; L.sd[i] = 0.f; L.t[i] = 0.f; }
    for (int i = tid; i < MB * NS; i += kThreads) L.dml[i] = 0.f;
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.xc[i] = 0.f; L.x2c[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) {
        L.r[i] = 0.f; L.g[i] = 0.f; L.vt[i] = 0.f; L.q[i] = 0.f; L.qpi[i] = 0.f; L.v[i] = 0.f; L.logp[i] = 0.f;
        L.dout[i] = 0.f; L.dvs[i] = 0.f;
    }
    for (int i = tid; i < MB * MSTRIDE / 4; i += kThreads) reinterpret_cast<lds_u32*>(L.mask)[i] = 0u;
    if (tid < 16) L.hbuf[MB * u.LDH + tid] = 0.0f;
    __syncthreads();

    stagger_start();
    f32x4 acc[MT][NTW];
#ifdef RLC_STAMPS
    // diagnostic build only: phase boundaries in shader cycles (scripts/phase_stamps_sac.py), written where the gradient tap lives
    long long t_prev = clock64();
    int stamp_i = 0;
#define STAMP()                                                                                  \
    do {                                                                                         \
        if (tid == 0 && stamp_buf) { const long long t = clock64(); stamp_buf[stamp_i] += (float)(t - t_prev); t_prev = t; } \
        stamp_i++;                                                                               \
    } while (0)
    if (stamp_buf) for (int i = tid; i < 64; i += kThreads) stamp_buf[i] = 0.0f;
    u.stamp_buf = stamp_buf;
    __syncthreads();
#else
#define STAMP() do {} while (0)
#endif
    for (int upd = 0; upd < n_updates; upd++) {
#ifdef RLC_STAMPS
        stamp_i = 0;
        if (tid == 0) t_prev = clock64();
#endif
        asm volatile("" : "+v"(u.c), "+v"(u.g), "+s"(u.wave));     // see ddpg_mfma_kernel.h
        if (rollout) {
            // on-device experiment loop: one environment step first; update when learn() would run
            if (!rlc_sac_train_step_device(rollout, agent, (float*)L.pol)) continue;
        }
#ifdef RLC_W1_STAGE
        // every first-layer pass of this update reads pre-update weights: all four first layers go in flight now and
        // land in LDS behind the minibatch gather
        float stg_vt[U::kStage], stg_q[U::kStage], stg_p[U::kStage], stg_v[U::kStage];
        u.H1 = L1C;
        u.stage_load(stg_vt, tt + d.vW1, tt + d.vb1);
        u.stage_load(stg_q, th + d.qW1, th + d.qb1);
        u.stage_load(stg_v, th + d.vW1, th + d.vb1);
        u.H1 = L1A;
        u.stage_load(stg_p, th + d.pW1, th + d.pb1);
#endif
        // ================= sample + gather (utils/replaybuffer.py:32-37) =================
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads) L.idx[b] = host_idx[((size_t)blockIdx.x * n_updates + upd) * B + b];
        }
        lds_barrier();
        const unsigned long long nctr = dv.noise_ctr[agent];
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * AD; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = (float)dv.rep.gr[slot]; L.g[b] = (float)dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * AD; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = (float)ld_gather(&dv.rep.rr[slot]); L.g[b] = (float)ld_gather(&dv.rep.rg[slot]);
            }
            for (int i = 0; i < S; i++) {
                L.x[b * SMAX + i] = ld_gather(&ps[i]);
                L.xc[b * SMAX + i] = rlc_clip_scalar(ld_gather(&ps[i]), dv.clip_state, dv.smin0, dv.smax0);
                L.x2c[b * SMAX + i] = rlc_clip_scalar(ld_gather(&ps2[i]), dv.clip_state, dv.smin0, dv.smax0);
            }
#pragma unroll
            for (int j = 0; j < AD; j++) {
                L.a[b * AD + j] = ld_gather(&pa[j]);
                float e;
                if (eps_in) {
                    e = eps_in[(((size_t)blockIdx.x * n_updates + upd) * B + b) * AD + j];
                } else {
                    const Philox4 p = philox4x32_10(dv.rep.seed[agent] ^ 0x9E3779B97F4A7C15ull, nctr,
                                                    (unsigned long long)(b * AD + j) >> 1);
                    float n0, n1;
                    philox_normal2(p, n0, n1);
                    e = ((b * AD + j) & 1) ? n1 : n0;
                }
                L.eps[b * AD + j] = e;
            }
        }
#ifdef RLC_W1_STAGE
        u.H1 = L1C;
        u.stage_store(stg_vt, L.w1vt);
        u.stage_store(stg_q, L.w1q);
        u.stage_store(stg_v, L.w1v);
        u.H1 = L1A;
        u.stage_store(stg_p, L.w1p);
#endif
        lds_barrier();
        if (tid == 0 && !eps_in) dv.noise_ctr[agent] = nctr + 1;

        STAMP();
        // ================= 1: V'(s') from the target network (sac_network.py:107) =================
        u.H1 = L1C;
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1vt, (const lds_f32*)(L.w1vt + S * L1C), L.x2c);
#else
        u.trunk(tt + d.vW1, tt + d.vb1, L.x2c);
#endif
        lds_barrier();
        u.template fwd_gemm<true>(acc, tt + d.vW2, L2C, L1C);
        u.template bias_relu<0>(acc, tt + d.vb2, L2C);
        u.template row_dot<false, 1>(acc, L2C, [&](int n, int) { return tt[d.vW3 + n]; }, L.part_q);
        lds_barrier();
        for (int b = tid; b < B; b += kThreads) L.vt[b] = u.template part_sum<1>(L.part_q, b, 0) + tt[d.vb3];
        STAMP();
        // ================= 2: the hidden contraction of Q, shared by Q(s,a) and Q(s,pi) =================
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1q, (const lds_f32*)(L.w1q + S * L1C), L.x);
#else
        u.trunk(th + d.qW1, th + d.qb1, L.x);
#endif
        lds_barrier();
        f32x4 accq[MT][NTW];
        u.fwd_gemm(accq, th + d.qW2, L2C, L1C);
        lds_barrier();                 // every wave is done reading hbuf = qh1
        STAMP();
        // ================= 3: pi forward (sac_network.py:234-301) =================
        u.H1 = L1A;
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1p, (const lds_f32*)(L.w1p + S * L1A), L.xc);
#else
        u.trunk(th + d.pW1, th + d.pb1, L.xc);
#endif
        for (int i = tid; i < NS * 256; i += kThreads) {      // [Wm | Ws] transposed: row j < A -> Wm[:, j], j >= A -> Ws[:, j-A]
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < L2A ? (j < AD ? th[d.pWm + n * AD + j] : th[d.pWs + n * AD + (j - AD)]) : 0.0f;
        }
        lds_barrier();
        u.fwd_gemm(acc, th + d.pW2, L2A, L1A);
        u.template bias_relu<0>(acc, th + d.pb2, L2A);
        u.template row_dot<false, NS>(acc, L2A, [&](int n, int j) { return L.wvec[j * 256 + n]; }, L.part_h);
        u.template store_masks<0, true>(acc, L2A);
        lds_barrier();
        for (int b = tid; b < B; b += kThreads) {
            float lp = 0.0f;
#pragma unroll
            for (int j = 0; j < AD; j++) {
                const int k = b * AD + j;
                const float mu = u.template part_sum<NS>(L.part_h, b, j) + th[d.pbm + j];
                const float lsp = u.template part_sum<NS>(L.part_h, b, AD + j) + th[d.pbs + j];
                const float t = tanhf(lsp);
                const float log_std = -20.0f + HALF_RANGE * (t + 1.0f);
                const float sd = expf(log_std);
                const float uu = mu + L.eps[k] * sd;
                const float z = (uu - mu) / (sd + EPS);
                lp += -0.5f * (z * z + 2.0f * log_std + LOG2PI);
                const float pt = tanhf(uu);
                L.t[k] = t; L.sd[k] = sd; L.pit[k] = pt; L.api[k] = pt * amax0;
            }
#pragma unroll
            for (int j = 0; j < AD; j++) {
                const float pt = L.pit[b * AD + j];
                lp -= logf(fminf(fmaxf(1.0f - pt * pt, 0.0f), 1.0f) + 1e-6f);
            }
            L.logp[b] = lp;
            dv.tap_logp[(size_t)agent * RLC_MAX_BATCH + b] = lp;
        }
        lds_barrier();
        STAMP();
        // ================= 4: Q(s,pi), dQ/da, Q(s,a) from the parked contraction =================
        u.template concat_head_dots<AD>(accq, th + d.qb2, L2C, L.api, th + d.qW2, d.arow0, th + d.qW3, L.part_p);
        u.template bias_relu<AD>(accq, th + d.qb2, L2C, L.a, th + d.qW2, d.arow0);
        u.template row_dot<false, 1>(accq, L2C, [&](int n, int) { return th[d.qW3 + n]; }, L.part_q);
        u.template store_masks<1, false>(accq, L2C);
        lds_barrier();
        float part_lp = 0.0f, part_qp = 0.0f, part_lp2 = 0.0f;
        for (int b = tid; b < B; b += kThreads) {
            const float q = u.template part_sum<1>(L.part_q, b, 0) + th[d.qb3];
            const float qpi = u.template part_sum<1 + AD>(L.part_p, b, 0) + th[d.qb3];
            L.q[b] = q; L.qpi[b] = qpi;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            dv.tap_qpi[(size_t)agent * RLC_MAX_BATCH + b] = qpi;
            L.dout[b] = -((L.r[b] + L.g[b] * L.vt[b]) - q) * invB;                 // d q_loss / d q
            part_lp += L.logp[b]; part_qp += qpi; part_lp2 += L.logp[b] * L.logp[b];
        }
        float sum_lp, sum_qp;
        sac_blk_sum2(part_lp, part_qp, L.red, sum_lp, sum_qp);
        const float mean_logp = sum_lp * invB, mean_qpi = sum_qp * invB;
        const float sum_lp2 = sac_blk_sum(part_lp2, L.red);       // for the v_loss tap below
        // pi seeds: d(alpha*mean(logp) - mean(Q(s,pi))) / d(mu_raw, log_std_pre)
        for (int it = tid; it < B * AD; it += kThreads) {
            const int b = it / AD, j = it % AD;
            const float ga = u.template part_sum<1 + AD>(L.part_p, b, 1 + j);
            const float pt = L.pit[it], om = 1.0f - pt * pt;
            const float dlogp_dpit = 2.0f * pt / (fminf(fmaxf(om, 0.0f), 1.0f) + 1e-6f);
            const float dL_dpit = (-1.0f * invB) * ga * amax0 + (alpha_ent * invB) * dlogp_dpit;
            const float dL_du = dL_dpit * om;
            const float sd = L.sd[it], e = L.eps[it];
            const float z = e * sd / (sd + EPS);
            const float dz_dls = e * sd * EPS / ((sd + EPS) * (sd + EPS));
            const float dL_dlogstd = dL_du * e * sd + (alpha_ent * invB) * (-z * dz_dls - 1.0f);
            L.dml[b * NS + j] = dL_du;
            L.dml[b * NS + AD + j] = dL_dlogstd * HALF_RANGE * (1.0f - L.t[it] * L.t[it]);
        }
        lds_barrier();
        // wave-local column reductions of the Q branch from the live (now relu'd) accumulators: d qW3, d qb2
        float g_qw3[NTW], g_qb2[NTW];
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const float w3 = (t < NT && n < L2C) ? th[d.qW3 + n] : 0.0f;
                float s3 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 d4 = *reinterpret_cast<const lds_f32x4*>(&L.dout[16 * mt + 4 * u.g]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float gv = accq[mt][i][r];
                        s3 += gv * d4[r];
                        s2 += gv > 0.0f ? d4[r] * w3 : 0.0f;
                    }
                }
                g_qw3[i] = col4_sum(s3);
                g_qb2[i] = col4_sum(s2);
            }
        }
        STAMP();
        // ================= 5: pi step =================
        float g_ph[NTW][NS], g_pb2[NTW];
        {
            const int NT = (L2A + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const bool ok = t < NT && n < L2A;
                float w3[NS], s3[NS];
#pragma unroll
                for (int j = 0; j < NS; j++) { w3[j] = ok ? L.wvec[j * 256 + n] : 0.0f; s3[j] = 0.0f; }
                float s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int b = 16 * mt + 4 * u.g + r;
                        const float hv = acc[mt][i][r];
                        float dd = 0.0f;
#pragma unroll
                        for (int j = 0; j < NS; j++) {
                            const float sj = L.dml[b * NS + j];
                            s3[j] += hv * sj;
                            dd += sj * w3[j];
                        }
                        s2 += hv > 0.0f ? dd : 0.0f;
                    }
#pragma unroll
                for (int j = 0; j < NS; j++) g_ph[i][j] = col4_sum(s3[j]);
                g_pb2[i] = col4_sum(s2);
            }
        }
        const float alpha_p = adam_alpha(dv.pi_lr[agent], pw[0], pw[1]);
        const float alpha_v = adam_alpha(dv.qv_lr[agent], pw[2], pw[3]);
        // this wave's first two weight-gradient items of pW2 (all it has at widths <= 128): W / m / v / W' in flight NOW,
        // under the backward GEMM, not under one k-loop (RLC_NO_EARLY_PREFETCH: the round-2 order, for A/B runs)
        typename U::WgPre2 pre;
#ifdef RLC_EARLY_PREFETCH
        constexpr int NPRE = (RLC_EARLY_PREFETCH + 0) >= 2 ? 2 : 1;      // -DRLC_EARLY_PREFETCH=<sets in flight>
        u.template wgrad_prefetch<false, NPRE>(pre, L2A, th + d.pW2, mm + d.pW2, vv + d.pW2, tt + d.pW2);
#else
        constexpr int NPRE = 0;
#endif
        u.template bwd_gemm<NS, 0>(acc, th + d.pW2, L2A, L1A, L.dml, L.wvec);
        lds_barrier();
        STAMP();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_p, d.pW1, d.pb1, tapg, tt, tau, L.xc);
        STAMP();
        u.template wgrad_adam_pre<NS, 0, 0, false, false, NPRE>(L.dml, nullptr, L2A, th + d.pW2, mm + d.pW2, vv + d.pW2, alpha_p,
                                        tapg ? tapg + d.pW2 : nullptr, tt + d.pW2, tau, L.wvec, pre);
        {
            const int NT = (L2A + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2A) {
                    // lane group 0 -> pb2[n]; targets 1..A -> Wm[n][j]; A+1..2A -> Ws[n][j]
                    for (int tg = u.g; tg <= NS; tg += 4) {
                        int p = d.pb2 + n;
                        float gr = g_pb2[i];
#pragma unroll
                        for (int j = 0; j < NS; j++)
                            if (tg == j + 1) { p = (j < AD ? d.pWm + n * AD + j : d.pWs + n * AD + (j - AD)); gr = g_ph[i][j]; }
                        U::adam_scalar(th, mm, vv, tt, tapg, p, gr, alpha_p, tau);
                    }
                }
            }
            if (u.wave < NS) {            // bm[j] / bs[j]: sum_b of the seeds, wave j
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dml[b * NS + j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) U::adam_scalar(th, mm, vv, tt, tapg, j < AD ? d.pbm + j : d.pbs + (j - AD), gr, alpha_p, tau);
            }
        }
        lds_barrier();
        STAMP();
        // ================= 6: Q step =================
        u.H1 = L1C;
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1q, (const lds_f32*)(L.w1q + S * L1C), L.x);
#else
        u.trunk(th + d.qW1, th + d.qb1, L.x);
#endif
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < L2C ? th[d.qW3 + n] : 0.0f;
        lds_barrier();
#ifdef RLC_EARLY_PREFETCH
        u.template wgrad_prefetch<false, NPRE>(pre, L2C, th + d.qW2, mm + d.qW2, vv + d.qW2, tt + d.qW2);
#endif
        u.template bwd_gemm<1, 1>(acc, th + d.qW2, L2C, L1C, L.dout, L.wvec);
        lds_barrier();
        STAMP();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_v, d.qW1, d.qb1, tapg, tt, tau, L.x);
        STAMP();
        u.template wgrad_adam_pre<1, AD, 1, false, false, NPRE>(L.dout, L.a, L2C, th + d.qW2, mm + d.qW2, vv + d.qW2, alpha_v,
                                        tapg ? tapg + d.qW2 : nullptr, tt + d.qW2, tau, L.wvec, pre);
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2C && u.g < 2)
                    U::adam_scalar(th, mm, vv, tt, tapg, u.g == 0 ? d.qW3 + n : d.qb2 + n, u.g == 0 ? g_qw3[i] : g_qb2[i],
                                   alpha_v, tau);
            }
            if (u.wave == 0) {            // qb3: sum_b dout[b] by one wave (fixed-order shuffle tree)
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dout[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) U::adam_scalar(th, mm, vv, tt, tapg, d.qb3, gr, alpha_v, tau);
            }
        }
        lds_barrier();
        STAMP();
        // ================= 7: V forward + step =================
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1v, (const lds_f32*)(L.w1v + S * L1C), L.xc);
#else
        u.trunk(th + d.vW1, th + d.vb1, L.xc);
#endif
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < L2C ? th[d.vW3 + n] : 0.0f;
        lds_barrier();
        u.fwd_gemm(acc, th + d.vW2, L2C, L1C);
        u.template bias_relu<0>(acc, th + d.vb2, L2C);
        u.template row_dot<false, 1>(acc, L2C, [&](int n, int) { return L.wvec[n]; }, L.part_q);
        u.template store_masks<0, true>(acc, L2C);
        lds_barrier();
        {   // v, its seed (quirk Q9), taps + losses (the reference fetches pi_loss, q_loss, v_loss: sac_network.py:135-136)
            float ql = 0.0f, vl = 0.0f;
            for (int b = tid; b < B; b += kThreads) {
                const float v = u.template part_sum<1>(L.part_q, b, 0) + th[d.vb3];
                L.v[b] = v;
                dv.tap_v[(size_t)agent * RLC_MAX_BATCH + b] = v;
                L.dvs[b] = -(L.qpi[b] - alpha_ent * mean_logp - v) * invB;
                const float e = (L.r[b] + L.g[b] * L.vt[b]) - L.q[b];
                ql += e * e;
                // sum_j (c - alpha logp[j])^2 over the [B, B] broadcast (Q9), c = q_pi[b] - v[b], in closed form from
                // sum logp and sum logp^2 (the j-loop cost 3 us per update for a diagnostic tap)
                const float cc = L.qpi[b] - v;
                vl += (float)B * cc * cc - 2.0f * cc * alpha_ent * sum_lp + alpha_ent * alpha_ent * sum_lp2;
            }
            sac_blk_sum2(ql, vl, L.red, ql, vl);
            if (tid == 0) {
                dv.tap_loss[agent * 4 + 0] = alpha_ent * mean_logp - mean_qpi;
                dv.tap_loss[agent * 4 + 1] = 0.5f * ql * invB;
                dv.tap_loss[agent * 4 + 2] = 0.5f * vl * invB * invB;
            }
        }
        lds_barrier();
        float g_vw3[NTW], g_vb2[NTW];
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const float w3 = (t < NT && n < L2C) ? L.wvec[n] : 0.0f;
                float s3 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 d4 = *reinterpret_cast<const lds_f32x4*>(&L.dvs[16 * mt + 4 * u.g]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float gv = acc[mt][i][r];
                        s3 += gv * d4[r];
                        s2 += gv > 0.0f ? d4[r] * w3 : 0.0f;
                    }
                }
                g_vw3[i] = col4_sum(s3);
                g_vb2[i] = col4_sum(s2);
            }
        }
#ifdef RLC_EARLY_PREFETCH
        u.template wgrad_prefetch<false, NPRE>(pre, L2C, th + d.vW2, mm + d.vW2, vv + d.vW2, tt + d.vW2);
#endif
        u.template bwd_gemm<1, 0>(acc, th + d.vW2, L2C, L1C, L.dvs, L.wvec);
        lds_barrier();
        STAMP();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_v, d.vW1, d.vb1, tapg, tt, tau, L.xc);
        STAMP();
        u.template wgrad_adam_pre<1, 0, 0, false, false, NPRE>(L.dvs, nullptr, L2C, th + d.vW2, mm + d.vW2, vv + d.vW2, alpha_v,
                                       tapg ? tapg + d.vW2 : nullptr, tt + d.vW2, tau, L.wvec, pre);
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2C && u.g < 2)
                    U::adam_scalar(th, mm, vv, tt, tapg, u.g == 0 ? d.vW3 + n : d.vb2 + n, u.g == 0 ? g_vw3[i] : g_vb2[i],
                                   alpha_v, tau);
            }
            if (u.wave == 0) {
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dvs[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) U::adam_scalar(th, mm, vv, tt, tapg, d.vb3, gr, alpha_v, tau);
            }
        }
        __syncthreads();
        STAMP();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; pw[2] *= 0.9f; pw[3] *= 0.999f; }
        __syncthreads();
    }
#undef STAMP
}

template <int MT, int NTW, int AD, bool T4>
int sac_launch_t(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
                 const float* eps_dev, int grad_taps, hipStream_t st, const RlcSacRollout* rollout) {
    constexpr int MSTRIDE = mask_stride(8 * NTW);
    const size_t lds = ssmem_carve<MSTRIDE>(dv.d, MT, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA SAC kernel needs %zu B of LDS (> 160 KiB)", lds);
    RLC_REQUIRE(!T4 || rlc_tail4(dv.d.B, MT), "tail-of-four kernel launched for batch %d", dv.d.B);
    auto kern = rlc_sac_update_mfma_kernel<MT, NTW, AD, T4>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source, idx_dev, eps_dev,
                       grad_taps, rollout);
    RLC_HIP(hipGetLastError());
    return 0;
}

}  // namespace
