// ddpg_mfma_kernel.h -- fused DDPG update on gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// Same contract as ddpg_generic.hip (one workgroup per agent, n_updates sequential updates per
// launch, every update = sample + gather + agents/DDPG.py:74-95), but the nine [B,200]x[200,200]-class
// contractions of one update run on the MFMA building blocks of mfma_blocks.h and nothing [B,H]-sized ever
// leaves the CU (see that header for the LDS / VGPR / HBM roles and the tiling).
//
// Supported shapes: S <= 8, A in {1,2}, H1/HA/HC multiples of 4 in [16,256], B <= 128.
#pragma once
#include "mfma_blocks.h"
#include "ddpg_rollout_device.h"

namespace {

using namespace mfb;

constexpr int NTW = 2;        // N tiles per wave  (N <= 256)
constexpr int NT16 = 16;      // N tiles per row at most (N <= 256)
constexpr int MSTRIDE = mask_stride(NT16);      // 272 B
static_assert(MSTRIDE == 272, "mask stride of the 256-wide kernels");

struct Smem {
    lds_f32* hbuf;
    lds_u8* mask;       // relu masks, one BYTE (0/1) per (row, unit): [MB][MSTRIDE]
    lds_f32* part;      // [kWaves][MB][AD]
    lds_f32* wvec;      // [AD][256] staged Wc3 (row 0) or Wa3 transposed
    lds_f32 *x, *x2, *a, *aout, *mu, *dz, *q, *y, *dq;
    lds_f64 *r, *g;
    lds_i64* idx;
    lds_i32* pool;
    lds_i32* dups;
    lds_f32x4* xbuf;    // hand-off of the split 13th tile (mfma_blocks.h)
    lds_f32 *w1t, *w1o; // first layer staged in LDS ([S][H1] weights, [H1] biases): target / online (mfma_blocks.h stage_*)
};

// carve the dynamic LDS; base may be null (host: only the size is wanted)
__host__ __device__ inline size_t smem_carve(const RlcDims& d, int MT, lds_u8* base, Smem* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        lds_u8* p = base + off;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int MB = MT * 16, A = d.A, LDH = ldh_for(d.H1);
    Smem L;
    // + 16 floats of tail: the unmasked fragment reads of the last k-chunk run up to 15 floats past a row's
    // end (into the next row, or into this zeroed tail after the last row)
    L.hbuf = (lds_f32*)take(sizeof(float) * (MB * LDH + 16));
    L.r = (lds_f64*)take(sizeof(double) * MB);
    L.g = (lds_f64*)take(sizeof(double) * MB);
    L.idx = (lds_i64*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.mask = take((size_t)MB * MSTRIDE);
    L.part = (lds_f32*)take(sizeof(float) * kWaves * MB * A);
    L.wvec = (lds_f32*)take(sizeof(float) * A * 256);
    L.x = (lds_f32*)take(sizeof(float) * MB * SMAX);      // rows padded to 8 floats: two ds_read_b128
    L.x2 = (lds_f32*)take(sizeof(float) * MB * SMAX);
    L.a = (lds_f32*)take(sizeof(float) * MB * A);
    L.aout = (lds_f32*)take(sizeof(float) * MB * A);
    L.mu = (lds_f32*)take(sizeof(float) * MB * A);
    L.dz = (lds_f32*)take(sizeof(float) * MB * A);
    L.q = (lds_f32*)take(sizeof(float) * MB);
    L.y = (lds_f32*)take(sizeof(float) * MB);
    L.dq = (lds_f32*)take(sizeof(float) * MB);
    L.pool = (lds_i32*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (lds_i32*)take(sizeof(int) * 4);
    L.xbuf = (lds_f32x4*)take(sizeof(float) * 4 * 64 * (MT - (MT + 3) / 4));
#ifdef RLC_W1_STAGE
    L.w1t = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.H1);
    L.w1o = (lds_f32*)take(sizeof(float) * (d.S + 1) * d.H1);
#else
    L.w1t = L.w1o = nullptr;
#endif
    if (out) *out = L;
    return off;
}

// FUSE: the actor's and the critic's second layers have one width: their hidden contractions over one trunk image run
// as ONE k-loop (target pair, online pair)
// T4: the minibatch ends within the first four rows of its last tile (mfma_blocks.h, Blk's T4; the launcher checks it)
template <int MT, int AD, bool FUSE, bool T4>
__global__ __launch_bounds__(kThreads) void rlc_ddpg_update_mfma_kernel(RlcDev dv, int first_agent, int n_updates,
                                                                        int source, const long long* host_idx,
                                                                        int grad_taps, const RlcRollout* rollout,
                                                                        int q8_first) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using U = Blk<MT, NTW, MSTRIDE, false, false, T4>;
    constexpr int MB = U::MB;
    const RlcDims d = dv.d;
    U u;
    u.init_geometry();
    u.S = d.S; u.H1 = d.H1; u.B = d.B; u.LDH = ldh_for(d.H1);
    Smem L;
    smem_carve(d, MT, (lds_u8*)smem, &L);
    u.L.hbuf = L.hbuf; u.L.mask = L.mask; u.L.xbuf = L.xbuf;
    const int tid = u.tid, S = d.S, H1 = d.H1, HA = d.HA, HC = d.HC, B = d.B;
    const int agent = first_agent + blockIdx.x;
    constexpr bool fuse_fwd = FUSE;      // the launcher picks FUSE when HA == HC and the networks share their first layer
    // `network: separate` (actor_network.py:73-96 / critic_network.py:77-99): the critic has a first layer of its own
    // (d.oWc1 / d.obc1; in the hydra network they alias d.oW1 / d.ob1), stepped and Polyak-averaged by the critic's Adam
    // alone.  The activation image then has to change hands: six first-layer passes per update instead of three.
    const bool sep = d.sep != 0;         // workgroup-uniform

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* m_a = dv.m_a + (size_t)agent * d.Ppad;
    float* v_a = dv.v_a + (size_t)agent * d.Ppad;
    float* m_c = dv.m_c + (size_t)agent * d.Ppad;
    float* v_c = dv.v_c + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 4;
    const float lr_a = dv.actor_lr[agent], lr_c = dv.critic_lr[agent], tau = dv.tau;
#ifdef RLC_STAMPS
    float* stamp_buf = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;   // diagnostic build: no gradient taps
    float* tap_gc = nullptr;
    float* tap_ga = nullptr;
#else
    float* tap_gc = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;
    float* tap_ga = grad_taps ? dv.tap_ga + (size_t)agent * d.Ppad : nullptr;
#endif
    float amax[AD];
#pragma unroll
    for (int j = 0; j < AD; j++) amax[j] = dv.amax[j];

    // zero the padded tails of the per-sample vectors once (rows >= B never change afterwards)
    for (int i = tid; i < MB * AD; i += kThreads) { L.a[i] = 0.f; L.aout[i] = 0.f; L.mu[i] = 0.f; L.dz[i] = 0.f; }
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.x2[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) { L.q[i] = 0.f; L.y[i] = 0.f; L.dq[i] = 0.f; }
    for (int i = tid; i < MB * MSTRIDE / 4; i += kThreads) reinterpret_cast<lds_u32*>(L.mask)[i] = 0u;
    if (tid < 16) L.hbuf[MB * u.LDH + tid] = 0.0f;
    __syncthreads();

    stagger_start();
#ifdef RLC_PRIO_YOUNG
    if (u.wave >= 4) __builtin_amdgcn_s_setprio(1);     // the second-dispatched half loses every issue arbitration otherwise
#endif
    f32x4 acc[MT][NTW];
#ifdef RLC_STAMPS
    // diagnostic build only: phase boundaries in shader cycles, written where the critic gradient tap lives
    long long t_prev = clock64();
    int stamp_i = 0;
#define STAMP()                                                                                  \
    do {                                                                                         \
        if (tid == 0 && stamp_buf) { const long long t = clock64(); stamp_buf[stamp_i] += (float)(t - t_prev); t_prev = t; } \
        stamp_i++;                                                                               \
    } while (0)
    if (stamp_buf) for (int i = tid; i < 64; i += kThreads) stamp_buf[i] = 0.0f;
    const long long t_k0 = clock64(), w_k0 = wall_clock64();
    u.stamp_buf = stamp_buf;
    __syncthreads();
#else
#define STAMP() do {} while (0)
#endif

    for (int upd = 0; upd < n_updates; upd++) {
#ifdef RLC_STAMPS
        stamp_i = 0;
#endif
        // Re-materialise lane geometry every update: without this hipcc hoists the address arithmetic of
        // all ~15 phases out of the update loop and then spills it (190 scratch stores in the prologue).
        asm volatile("" : "+v"(u.c), "+v"(u.g), "+s"(u.wave));
        if (rollout) {
            // on-device experiment loop: one environment step first; the update runs when learn() would
            // (agents/base_agent.py:65-70).  hbuf is free here (the trunk overwrites it below).
            if (!rlc_train_step_device(rollout, agent, (float*)L.hbuf, upd == 0 ? q8_first : 0)) continue;
        }
#ifdef RLC_W1_STAGE
        // both first layers go in flight now and land in LDS behind the minibatch gather
        float stg_t[U::kStage], stg_o[U::kStage];
        u.stage_load(stg_t, tt + d.oW1, tt + d.ob1);
        u.stage_load(stg_o, th + d.oW1, th + d.ob1);
#endif
        // ================= sample + gather (utils/replaybuffer.py:32-37) =================
        if (!(ablate(4) && upd > 0)) {
        u.sub_begin();
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            u.sub_stamp(25);
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
            u.sub_stamp(26);
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads) L.idx[b] = host_idx[((size_t)blockIdx.x * n_updates + upd) * B + b];
        }
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * AD; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = dv.rep.gr[slot]; L.g[b] = dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * AD; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = ld_gather(&dv.rep.rr[slot]); L.g[b] = ld_gather(&dv.rep.rg[slot]);
            }
            for (int i = 0; i < S; i++) {
                L.x[b * SMAX + i] = clip_state_val(ld_gather(&ps[i]), dv.clip_state, dv.smin[i], dv.smax[i]);
                L.x2[b * SMAX + i] = clip_state_val(ld_gather(&ps2[i]), dv.clip_state, dv.smin[i], dv.smax[i]);
            }
#pragma unroll
            for (int j = 0; j < AD; j++) L.a[b * AD + j] = ld_gather(&pa[j]);
        }
        u.sub_stamp(27);
        }
#ifdef RLC_W1_STAGE
        u.stage_store(stg_t, L.w1t);
        u.stage_store(stg_o, L.w1o);
#endif
        lds_barrier();
        STAMP();

        // ================= steps 1-2: target networks on s' (DDPG.py:77) =================
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1t, (const lds_f32*)(L.w1t + S * H1), L.x2);
#else
        u.trunk(tt + d.oW1, tt + d.ob1, L.x2);
#endif
        lds_barrier();
        STAMP();
        // target actor and target critic read the same trunk image and the critic needs the actor's output only in its
        // epilogue (the action rows): one k-loop for both hidden contractions when the two layers have one width
        // (mfma_blocks.h fwd_gemm2; +0.9 % at the BASELINE shape, profiles/r03_variant_timings_s9.txt)
        f32x4 acc2[MT][NTW];
        if constexpr (fuse_fwd) u.template fwd_gemm2<true>(acc, acc2, tt + d.oWa2, tt + d.oWc2, HA, H1);
        else u.template fwd_gemm<true>(acc, tt + d.oWa2, HA, H1);
        u.template bias_relu<0>(acc, tt + d.oba2, HA);
        u.template row_dot<false, AD>(acc, HA, [&](int n, int j) { return tt[d.oWa3 + n * AD + j]; }, L.part);   // z' partials
        lds_barrier();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            L.aout[i] = tanhf(u.template part_sum<AD>(L.part, b, j) + tt[d.oba3 + j]) * amax[j];
        }
        lds_barrier();
        STAMP();
        if constexpr (fuse_fwd) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < NTW; i++) acc[mt][i] = acc2[mt][i];
        } else {
            if (sep) {
                u.trunk(tt + d.oWc1, tt + d.obc1, L.x2);      // the target critic's own first layer
                lds_barrier();
            }
            u.template fwd_gemm<true>(acc, tt + d.oWc2, HC, H1);
        }
        u.template bias_relu<AD>(acc, tt + d.obc2, HC, L.aout, tt + d.oWc2, d.arow0);
        u.template row_dot<false, 1>(acc, HC, [&](int n, int) { return tt[d.oWc3 + n]; }, L.part);      // q' partials
        lds_barrier();
        STAMP();
        for (int b = tid; b < B; b += kThreads) {
            const float qt = u.template part_sum<1>(L.part, b, 0) + tt[d.obc3];
            const float y = (float)(L.r[b] + L.g[b] * (double)qt);     // float64 TD glue (DDPG.py:80-84)
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + b] = y;
        }
        lds_barrier();
        STAMP();

        // ================= step 3: critic step =================
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1o, (const lds_f32*)(L.w1o + S * H1), L.x);
#else
        u.trunk(th + d.oWc1, th + d.obc1, L.x);           // the critic's first layer (= the shared trunk in the hydra network)
#endif
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < HC ? th[d.oWc3 + n] : 0.0f;
        lds_barrier();
        STAMP();
        u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        u.template bias_relu<AD>(acc, th + d.obc2, HC, L.a, th + d.oWc2, d.arow0);
        u.template row_dot<false, 1>(acc, HC, [&](int n, int) { return th[d.oWc3 + n]; }, L.part);      // q partials
        lds_barrier();
        STAMP();
        for (int b = tid; b < B; b += kThreads) {
            const float q = u.template part_sum<1>(L.part, b, 0) + th[d.obc3];
            L.q[b] = q;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            L.dq[b] = 2.0f * (q - L.y[b]) / (float)B;                  // d mean((y-q)^2)/dq
        }
        lds_barrier();
        STAMP();
        // wave-local column reductions from the live g2 accumulators: dWc3, dbc2; then the relu masks
        float g_wc3[NTW], g_bc2[NTW];
        {
            const int NT = (HC + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const float w3 = (t < NT && n < HC) ? L.wvec[n] : 0.0f;
                float s3 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 dq4 = *reinterpret_cast<const lds_f32x4*>(&L.dq[16 * mt + 4 * u.g]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float gv = acc[mt][i][r];
                        s3 += gv * dq4[r];
                        s2 += gv > 0.0f ? dq4[r] * w3 : 0.0f;
                    }
                }
                g_wc3[i] = col4_sum(s3);
                g_bc2[i] = col4_sum(s2);
            }
        }
        u.template store_masks<-2, true>(acc, HC);
        lds_barrier();
        STAMP();
        // dh1 = (dg2 . Wc2[:H1]^T) * relu'(h1) -> W1/b1 gradients -> critic Adam on the trunk (Q1)
        const float alpha_c = adam_alpha(lr_c, pw[2], pw[3]);
        u.template bwd_gemm<1, -2>(acc, th + d.oWc2, HC, H1, L.dq, L.wvec);
        lds_barrier();      // every wave has finished reading the pre-step Wc2 rows and W1
        STAMP();
        // the first weight-gradient item's W / m / v / W' go in flight before the first-layer gradient, not after it
        typename U::WgPre2 pre;
#ifdef RLC_EARLY_PREFETCH
        constexpr int NPRE = 1;
        u.template wgrad_prefetch<false, 1>(pre, HC, th + d.oWc2, m_c + d.oWc2, v_c + d.oWc2, tt + d.oWc2);
#else
        constexpr int NPRE = 0;
#endif
#ifdef RLC_W1_STAGE
        u.trunk_grad_adam(acc, th, m_c, v_c, alpha_c, d.oW1, d.ob1, tap_gc, nullptr, 0.0f, L.x, NoExtra{}, L.w1o);   // step 4 reads the stepped trunk
#else
        // hydra: the trunk's target copy follows in the actor step; separate networks: the critic's first layer is
        // Polyak-averaged here, by the only optimizer that owns it
        u.trunk_grad_adam(acc, th, m_c, v_c, alpha_c, d.oWc1, d.obc1, tap_gc, sep ? tt : nullptr, tau, L.x);
#endif
        STAMP();
        // dWc2 = [h1|a]^T . dg2 with Adam + Polyak in the epilogue
        u.template wgrad_adam_pre<1, AD, -1, false, false, NPRE>(L.dq, L.a, HC, th + d.oWc2, m_c + d.oWc2, v_c + d.oWc2, alpha_c,
                     tap_gc ? tap_gc + d.oWc2 : nullptr, tt + d.oWc2, tau, L.wvec, pre);
        // small critic tensors: Wc3, bc2 (column owners), bc3 (one thread)
        u.sub_begin();
        {
            const int NT = (HC + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < HC && u.g < 2) {
                    const int p = (u.g == 0) ? d.oWc3 + n : d.obc2 + n;
                    const float gr = (u.g == 0) ? g_wc3[i] : g_bc2[i];
                    float mm = m_c[p], vv = v_c[p];
                    const float o = tt[p];              // (with the other loads: one memory round trip, not two)
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_c);
                    m_c[p] = mm; v_c[p] = vv; th[p] = nv;
                    if (tap_gc) tap_gc[p] = gr;
                    tt[p] = o + tau * (nv - o);
                }
            }
            if (u.wave == 0) {            // bc3: sum_b dq[b] by one wave (fixed-order shuffle tree)
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dq[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) {
                    const int p = d.obc3;
                    float mm = m_c[p], vv = v_c[p];
                    const float o = tt[p];              // (with the other loads: one memory round trip, not two)
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_c);
                    m_c[p] = mm; v_c[p] = vv; th[p] = nv;
                    if (tap_gc) tap_gc[p] = gr;
                    tt[p] = o + tau * (nv - o);
                }
            }
        }
        u.sub_stamp(29);              // (diagnostic build) the small tensors; 30: wave 0 waiting for the slowest wave
        __syncthreads();
        u.sub_stamp(30);
        STAMP();
        if (tid == 0) { pw[2] *= 0.9f; pw[3] *= 0.999f; }

        // ================= step 4: actor forward with the updated trunk (DDPG.py:90) =================
#ifdef RLC_W1_STAGE
        u.trunk((const lds_f32*)L.w1o, (const lds_f32*)(L.w1o + S * H1), L.x);
#else
        u.trunk(th + d.oW1, th + d.ob1, L.x);
#endif
        for (int i = tid; i < AD * 256; i += kThreads) {
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < HA ? th[d.oWa3 + n * AD + j] : 0.0f;       // Wa3 transposed [j][n]
        }
        __syncthreads();
        STAMP();
        f32x4 acc7[MT][NTW];          // the critic's hidden contraction at the new trunk (its action rows enter in the epilogue)
        if constexpr (fuse_fwd) u.fwd_gemm2(acc, acc7, th + d.oWa2, th + d.oWc2, HA, H1);
        else u.fwd_gemm(acc, th + d.oWa2, HA, H1);
        u.template bias_relu<0>(acc, th + d.oba2, HA);
        u.template row_dot<false, AD>(acc, HA, [&](int n, int j) { return th[d.oWa3 + n * AD + j]; }, L.part);   // z partials
        u.template store_masks<-2, true>(acc, HA);
        lds_barrier();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float mu = tanhf(u.template part_sum<AD>(L.part, b, j) + th[d.oba3 + j]);
            L.mu[i] = mu;
            const float ao = mu * amax[j];
            L.aout[i] = ao;
            dv.tap_aout[(size_t)agent * RLC_MAX_BATCH * AD + i] = ao;
        }
        lds_barrier();
        STAMP();
        // the h2 accumulators are needed again for dWa3 once dz is known: park them in registers
        f32x4 h2acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) h2acc[mt][i] = acc[mt][i];

        // ================= step 5: dQ/da at the scaled action, updated critic (DDPG.py:91) =================
        if constexpr (fuse_fwd) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int i = 0; i < NTW; i++) acc[mt][i] = acc7[mt][i];
        } else {
            if (sep) {
                u.trunk(th + d.oWc1, th + d.obc1, L.x);       // the stepped critic's first layer at s
                lds_barrier();
            }
            u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        }
        u.template bias_relu<AD>(acc, th + d.obc2, HC, L.aout, th + d.oWc2, d.arow0);
        // dqda[b][j] = sum_n step(g2[b][n]) * Wc3[n] * Wc2[H1+j][n]
        u.template row_dot<true, AD>(acc, HC, [&](int n, int j) { return th[d.oWc2 + rlc_blk_index(d.arow0 + j, n, HC)] * th[d.oWc3 + n]; },
                                     L.part);
        lds_barrier();
        STAMP();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float dqda = u.template part_sum<AD>(L.part, b, j);
            dv.tap_dqda[(size_t)agent * RLC_MAX_BATCH * AD + i] = dqda;
            const float mu = L.mu[i];
            L.dz[i] = -dqda * (1.0f - mu * mu);                         // grad_ys = -dQ/da on tanh output (Q3)
        }
        lds_barrier();
        STAMP();

        // ================= step 6: actor step =================
        float g_wa3[NTW][AD], g_ba2[NTW];
        {
            const int NT = (HA + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const bool ok = t < NT && n < HA;
                float w3[AD], s3[AD];
#pragma unroll
                for (int j = 0; j < AD; j++) { w3[j] = ok ? L.wvec[j * 256 + n] : 0.0f; s3[j] = 0.0f; }
                float s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int b = 16 * mt + 4 * u.g + r;
                        const float hv = h2acc[mt][i][r];
                        float dd = 0.0f;
#pragma unroll
                        for (int j = 0; j < AD; j++) {
                            const float dzb = L.dz[b * AD + j];
                            s3[j] += hv * dzb;
                            dd += dzb * w3[j];
                        }
                        s2 += hv > 0.0f ? dd : 0.0f;
                    }
#pragma unroll
                for (int j = 0; j < AD; j++) g_wa3[i][j] = col4_sum(s3[j]);
                g_ba2[i] = col4_sum(s2);
            }
        }
        const float alpha_a = adam_alpha(lr_a, pw[0], pw[1]);
        u.template bwd_gemm<AD, -2>(acc, th + d.oWa2, HA, H1, L.dz, L.wvec);
        lds_barrier();
        if (sep) {
            u.trunk(th + d.oW1, th + d.ob1, L.x);             // the actor's image again: its relu mask and the weight-gradient operand
            lds_barrier();
        }
        STAMP();
#ifdef RLC_EARLY_PREFETCH
        u.template wgrad_prefetch<false, 1>(pre, HA, th + d.oWa2, m_a + d.oWa2, v_a + d.oWa2, tt + d.oWa2);
#endif
        u.trunk_grad_adam(acc, th, m_a, v_a, alpha_a, d.oW1, d.ob1, tap_ga, tt, tau, L.x);
        STAMP();
        u.template wgrad_adam_pre<AD, 0, -1, false, false, NPRE>(L.dz, nullptr, HA, th + d.oWa2, m_a + d.oWa2, v_a + d.oWa2, alpha_a,
                     tap_ga ? tap_ga + d.oWa2 : nullptr, tt + d.oWa2, tau, L.wvec, pre);
        {
            const int NT = (HA + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < HA && u.g <= AD) {
                    // lane group 0 -> ba2[n]; groups 1..AD -> Wa3[n][j]
                    int p = d.oba2 + n;
                    float gr = g_ba2[i];
#pragma unroll
                    for (int j = 0; j < AD; j++)
                        if (u.g == j + 1) { p = d.oWa3 + n * AD + j; gr = g_wa3[i][j]; }
                    float mm = m_a[p], vv = v_a[p];
                    const float o = tt[p];              // (with the other loads: one memory round trip, not two)
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_a);
                    m_a[p] = mm; v_a[p] = vv; th[p] = nv;
                    if (tap_ga) tap_ga[p] = gr;
                    tt[p] = o + tau * (nv - o);
                }
            }
            if (u.wave < AD) {            // ba3[j]: sum_b dz[b][j], wave j
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dz[b * AD + j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) {
                    const int p = d.oba3 + j;
                    float mm = m_a[p], vv = v_a[p];
                    const float o = tt[p];              // (with the other loads: one memory round trip, not two)
                    const float nv = adam_step(th[p], gr, mm, vv, alpha_a);
                    m_a[p] = mm; v_a[p] = vv; th[p] = nv;
                    if (tap_ga) tap_ga[p] = gr;
                    tt[p] = o + tau * (nv - o);
                }
            }
        }
        __syncthreads();
        STAMP();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; }
        __syncthreads();
    }
#ifdef RLC_STAMPS
    if (tid == 0 && stamp_buf) {
        stamp_buf[40] = (float)(clock64() - t_k0);
        stamp_buf[41] = (float)(wall_clock64() - w_k0);
    }
#endif
}

template <int MT, int AD, bool FUSE, bool T4>
int launch_tf(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
              int grad_taps, hipStream_t st, const RlcRollout* rollout, int q8_first) {
    const size_t lds = smem_carve(dv.d, MT, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA DDPG kernel needs %zu B of LDS (> 160 KiB)", lds);
    RLC_REQUIRE(!T4 || rlc_tail4(dv.d.B, MT), "tail-of-four kernel launched for batch %d", dv.d.B);
    auto kern = rlc_ddpg_update_mfma_kernel<MT, AD, FUSE, T4>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source, idx_dev,
                       grad_taps, rollout, q8_first);
    RLC_HIP(hipGetLastError());
    return 0;
}

template <int MT, int AD, bool T4>
int launch_t(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
             int grad_taps, hipStream_t st, const RlcRollout* rollout, int q8_first) {
#ifndef RLC_DDPG_SEPARATE_FWD
    if (dv.d.HA == dv.d.HC && !dv.d.sep)
        return launch_tf<MT, AD, true, T4>(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, rollout, q8_first);
#endif
    return launch_tf<MT, AD, false, T4>(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, rollout, q8_first);
}

}  // namespace
