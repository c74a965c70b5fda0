// naf_mfma_kernel.h -- fused NAF update on gfx950 fp32 matrix cores.
//
// Same contract as naf_generic.hip (one workgroup per agent, n_updates sequential updates per launch, every update =
// sample_batch + NAF_Network_Manager.update_network, agents/NAF.py:69-75), built from the MFMA blocks of
// mfma_blocks.h.  Network (agents/network/naf_network.py:79-123): trunk S -> L1; mu branch L1 -> L2 -> A (tanh * a_max);
// V branch L1 -> L2 -> 1; the columns of the lower-triangular L hang off the TRUNK (A diagonal heads through
// exp(clip(., -5, 5)), A(A-1)/2 below-diagonal heads).  Per update:
//
//   1  V'(s')       trunk(target, s') -> GEMM Wv2' -> V' -> y = r + gamma V' in float64 (agents/NAF.py:70)
//   2  forward      trunk(s); L heads from the trunk image (VALU, 16 x 4 lane tiles); GEMM Wa2 -> mu; GEMM Wv2 -> V
//   3  per sample   L columns, A(s,a) = -1/2 |L^T (a - mu)|^2, Q = V + A, loss = SUM (y - Q)^2 (naf_network.py:53)
//   4  backward     GEMM (V branch, rank one) + GEMM (mu branch, accumulated) + head terms -> d trunk -> W1/b1
//   5  weights      GEMM (d Wa2), GEMM (d Wv2) with Adam + Polyak in their epilogues; heads and output layers on VALU
//
// 7 contractions of [B, L1] x [L1, L2] per update.  One Adam over every tensor (naf_network.py:54), Polyak by
// assign_add (:62-63).  Supported shapes: S <= 8, A in {1,2}, L1/L2 multiples of 4 in [16, 128*NTW], B <= 128.
#pragma once
#include "mfma_blocks.h"
#include "naf_common.h"
#include "naf_rollout_device.h"

namespace {

using namespace mfb;

constexpr int NHP = 4;       // L-head outputs per sample, padded (A + A(A-1)/2 <= 3 for A <= 2)

struct NSmem {
    lds_f32* hbuf;
    lds_u8* mask;        // bit 0: mu-branch hidden, bit 1: V-branch hidden
    lds_f32* part_a;     // [kWaves][MB][A]  pre-tanh action partials
    lds_f32* part_v;     // [kWaves][MB]     V partials (also V'(s'))
    lds_f32* wvec;       // [A+1][256]: rows < A = Wa3 transposed, row A = Wv3
    lds_f32* wh;         // [NHP][256] the head weight vectors (head j: j < A diagonal c = j, then the below-diagonal ones)
    lds_f32 *x, *x2;     // [MB][SMAX]
    lds_f32 *a, *dz;     // [MB][A]
    lds_f32 *hd, *dhd;   // [MB][NHP] head pre-activations / their gradients
    lds_f32 *y, *dV;     // [MB]
    lds_f64 *r, *g;
    lds_i64* idx;
    lds_i32* pool;
    lds_i32* dups;
    lds_f32* pol;        // scratch of the on-device training step (naf_rollout_device.h)
    lds_f32x4* xbuf;     // hand-off of the split 13th tile (mfma_blocks.h)
};

template <int MSTRIDE>
__host__ __device__ inline size_t nsmem_carve(const RlcNafDims& d, int MT, lds_u8* base, NSmem* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        lds_u8* p = base + off;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int MB = MT * 16, A = d.A, LDH = ldh_for(d.L1);
    NSmem L;
    L.hbuf = (lds_f32*)take(sizeof(float) * (MB * LDH + 16));
    L.r = (lds_f64*)take(sizeof(double) * MB);
    L.g = (lds_f64*)take(sizeof(double) * MB);
    L.idx = (lds_i64*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.mask = take((size_t)MB * MSTRIDE);
    L.part_a = (lds_f32*)take(sizeof(float) * kWaves * MB * A);
    L.part_v = (lds_f32*)take(sizeof(float) * kWaves * MB);
    L.wvec = (lds_f32*)take(sizeof(float) * (A + 1) * 256);
    L.wh = (lds_f32*)take(sizeof(float) * NHP * 256);
    L.x = (lds_f32*)take(sizeof(float) * MB * SMAX);
    L.x2 = (lds_f32*)take(sizeof(float) * MB * SMAX);
    L.a = (lds_f32*)take(sizeof(float) * MB * A);
    L.dz = (lds_f32*)take(sizeof(float) * MB * A);
    L.hd = (lds_f32*)take(sizeof(float) * MB * NHP);
    L.dhd = (lds_f32*)take(sizeof(float) * MB * NHP);
    L.y = (lds_f32*)take(sizeof(float) * MB);
    L.dV = (lds_f32*)take(sizeof(float) * MB);
    L.pool = (lds_i32*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (lds_i32*)take(sizeof(int) * 4);
    L.pol = (lds_f32*)take(sizeof(float) * (naf_policy_lds_floats(d) + 4));
    L.xbuf = (lds_f32x4*)take(sizeof(float) * 4 * 64 * (MT - (MT + 3) / 4));
    if (out) *out = L;
    return off;
}

// device offset of head j's weight vector / bias (j < A: diagonal of column j; then column c's below-diagonal entries)
__device__ __forceinline__ void naf_head_ref(const RlcNafDims& d, int j, int& ow, int& stride, int& ob) {
    if (j < d.A) { ow = d.Wd[j]; stride = 1; ob = d.bd[j]; return; }
    int jj = j - d.A, c = 0;
    while (jj >= d.A - 1 - c) { jj -= d.A - 1 - c; c++; }
    ow = d.Wn[c] + jj; stride = d.A - 1 - c; ob = d.bn[c] + jj;
}

// T4: the minibatch ends within the first four rows of its last tile (mfma_blocks.h, Blk's T4; the launcher checks it)
template <int MT, int NTW, int AD, bool T4>
__global__ __launch_bounds__(kThreads) void rlc_naf_update_mfma_kernel(RlcNafDev dv, int first_agent, int n_updates,
                                                                       int source, const long long* host_idx,
                                                                       int grad_taps, const RlcNafRollout* rollout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int MSTRIDE = mask_stride(8 * NTW);
    constexpr int NN = AD * (AD - 1) / 2, NH = AD + NN;
    static_assert(NH <= NHP, "head count");
    using U = Blk<MT, NTW, MSTRIDE, false, false, T4>;
    constexpr int MB = U::MB;
    const RlcNafDims d = dv.d;
    U u;
    u.init_geometry();
    u.S = d.S; u.H1 = d.L1; u.B = d.B; u.LDH = ldh_for(d.L1);
    NSmem L;
    nsmem_carve<MSTRIDE>(d, MT, (lds_u8*)smem, &L);
    u.L.hbuf = L.hbuf; u.L.mask = L.mask; u.L.xbuf = L.xbuf;
    const int tid = u.tid, S = d.S, L1 = d.L1, L2 = d.L2, B = d.B, LDH = u.LDH;
    const int agent = first_agent + blockIdx.x;

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* mm = dv.m + (size_t)agent * d.Ppad;
    float* vv = dv.v + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 2;
#ifdef RLC_STAMPS
    float* stamp_buf = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;   // diagnostic build: no gradient taps
    float* tapg = nullptr;
#else
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;
#endif
    const float tau = dv.tau;
    float amax[AD];
#pragma unroll
    for (int j = 0; j < AD; j++) amax[j] = dv.amax[j];
    lds_f32* wv3 = L.wvec + AD * 256;

    // zero the padded tails of the per-sample vectors once (rows >= B never change afterwards)
    for (int i = tid; i < MB * AD; i += kThreads) { L.a[i] = 0.f; L.dz[i] = 0.f; }
    for (int i = tid; i < MB * NHP; i += kThreads) { L.hd[i] = 0.f; L.dhd[i] = 0.f; }
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.x2[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) { L.y[i] = 0.f; L.dV[i] = 0.f; }
    for (int i = tid; i < NHP * 256; i += kThreads) L.wh[i] = 0.f;
    for (int i = tid; i < MB * MSTRIDE / 4; i += kThreads) reinterpret_cast<lds_u32*>(L.mask)[i] = 0u;
    if (tid < 16) L.hbuf[MB * LDH + tid] = 0.0f;
    __syncthreads();

    stagger_start();
    f32x4 acc[MT][NTW];
#ifdef RLC_STAMPS
    // diagnostic build only: phase boundaries in shader cycles (scripts/phase_stamps_naf.py), written where the gradient tap lives
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
            if (!rlc_naf_train_step_device(rollout, agent, (float*)L.pol)) continue;
        }
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
        lds_barrier();

        STAMP();
        // ================= 1: target V'(s') and the float64 TD glue (agents/NAF.py:70) =================
        u.trunk(tt + d.W1, tt + d.b1, L.x2);
        lds_barrier();
        u.template fwd_gemm<true>(acc, tt + d.Wv2, L2, L1);
        u.template bias_relu<0>(acc, tt + d.bv2, L2);
        u.template row_dot<false, 1>(acc, L2, [&](int n, int) { return tt[d.Wv3 + n]; }, L.part_v);
        lds_barrier();
        for (int b = tid; b < B; b += kThreads) {
            const float vt = u.template part_sum<1>(L.part_v, b, 0) + tt[d.bv3];
            const float y = (float)(L.r[b] + L.g[b] * (double)vt);
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + b] = y;
        }
        STAMP();
        // ================= 2: online forward =================
        u.trunk(th + d.W1, th + d.b1, L.x);
        for (int i = tid; i < (AD + 1) * 256; i += kThreads) {      // rows < A: Wa3 transposed; row A: Wv3
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < L2 ? (j < AD ? th[d.Wa3 + n * AD + j] : th[d.Wv3 + n]) : 0.0f;
        }
        for (int i = tid; i < NH * 256; i += kThreads) {
            const int j = i / 256, k = i % 256;
            int ow, st, ob;
            naf_head_ref(d, j, ow, st, ob);
            L.wh[i] = k < L1 ? th[ow + k * st] : 0.0f;
        }
        lds_barrier();
        // L heads from the trunk image: wave w < MT takes batch tile w; lane (c, g) sums k = 16 ch + 4 g .. + 3 of row
        // 16 w + c; the four lane groups are combined in a fixed order
        if (u.wave < MT) {
            const lds_f32* hp = L.hbuf + (16 * u.wave + u.c) * LDH + 4 * u.g;
            float hs[NH];
#pragma unroll
            for (int j = 0; j < NH; j++) hs[j] = 0.0f;
            const int KB = (L1 + 15) >> 4;
            for (int ch = 0; ch < KB; ch++) {
                const f32x4 hv = *reinterpret_cast<const lds_f32x4*>(hp + 16 * ch);
#pragma unroll
                for (int j = 0; j < NH; j++) {
                    const f32x4 wj = *reinterpret_cast<const lds_f32x4*>(L.wh + j * 256 + 16 * ch + 4 * u.g);
#pragma unroll
                    for (int e = 0; e < 4; e++) hs[j] += hv[e] * wj[e];
                }
            }
#pragma unroll
            for (int j = 0; j < NH; j++) {
                const float s = col4_sum(hs[j]);
                if (u.g == 0) {
                    int ow, st, ob;
                    naf_head_ref(d, j, ow, st, ob);
                    L.hd[(16 * u.wave + u.c) * NHP + j] = s + th[ob];
                }
            }
        }
        f32x4 acca[MT][NTW];
#ifndef RLC_NAF_SEPARATE_FWD
        // the mu and V branches read the same trunk image: one k-loop for both (mfma_blocks.h fwd_gemm2)
        u.fwd_gemm2(acca, acc, th + d.Wa2, th + d.Wv2, L2, L1);
        u.template bias_relu<0>(acca, th + d.ba2, L2);
        u.template row_dot<false, AD>(acca, L2, [&](int n, int j) { return L.wvec[j * 256 + n]; }, L.part_a);
        u.template store_masks<0, true>(acca, L2);
#else
        u.fwd_gemm(acca, th + d.Wa2, L2, L1);
        u.template bias_relu<0>(acca, th + d.ba2, L2);
        u.template row_dot<false, AD>(acca, L2, [&](int n, int j) { return L.wvec[j * 256 + n]; }, L.part_a);
        u.template store_masks<0, true>(acca, L2);
        if (u.split_mode((L2 + 15) >> 4)) lds_barrier();      // the split tile's hand-off buffer is reused
        u.fwd_gemm(acc, th + d.Wv2, L2, L1);
#endif
        u.template bias_relu<0>(acc, th + d.bv2, L2);
        u.template row_dot<false, 1>(acc, L2, [&](int n, int) { return wv3[n]; }, L.part_v);
        u.template store_masks<1, false>(acc, L2);
        lds_barrier();
        STAMP();
        // ================= 3: per sample: L columns, advantage, Q, and the seeds of every head's gradient =================
        for (int b = tid; b < B; b += kThreads) {
            float diff[AD], ddiff[AD], tanhv[AD];
#pragma unroll
            for (int j = 0; j < AD; j++) {
                tanhv[j] = tanhf(u.template part_sum<AD>(L.part_a, b, j) + th[d.ba3 + j]);
                diff[j] = L.a[b * AD + j] - tanhv[j] * amax[j];
                ddiff[j] = 0.0f;
            }
            const float V = u.template part_sum<1>(L.part_v, b, 0) + th[d.bv3];
            float hdv[NHP];
#pragma unroll
            for (int j = 0; j < NH; j++) hdv[j] = L.hd[b * NHP + j];
            float p[AD], l0[AD];
            float adv = 0.0f;
            int off = 0;
#pragma unroll
            for (int c = 0; c < AD; c++) {
                l0[c] = expf(fminf(fmaxf(hdv[c], -5.0f), 5.0f));
                float pc = diff[c] * l0[c];
#pragma unroll
                for (int k = 1; k < AD - c; k++) pc += diff[c + k] * hdv[AD + off + k - 1];
                off += AD - 1 - c;
                p[c] = pc;
                adv += pc * pc;
            }
            const float q = V + (-0.5f * adv);
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            dv.tap_V[(size_t)agent * RLC_MAX_BATCH + b] = V;
            const float dq = 2.0f * (q - L.y[b]);            // loss = SUM (y - q)^2
            L.dV[b] = dq;
            off = 0;
#pragma unroll
            for (int c = 0; c < AD; c++) {
                const float dp = -p[c] * dq;
                ddiff[c] += dp * l0[c];
#pragma unroll
                for (int k = 1; k < AD - c; k++) ddiff[c + k] += dp * hdv[AD + off + k - 1];
                const float xpre = hdv[c];
                L.dhd[b * NHP + c] = (xpre >= -5.0f && xpre <= 5.0f) ? dp * diff[c] * l0[c] : 0.0f;
#pragma unroll
                for (int k = 1; k < AD - c; k++) L.dhd[b * NHP + AD + off + k - 1] = dp * diff[c + k];
                off += AD - 1 - c;
            }
#pragma unroll
            for (int j = 0; j < AD; j++) L.dz[b * AD + j] = -ddiff[j] * amax[j] * (1.0f - tanhv[j] * tanhv[j]);
        }
        lds_barrier();
        STAMP();
        // ================= 4: output-layer / bias gradients from the live accumulators =================
        float g_wa3[NTW][AD], g_ba2[NTW], g_wv3[NTW], g_bv2[NTW];
        {
            const int NT = (L2 + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                const bool ok = t < NT && n < L2;
                float w3[AD], s3[AD];
#pragma unroll
                for (int j = 0; j < AD; j++) { w3[j] = ok ? L.wvec[j * 256 + n] : 0.0f; s3[j] = 0.0f; }
                const float wv = ok ? wv3[n] : 0.0f;
                float s2 = 0.0f, sv3 = 0.0f, sv2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 d4 = *reinterpret_cast<const lds_f32x4*>(&L.dV[16 * mt + 4 * u.g]);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int b = 16 * mt + 4 * u.g + r;
                        const float hv = acca[mt][i][r];
                        float dd = 0.0f;
#pragma unroll
                        for (int j = 0; j < AD; j++) {
                            const float dzb = L.dz[b * AD + j];
                            s3[j] += hv * dzb;
                            dd += dzb * w3[j];
                        }
                        s2 += hv > 0.0f ? dd : 0.0f;
                        const float gv = acc[mt][i][r];
                        sv3 += gv * d4[r];
                        sv2 += gv > 0.0f ? d4[r] * wv : 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < AD; j++) g_wa3[i][j] = col4_sum(s3[j]);
                g_ba2[i] = col4_sum(s2);
                g_wv3[i] = col4_sum(sv3);
                g_bv2[i] = col4_sum(sv2);
            }
        }
        STAMP();
        // ================= 5: d trunk = V branch (rank one) + mu branch (accumulated) + heads =================
        const float alpha = adam_alpha(dv.lr[agent], pw[0], pw[1]);
        u.template bwd_gemm<1, 1, false>(acc, th + d.Wv2, L2, L1, L.dV, wv3);
        if (u.split_mode((L1 + 15) >> 4)) lds_barrier();      // the split tile's hand-off buffer is reused
        u.template bwd_gemm<AD, 0, true>(acc, th + d.Wa2, L2, L1, L.dz, L.wvec);
        lds_barrier();
        // the first weight-gradient item's W / m / v / W' go in flight before the first-layer and head gradients
        typename U::WgPre2 pre;
#ifdef RLC_EARLY_PREFETCH
        constexpr int NPRE = 1;
        u.template wgrad_prefetch<false, 1>(pre, L2, th + d.Wa2, mm + d.Wa2, vv + d.Wa2, tt + d.Wa2);
#else
        constexpr int NPRE = 0;
#endif
        // heads off the trunk: dL/dh1[b][k] += sum_j dhd[b][j] * wh[j][k]  (dhd / wh rows beyond the NH heads in use are zero)
        STAMP();
        u.trunk_grad_adam(acc, th, mm, vv, alpha, d.W1, d.b1, tapg, tt, tau, L.x, HeadExtra{L.dhd, L.wh});
        STAMP();
        // head weights: g[k][j] = sum_b h1[b][k] dhd[b][j] (thread k), head biases (wave j)
        for (int k = tid; k < L1; k += kThreads) {
            float gs[NH];
#pragma unroll
            for (int j = 0; j < NH; j++) gs[j] = 0.0f;
            for (int b = 0; b < B; b++) {
                const float hv = L.hbuf[b * LDH + k];
                const f32x4 dh = *reinterpret_cast<const lds_f32x4*>(&L.dhd[b * NHP]);
#pragma unroll
                for (int j = 0; j < NH; j++) gs[j] += hv * dh[j];
            }
#pragma unroll
            for (int j = 0; j < NH; j++) {
                int ow, st, ob;
                naf_head_ref(d, j, ow, st, ob);
                U::adam_scalar(th, mm, vv, tt, tapg, ow + k * st, gs[j], alpha, tau);
            }
        }
        if (u.wave < NH) {
            const int j = u.wave;
            float gr = 0.0f;
            for (int b = u.lane; b < MB; b += 64) gr += L.dhd[b * NHP + j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
            if (u.lane == 0) {
                int ow, st, ob;
                naf_head_ref(d, j, ow, st, ob);
                U::adam_scalar(th, mm, vv, tt, tapg, ob, gr, alpha, tau);
            }
        }
        STAMP();
        // ================= 6: the two L1 x L2 matrices, Adam + Polyak in the GEMM epilogues =================
        u.template wgrad_adam_pre<AD, 0, 0, false, false, NPRE>(L.dz, nullptr, L2, th + d.Wa2, mm + d.Wa2, vv + d.Wa2, alpha,
                                        tapg ? tapg + d.Wa2 : nullptr, tt + d.Wa2, tau, L.wvec, pre);
        u.template wgrad_adam<1, 0, 1>(L.dV, nullptr, L2, th + d.Wv2, mm + d.Wv2, vv + d.Wv2, alpha,
                                       tapg ? tapg + d.Wv2 : nullptr, tt + d.Wv2, tau, wv3);
        {
            const int NT = (L2 + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2) {
                    // targets: 0 ba2[n], 1..A Wa3[n][j], A+1 Wv3[n], A+2 bv2[n]
                    for (int tg = u.g; tg < AD + 3; tg += 4) {
                        int p = d.ba2 + n;
                        float gr = g_ba2[i];
#pragma unroll
                        for (int j = 0; j < AD; j++)
                            if (tg == j + 1) { p = d.Wa3 + n * AD + j; gr = g_wa3[i][j]; }
                        if (tg == AD + 1) { p = d.Wv3 + n; gr = g_wv3[i]; }
                        if (tg == AD + 2) { p = d.bv2 + n; gr = g_bv2[i]; }
                        U::adam_scalar(th, mm, vv, tt, tapg, p, gr, alpha, tau);
                    }
                }
            }
            if (u.wave <= AD) {           // ba3[j] (wave j < A): sum_b dz[b][j]; bv3 (wave A): sum_b dV[b]
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += j < AD ? L.dz[b * AD + j] : L.dV[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) U::adam_scalar(th, mm, vv, tt, tapg, j < AD ? d.ba3 + j : d.bv3, gr, alpha, tau);
            }
        }
        __syncthreads();
        STAMP();
        if (tid == 0) { pw[0] *= 0.9f; pw[1] *= 0.999f; }
        __syncthreads();
    }
#undef STAMP
}

template <int MT, int NTW, int AD, bool T4>
int naf_launch_t(const RlcNafDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
                 int grad_taps, hipStream_t st, const RlcNafRollout* rollout) {
    constexpr int MSTRIDE = mask_stride(8 * NTW);
    const size_t lds = nsmem_carve<MSTRIDE>(dv.d, MT, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA NAF kernel needs %zu B of LDS (> 160 KiB)", lds);
    RLC_REQUIRE(!T4 || rlc_tail4(dv.d.B, MT), "tail-of-four kernel launched for batch %d", dv.d.B);
    auto kern = rlc_naf_update_mfma_kernel<MT, NTW, AD, T4>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source, idx_dev,
                       grad_taps, rollout);
    RLC_HIP(hipGetLastError());
    return 0;
}

}  // namespace
