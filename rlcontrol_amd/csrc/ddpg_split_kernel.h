// ddpg_split_kernel.h -- latency mode of the fused DDPG update: ONE agent's minibatch split over C workgroups (CUs).
//
// The fused kernel (ddpg_mfma_kernel.h) keeps an agent on one CU: 256 agents fill the chip, but the reference's own
// deployment unit -- one agent per process -- then runs at one CU's speed (~300 us per update).  Here workgroup c of C
// takes the batch rows [c*MB, (c+1)*MB) of every minibatch (MB = 16*MT) through the SAME nine contractions
// (mfma_blocks.h; the forward / backward chains of different rows are independent), produces PARTIAL weight gradients
// (trunk_grad_adam / wgrad_adam in gradient-only mode) into its own blob-shaped buffer, and after a barrier over the C
// workgroups each of them reduces one slice of the parameters over the C partials (fixed order) and applies Adam
// (+ Polyak) to that slice.  Per update (agents/DDPG.py:74-95 keeps its order: the actor phase sees the stepped critic
// and trunk):
//     targets, critic forward / backward on my rows -> critic partials -> BARRIER -> reduce + critic Adam on my slice
//     -> BARRIER -> actor forward, dQ/da, actor backward on my rows -> actor partials -> BARRIER -> reduce + actor Adam
//     + Polyak on my slice -> BARRIER
// Results equal the one-CU kernel up to the summation order over the batch (partial sums per CU, then over CUs).
// Hand-off rules (MI355X_MICROARCH.md, inter-workgroup visibility): every storing wave drains its stores
// (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane runs an agent-scope release, adds to the agent's arrival counter
// and polls it with L1-bypassing loads, then an agent-scope acquire (invalidates this CU's L1), workgroup barrier,
// plain loads.  The C workgroups of an agent must be co-resident: the launcher refuses grids above the CU count, and
// every poll loop is bounded (a timeout raises the error word instead of hanging the GPU).
// Workgroup -> (agent, c): workgroups are dealt round-robin to the 8 XCDs, so blockIdx = x + 8*(c + C*y) puts the C
// workgroups of agent 8y + x on XCD x (one L2: the partials never cross the fabric).  Correctness does not depend on it.
#pragma once
#include "ddpg_mfma_kernel.h"

namespace {

struct RlcSplit {
    float* part;            // [n_agents][C][Ppad] partial gradients, blob layout; zeroed once (pads stay zero)
    unsigned int* bar;      // [n_agents] monotonic arrival counters, zero at launch
    int* err;               // [1] set when a barrier poll timed out
    int C;
};

// barrier over the C workgroups of one agent (see the header comment); `gen` counts this workgroup's barriers
// Returns false -- for EVERY thread of the workgroup -- when the barrier did not complete (a poll timed out here or in
// another workgroup of the launch: the error word is set).  The caller returns at once: no Adam / Polyak store of a
// phase whose inputs are incomplete is ever issued, so parameters and optimizer state stay those of the last
// completed phase; the host reports the failure and poisons the handle (rlc_api.hip launch_update).
__device__ __forceinline__ bool split_barrier(unsigned int* ctr, int C, unsigned int& gen, int* err, mfb::lds_i32* failed /* one LDS word */) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gen += 1;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = gen * (unsigned int)C;
        int spins = 0, bad = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            // ~seconds: a peer is not resident -- or a peer has already given up and left (its error word is set)
            if (++spins > (1 << 22) || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                atomicExch(err, 1);
                bad = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // a workgroup that passed the poll still stops when another one of the launch has failed: nobody may go on
        // to reduce partials (or read an image) its peers have not finished
        *failed = bad | __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return *failed == 0;
}

// g[p] = sum_cc part[cc][p] for the float4 groups of [lo, hi) dealt to workgroup c; Adam on (th, m, v) with alpha;
// target update where pol_lo <= p < pol_hi; optional gradient tap.  lo, hi multiples of 4.
template <class U>
__device__ __forceinline__ void split_reduce_adam(const float* parts, size_t part_stride, int C, int c, int lo, int hi,
                                                  float* th, float* m, float* v, float alpha, float* tt, float tau,
                                                  int pol_lo, int pol_hi, float* tap) {
    const int n4 = (hi - lo) >> 2, per = (n4 + C - 1) / C;
    const int q0 = c * per, q1 = min(n4, q0 + per);
    for (int q = q0 + (int)threadIdx.x; q < q1; q += kThreads) {
        const int p = lo + 4 * q;
        f32x4 g = *reinterpret_cast<const f32x4*>(&parts[p]);
        for (int cc = 1; cc < C; cc++) g += *reinterpret_cast<const f32x4*>(&parts[(size_t)cc * part_stride + p]);
        f32x4 w = *reinterpret_cast<const f32x4*>(&th[p]);
        f32x4 mm = *reinterpret_cast<const f32x4*>(&m[p]);
        f32x4 vv = *reinterpret_cast<const f32x4*>(&v[p]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float a = mm[r], b = vv[r];
            w[r] = adam_step_fast(w[r], g[r], a, b, alpha);
            mm[r] = a; vv[r] = b;
        }
        *reinterpret_cast<f32x4*>(&th[p]) = w;
        *reinterpret_cast<f32x4*>(&m[p]) = mm;
        *reinterpret_cast<f32x4*>(&v[p]) = vv;
        if (tap) *reinterpret_cast<f32x4*>(&tap[p]) = g;
        if (p >= pol_lo && p < pol_hi) {
            f32x4 t = *reinterpret_cast<const f32x4*>(&tt[p]);
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = U::polyak(t[r], w[r], tau);
            *reinterpret_cast<f32x4*>(&tt[p]) = t;
        }
    }
}

template <int MT, int AD>
__global__ __launch_bounds__(kThreads) void rlc_ddpg_update_split_kernel(RlcDev dv, RlcSplit sp, int first_agent,
                                                                         int n_agents, int n_updates, int source,
                                                                         const long long* host_idx, int grad_taps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using U = Blk<MT, NTW, MSTRIDE>;
    constexpr int MB = U::MB;
    const RlcDims d = dv.d;
    const int C = sp.C;
    // workgroup -> (agent, c), the C workgroups of an agent on one XCD
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int c = jj % C, rel = (jj / C) * 8 + xcd;
    if (rel >= n_agents) return;                    // whole workgroups of a padded grid: no barrier includes them
    const int agent = first_agent + rel;
    U u;
    u.init_geometry();
    const int Bfull = d.B;
    const int row0 = c * MB;                                         // first batch row of this workgroup
    const int B = max(0, min(MB, Bfull - row0));                     // rows it owns (may be 0)
    u.S = d.S; u.H1 = d.H1; u.B = B; u.LDH = ldh_for(d.H1);
    Smem L;
    smem_carve(d, MT, (lds_u8*)smem, &L);
    u.L.hbuf = L.hbuf; u.L.mask = L.mask; u.L.xbuf = MT >= 2 ? L.xbuf : nullptr;
    const int tid = u.tid, S = d.S, H1 = d.H1, HA = d.HA, HC = d.HC;

    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* m_a = dv.m_a + (size_t)agent * d.Ppad;
    float* v_a = dv.v_a + (size_t)agent * d.Ppad;
    float* m_c = dv.m_c + (size_t)agent * d.Ppad;
    float* v_c = dv.v_c + (size_t)agent * d.Ppad;
    float* pw = dv.pw + agent * 4;
    const float lr_a = dv.actor_lr[agent], lr_c = dv.critic_lr[agent], tau = dv.tau;
    float* tap_gc = grad_taps ? dv.tap_gc + (size_t)agent * d.Ppad : nullptr;
    float* tap_ga = grad_taps ? dv.tap_ga + (size_t)agent * d.Ppad : nullptr;
    const size_t pstride = d.Ppad;
    float* parts = sp.part + (size_t)agent * C * pstride;            // the agent's C partial blobs
    float* mine = parts + (size_t)c * pstride;
    unsigned int* ctr = sp.bar + agent;
    unsigned int gen = 0;
    float amax[AD];
#pragma unroll
    for (int j = 0; j < AD; j++) amax[j] = dv.amax[j];
    // beta powers and the sampler's call counter advance identically on every workgroup; workgroup 0 stores them back
    float pw0 = pw[0], pw1 = pw[1], pw2 = pw[2], pw3 = pw[3];
    const unsigned long long call0 = dv.rep.sample_ctr[agent];

    for (int i = tid; i < MB * AD; i += kThreads) { L.a[i] = 0.f; L.aout[i] = 0.f; L.mu[i] = 0.f; L.dz[i] = 0.f; }
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.x2[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) { L.q[i] = 0.f; L.y[i] = 0.f; L.dq[i] = 0.f; }
    for (int i = tid; i < MB * MSTRIDE / 4; i += kThreads) reinterpret_cast<lds_u32*>(L.mask)[i] = 0u;
    if (tid < 16) L.hbuf[MB * u.LDH + tid] = 0.0f;
    __syncthreads();

    f32x4 acc[MT][NTW];
    for (int upd = 0; upd < n_updates; upd++) {
        asm volatile("" : "+v"(u.c), "+v"(u.g), "+s"(u.wave));
        // ================= sample (every workgroup draws the whole index set) + gather of my rows =================
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            __syncthreads();
            rlc_sample_distinct(ring.size, Bfull, dv.rep.seed[agent], call0 + upd, L.pool, L.idx, L.dups);
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < Bfull; b += kThreads)
                L.idx[b] = host_idx[((size_t)rel * n_updates + upd) * Bfull + b];
        }
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const int bg = row0 + b;
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + bg;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot * AD; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = dv.rep.gr[slot]; L.g[b] = dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[bg]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot * AD; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = dv.rep.rr[slot]; L.g[b] = dv.rep.rg[slot];
            }
            for (int i = 0; i < S; i++) {
                L.x[b * SMAX + i] = clip_state_val(ps[i], dv.clip_state, dv.smin[i], dv.smax[i]);
                L.x2[b * SMAX + i] = clip_state_val(ps2[i], dv.clip_state, dv.smin[i], dv.smax[i]);
            }
#pragma unroll
            for (int j = 0; j < AD; j++) L.a[b * AD + j] = pa[j];
        }
        __syncthreads();

        // ================= steps 1-2: target networks on s' =================
        u.trunk(tt + d.oW1, tt + d.ob1, L.x2);
        __syncthreads();
        u.fwd_gemm(acc, tt + d.oWa2, HA, H1);
        u.template bias_relu<0>(acc, tt + d.oba2, HA);
        u.template row_dot<false, AD>(acc, HA, [&](int n, int j) { return tt[d.oWa3 + n * AD + j]; }, L.part);
        __syncthreads();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            L.aout[i] = tanhf(u.template part_sum<AD>(L.part, b, j) + tt[d.oba3 + j]) * amax[j];
        }
        __syncthreads();
        u.fwd_gemm(acc, tt + d.oWc2, HC, H1);
        u.template bias_relu<AD>(acc, tt + d.obc2, HC, L.aout, tt + d.oWc2, d.arow0);
        u.template row_dot<false, 1>(acc, HC, [&](int n, int) { return tt[d.oWc3 + n]; }, L.part);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float qt = u.template part_sum<1>(L.part, b, 0) + tt[d.obc3];
            const float y = (float)(L.r[b] + L.g[b] * (double)qt);
            L.y[b] = y;
            dv.tap_y[(size_t)agent * RLC_MAX_BATCH + row0 + b] = y;
        }
        __syncthreads();

        // ================= step 3: critic forward / backward on my rows, partial gradients =================
        u.trunk(th + d.oW1, th + d.ob1, L.x);
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < HC ? th[d.oWc3 + n] : 0.0f;
        __syncthreads();
        u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        u.template bias_relu<AD>(acc, th + d.obc2, HC, L.a, th + d.oWc2, d.arow0);
        u.template row_dot<false, 1>(acc, HC, [&](int n, int) { return th[d.oWc3 + n]; }, L.part);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float q = u.template part_sum<1>(L.part, b, 0) + th[d.obc3];
            L.q[b] = q;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + row0 + b] = q;
            L.dq[b] = 2.0f * (q - L.y[b]) / (float)Bfull;              // mean over the WHOLE minibatch
        }
        __syncthreads();
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
                s3 = col4_sum(s3);
                s2 = col4_sum(s2);
                if (t < NT && n < HC && u.g < 2) mine[(u.g == 0) ? d.oWc3 + n : d.obc2 + n] = (u.g == 0) ? s3 : s2;
            }
            if (u.wave == 0) {
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dq[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) mine[d.obc3] = gr;
            }
        }
        u.template store_masks<-2, true>(acc, HC);
        __syncthreads();
        u.template bwd_gemm<1, -2>(acc, th + d.oWc2, HC, H1, L.dq, L.wvec);
        __syncthreads();
        u.template trunk_grad_adam<NoExtra, true>(acc, nullptr, nullptr, nullptr, 0.0f, d.oW1, d.ob1, mine, nullptr, 0.0f, L.x);
        u.template wgrad_adam<1, AD, -2, true>(L.dq, L.a, HC, nullptr, nullptr, nullptr, 0.0f, mine + d.oWc2, nullptr, 0.0f,
                                               L.wvec);
        if (!split_barrier(ctr, C, gen, sp.err, L.dups + 3)) return;
        // ---- reduce + critic Adam on my slice: trunk [0, oWa2) without target update, critic block with it ----
        {
            const float alpha_c = adam_alpha(lr_c, pw2, pw3);
            split_reduce_adam<U>(parts, pstride, C, c, 0, d.oWa2, th, m_c, v_c, alpha_c, tt, tau, 0, 0, tap_gc);
            split_reduce_adam<U>(parts, pstride, C, c, d.ocritic0, d.Pdev, th, m_c, v_c, alpha_c, tt, tau, d.ocritic0,
                                 d.Pdev, tap_gc);
            pw2 *= 0.9f; pw3 *= 0.999f;
        }
        if (!split_barrier(ctr, C, gen, sp.err, L.dups + 3)) return;

        // ================= step 4: actor forward with the updated trunk =================
        u.trunk(th + d.oW1, th + d.ob1, L.x);
        for (int i = tid; i < AD * 256; i += kThreads) {
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < HA ? th[d.oWa3 + n * AD + j] : 0.0f;
        }
        __syncthreads();
        u.fwd_gemm(acc, th + d.oWa2, HA, H1);
        u.template bias_relu<0>(acc, th + d.oba2, HA);
        u.template row_dot<false, AD>(acc, HA, [&](int n, int j) { return th[d.oWa3 + n * AD + j]; }, L.part);
        u.template store_masks<-2, true>(acc, HA);
        __syncthreads();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float mu = tanhf(u.template part_sum<AD>(L.part, b, j) + th[d.oba3 + j]);
            L.mu[i] = mu;
            const float ao = mu * amax[j];
            L.aout[i] = ao;
            dv.tap_aout[((size_t)agent * RLC_MAX_BATCH + row0) * AD + i] = ao;
        }
        __syncthreads();
        f32x4 h2acc[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) h2acc[mt][i] = acc[mt][i];

        // ================= step 5: dQ/da at the scaled action, updated critic =================
        u.fwd_gemm(acc, th + d.oWc2, HC, H1);
        u.template bias_relu<AD>(acc, th + d.obc2, HC, L.aout, th + d.oWc2, d.arow0);
        u.template row_dot<true, AD>(acc, HC, [&](int n, int j) { return th[d.oWc2 + rlc_blk_index(d.arow0 + j, n, HC)] * th[d.oWc3 + n]; },
                                     L.part);
        __syncthreads();
        for (int i = tid; i < B * AD; i += kThreads) {
            const int b = i / AD, j = i % AD;
            const float dqda = u.template part_sum<AD>(L.part, b, j);
            dv.tap_dqda[((size_t)agent * RLC_MAX_BATCH + row0) * AD + i] = dqda;
            const float mu = L.mu[i];
            L.dz[i] = -dqda * (1.0f - mu * mu);
        }
        __syncthreads();

        // ================= step 6: actor backward on my rows, partial gradients =================
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
                s2 = col4_sum(s2);
#pragma unroll
                for (int j = 0; j < AD; j++) s3[j] = col4_sum(s3[j]);
                if (ok && u.g <= AD) {
                    int p = d.oba2 + n;
                    float gr = s2;
#pragma unroll
                    for (int j = 0; j < AD; j++)
                        if (u.g == j + 1) { p = d.oWa3 + n * AD + j; gr = s3[j]; }
                    mine[p] = gr;
                }
            }
            if (u.wave < AD) {
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dz[b * AD + j];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) gr += __shfl_xor(gr, off, 64);
                if (u.lane == 0) mine[d.oba3 + j] = gr;
            }
        }
        u.template bwd_gemm<AD, -2>(acc, th + d.oWa2, HA, H1, L.dz, L.wvec);
        __syncthreads();
        u.template trunk_grad_adam<NoExtra, true>(acc, nullptr, nullptr, nullptr, 0.0f, d.oW1, d.ob1, mine, nullptr, 0.0f, L.x);
        u.template wgrad_adam<AD, 0, -2, true>(L.dz, nullptr, HA, nullptr, nullptr, nullptr, 0.0f, mine + d.oWa2, nullptr, 0.0f,
                                               L.wvec);
        if (!split_barrier(ctr, C, gen, sp.err, L.dups + 3)) return;
        {
            const float alpha_a = adam_alpha(lr_a, pw0, pw1);
            split_reduce_adam<U>(parts, pstride, C, c, 0, d.ocritic0, th, m_a, v_a, alpha_a, tt, tau, 0, d.ocritic0, tap_ga);
            pw0 *= 0.9f; pw1 *= 0.999f;
        }
        if (!split_barrier(ctr, C, gen, sp.err, L.dups + 3)) return;
    }
    if (c == 0 && tid == 0) {
        pw[0] = pw0; pw[1] = pw1; pw[2] = pw2; pw[3] = pw3;
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) dv.rep.sample_ctr[agent] = call0 + n_updates;
    }
}

template <int MT, int AD>
int launch_split_t(const RlcDev& dv, const RlcSplit& sp, int first_agent, int n_agents, int n_updates, int source,
                   const long long* idx_dev, int grad_taps, hipStream_t st) {
    const size_t lds = smem_carve(dv.d, MT, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "split DDPG kernel needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = rlc_ddpg_update_split_kernel<MT, AD>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int groups = (n_agents + 7) / 8;
    hipLaunchKernelGGL(kern, dim3(groups * 8 * sp.C), dim3(kThreads), lds, st, dv, sp, first_agent, n_agents, n_updates,
                       source, idx_dev, grad_taps);
    RLC_HIP(hipGetLastError());
    return 0;
}

}  // namespace
