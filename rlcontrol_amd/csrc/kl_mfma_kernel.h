// kl_mfma_kernel.h -- fused ReverseKL / ForwardKL update on gfx950 fp32 matrix cores.
//
// Same contract as kl_generic.hip (one workgroup per agent, n_updates sequential updates per launch, every update =
// sample_batch + update_network + update_target_network of agents/ReverseKL.py:83-93 / agents/ForwardKL.py), built from
// the MFMA blocks of mfma_blocks.h.  The three small networks are SoftActorCritic's family (sac_mfma_kernel.h) with
// Q's action as one more INPUT column of its first layer (reversekl_network.py:257-276: xq = [s | a] goes through the
// same first-layer block as a state of dimension S+1), torch's Adam (Blk<..., TADAM>), and a target for V only.
// What is particular to these agents is the action integral: Q at the B x K (state, node) pairs,
//     [B*K, L1c] x [L1c, L2c]   (1984 x 200 x 200 for the shipped jsons: 87 % of the update's flops),
// runs as ceil(B*K / 16 MTQ) passes of the forward GEMM block at MTQ batch tiles: per pass the rows
// relu(z1s[b] + a_k * W1[action row]) -- the first layer is separable, z1s = s W1[:S] + b1 is formed once per state --
// are GENERATED into the LDS activation image, the GEMM streams qW2 from L2, and bias / relu / the qW3 head fold into
// its epilogue; only the B*K head values leave the CU (a global scratch row, 8 KB).
//
//   1  V'(s')      2  pi forward, draw, log pi     3  Q(s,a) + seeds     4  Q(s,a_new)     5  V(s)
//   6  Q at the nodes (passes)       7  per state: log pi at the nodes, integrand derivative -> seeds of mean / log_std
//   8  pi step     9  Q step     10  V forward again (masks, accumulators) + V step + Polyak(V)
// Everything reads the PRE-update weights until its own network is stepped (the reference builds the three losses
// before the first optimizer.step(), reversekl_network.py:139-218).
//
// Supported: action_dim 1, S <= 7, widths multiples of 4 in [16, 256], B <= 32, K <= 256 nodes, LDS <= 160 KiB.
#pragma once
#include "mfma_blocks.h"
#include "sac_rollout_device.h"
#include "sac_common.h"

namespace {

using namespace mfb;

constexpr int KL_NTW = 2;
constexpr int KL_MSTRIDE = mask_stride(16);
constexpr int KL_MAXNODES = 256;

// ---- latency mode (rlc_kl_set_split): the node passes of ONE agent's action integral dealt over C workgroups ----
// Two thirds of an update are the ceil(B K / 112) forward passes of Q at the (state, node) pairs, and they are independent
// of each other.  Workgroup 0 of an agent (the owner) runs the whole update as the one-workgroup kernel does; workgroups
// 1..C-1 (helpers) only take their share of the passes: per update the owner publishes the first-layer image z1s
// (global, [MB][LDH]) -> BARRIER -> every workgroup runs the passes p with p mod C == its index and writes Q at those
// rows to the agent's iq buffer -> BARRIER -> the owner carries on.  The arithmetic of a pass does not depend on who
// runs it, so the result is bit-identical to the one-workgroup kernel's.  Hand-off as in ddpg_split_kernel.h: stores
// drained, workgroup barrier, agent-scope release, relaxed arrival counter, bounded L1-bypassing poll (a timeout sets
// the error word instead of hanging), agent-scope acquire, workgroup barrier.  blockIdx = x + 8 (c + C y): the C
// workgroups of agent 8y + x share XCD x (one L2 holds the weights and the exchange).
struct KlSplit {
    float* zbuf;            // [n_agents][MB * LDH] first-layer image of the minibatch (owner -> helpers)
    unsigned int* bar;      // [n_agents] monotonic arrival counters, zero at launch
    int* err;               // [1] set when a barrier poll timed out
    int C, n_agents;
};

// Returns false -- for EVERY thread of the workgroup -- when the barrier did not complete (a poll timed out here or in
// another workgroup of the launch: the error word is set).  The caller returns at once: no Adam / Polyak store of a
// phase whose inputs are incomplete is ever issued, so parameters and optimizer state stay those of the last
// completed phase; the host reports the failure and poisons the handle (rlc_api.hip launch_update).
__device__ __forceinline__ bool kl_group_barrier(unsigned int* ctr, int C, unsigned int& gen, int* err, mfb::lds_i32* failed /* one LDS word */) {
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

struct KSmem {
    lds_f32* hbuf;                       // [MBQ][LDH] (+16): activation image of the GEMM in flight
    lds_u8* mask;                        // [MB][MSTRIDE] bit 0: pi hidden (later V hidden), bit 1: Q(s,a) hidden
    lds_f32* part_h;                     // [kWaves][MB][2]   mean | log_std head partials
    lds_f32* part_q;                     // [kWaves][MBQ]     one-column head partials (Q, V, Q at the nodes)
    lds_f32* wvec;                       // [2][256] staged output-layer weights
    lds_f32 *x, *x2, *xq, *xn;           // [MB][SMAX]  s | s' | [s, a] | [s, a_new]
    lds_f32* z1s;                        // [MB][LDH]   Q's first-layer pre-activation without the action
    lds_f32* w1a;                        // [256]       Q's first-layer action row
    lds_f32 *node_a, *node_w, *node_u, *node_j;     // [KL_MAXNODES] node action, weight, atanh(a/amax), log(1 - (a/amax)^2 + 1e-6)
    lds_f32 *a, *eps, *mu, *lsr, *sd, *z, *lp, *pls, *r, *g, *vt, *q, *qn, *v, *dq, *dvs;   // [MB]
    lds_f32* dml;                        // [MB][2] seeds of the pi heads: d mean | d log_std (pre-clamp)
    lds_f32* red;                        // 16
    lds_f32* adam;                       // 4: alpha_pi, alpha_qv, eps
    lds_i64* idx;
    lds_i32* pool;
    lds_i32* dups;
    lds_f32x4* xbuf;
};

__host__ __device__ inline int kl_mfma_ldh(const RlcSacDims& d) { return ldh_for(d.L1A > d.L1C ? d.L1A : d.L1C); }
__host__ __device__ constexpr bool kl_z1s_in_global(int MT) { return MT > 2; }
// floats of the agent's scratch row the MFMA kernel uses: Q at the B x K nodes, then (more than two batch tiles) z1s
__host__ __device__ inline size_t kl_mfma_scratch_floats(const RlcSacDims& d, int nodes, int MT) {
    const size_t rows = ((size_t)d.B * (size_t)nodes + 63) & ~(size_t)63;
    return rows + (kl_z1s_in_global(MT) ? (size_t)MT * 16 * kl_mfma_ldh(d) : 0);
}

__host__ __device__ inline size_t ksmem_carve(const RlcSacDims& d, int MT, int MTQ, lds_u8* base, KSmem* out) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        lds_u8* p = base + off;
        off += (bytes + 15) & ~(size_t)15;
        return p;
    };
    const int MB = MT * 16, MBQ = MTQ * 16, LDH = kl_mfma_ldh(d);
    const int MTX = MTQ > MT ? MTQ : MT;
    KSmem L;
    L.hbuf = (lds_f32*)take(sizeof(float) * (MTX * 16 * LDH + 16));
    L.idx = (lds_i64*)take(sizeof(long long) * RLC_MAX_BATCH);
    L.mask = take((size_t)MB * KL_MSTRIDE);
    // the two partial buffers are never live together (a barrier separates every use of one from the next use of the
    // other): one region of the larger size
    {
        const size_t bh = sizeof(float) * kWaves * MB * 2, bq = sizeof(float) * kWaves * MTX * 16;
        L.part_h = (lds_f32*)take(bh > bq ? bh : bq);
        L.part_q = L.part_h;
    }
    L.wvec = (lds_f32*)take(sizeof(float) * 2 * 256);
    // x (s) is a prefix of xq ([s, a]): the first-layer passes of pi and V read S columns of it and multiply the rest by
    // zero weights, so the two share one array
    lds_f32** ps[] = {&L.x2, &L.xq, &L.xn};
    for (auto p : ps) *p = (lds_f32*)take(sizeof(float) * MB * SMAX);
    L.x = L.xq;
    // at more than two batch tiles the first-layer image of Q (MB x LDH floats: 90 KB at seven tiles) lives in the agent's
    // global scratch instead (kl_z1s_in_global): the node passes read two or three of its rows each, L1-resident
    L.z1s = kl_z1s_in_global(MT) ? nullptr : (lds_f32*)take(sizeof(float) * MB * LDH);
    L.w1a = (lds_f32*)take(sizeof(float) * 256);
    lds_f32** pn[] = {&L.node_a, &L.node_w, &L.node_u, &L.node_j};
    for (auto p : pn) *p = (lds_f32*)take(sizeof(float) * KL_MAXNODES);
    lds_f32** pb[] = {&L.a, &L.eps, &L.mu, &L.lsr, &L.sd, &L.z, &L.lp, &L.pls, &L.r, &L.g, &L.vt, &L.q, &L.qn, &L.v,
                      &L.dq, &L.dvs};
    for (auto p : pb) *p = (lds_f32*)take(sizeof(float) * MB);
    L.dml = (lds_f32*)take(sizeof(float) * MB * 2);
    L.red = (lds_f32*)take(sizeof(float) * 16);
    L.adam = (lds_f32*)take(sizeof(float) * 4);
    L.pool = (lds_i32*)take(sizeof(int) * 3 * RLC_MAX_BATCH);
    L.dups = (lds_i32*)take(sizeof(int) * 4);
    L.xbuf = (lds_f32x4*)take(sizeof(float) * 4 * 64 * (MTX - (MTX + 3) / 4));
    if (out) *out = L;
    return off;
}

__device__ inline float kl_blk_sum(float v, lds_f32* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < kWaves; w++) s += red[w];
    __syncthreads();
    return s;
}

__device__ __forceinline__ float kl_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int MT, int MTQ, bool SPLIT = false>
__global__ __launch_bounds__(kThreads) void rlc_kl_update_mfma_kernel(RlcSacDev dv, int first_agent, int n_updates,
                                                                      int source, const long long* host_idx,
                                                                      const float* eps_in, int grad_taps,
                                                                      const RlcSacRollout* rollout, KlSplit sp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using U = Blk<MT, KL_NTW, KL_MSTRIDE, true, true>;      // the three small networks: LERP target update, torch Adam
    using UQ = Blk<MTQ, KL_NTW, KL_MSTRIDE, true, true>;    // the node passes (forward only)
    constexpr int MB = U::MB, MBQ = UQ::MB, NTW = KL_NTW, NS = 2;
    const RlcSacDims d = dv.d;
    KSmem L;
    ksmem_carve(d, MT, MTQ, (lds_u8*)smem, &L);
    U u;
    u.init_geometry();
    u.S = d.S; u.H1 = d.L1A; u.B = d.B; u.LDH = kl_mfma_ldh(d);
    u.L.hbuf = L.hbuf; u.L.mask = L.mask; u.L.xbuf = L.xbuf;
    UQ uq;
    uq.init_geometry();
    uq.S = d.S; uq.H1 = d.L1C; uq.B = MBQ; uq.LDH = u.LDH;
    uq.L.hbuf = L.hbuf; uq.L.mask = L.mask; uq.L.xbuf = L.xbuf;
    const int tid = u.tid, S = d.S, L1A = d.L1A, L2A = d.L2A, L1C = d.L1C, L2C = d.L2C, B = d.B, K = dv.kl_nodes;
    const int LDH = u.LDH;
    // SPLIT: workgroup -> (agent, c), the C workgroups of an agent on one XCD; c == 0 owns the update
    int c_split = 0, rel_agent = blockIdx.x;
    if constexpr (SPLIT) {
        const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
        c_split = jj % sp.C;
        rel_agent = (jj / sp.C) * 8 + xcd;
        if (rel_agent >= sp.n_agents) return;       // whole workgroups of a padded grid: no barrier includes them
    }
    const int agent = first_agent + rel_agent;
    unsigned int bar_gen = 0;
    float* th = dv.theta + (size_t)agent * d.Ppad;
    float* tt = dv.theta_t + (size_t)agent * d.Ppad;
    float* mm = dv.m + (size_t)agent * d.Ppad;
    float* vv = dv.v + (size_t)agent * d.Ppad;
    float* tapg = grad_taps ? dv.tap_g + (size_t)agent * d.Ppad : nullptr;
    const float alpha_ent = dv.alpha[agent], amax0 = dv.amax0, tau = dv.tau;
    const bool integral = dv.kl_optim == RLC_KL_OPTIM_INTG || dv.kl_optim == RLC_KL_OPTIM_HARD_INTG;
    const int rows = B * K;
    float* iq = dv.scratch + (size_t)agent * dv.scratch_stride;            // [rows] Q at the nodes
    constexpr bool ZG = kl_z1s_in_global(MT);
    static_assert(!(ZG && SPLIT), "latency mode keeps the first-layer image in LDS (two batch tiles)");
    float* z1g = iq + (((size_t)rows + 63) & ~(size_t)63);                  // ZG: [MB][LDH] first-layer image of Q
    const float LOG_SQRT_2PI = 0.9189385332046727f, EPS = 1e-6f;
    const float invB = 1.0f / (float)B;

    // once per launch: zero the padded tails, stage the node tables
    for (int i = tid; i < MB * SMAX; i += kThreads) { L.x[i] = 0.f; L.x2[i] = 0.f; L.xq[i] = 0.f; L.xn[i] = 0.f; }
    for (int i = tid; i < MB; i += kThreads) {
        L.a[i] = 0.f; L.eps[i] = 0.f; L.mu[i] = 0.f; L.lsr[i] = 0.f; L.sd[i] = 0.f; L.z[i] = 0.f; L.lp[i] = 0.f; L.pls[i] = 0.f;
        L.r[i] = 0.f; L.g[i] = 0.f; L.vt[i] = 0.f; L.q[i] = 0.f; L.qn[i] = 0.f; L.v[i] = 0.f; L.dq[i] = 0.f; L.dvs[i] = 0.f;
    }
    for (int i = tid; i < MB * NS; i += kThreads) L.dml[i] = 0.f;
    if constexpr (!ZG) for (int i = tid; i < MB * LDH; i += kThreads) L.z1s[i] = 0.f;
    for (int i = tid; i < MB * KL_MSTRIDE / 4; i += kThreads) reinterpret_cast<lds_u32*>(L.mask)[i] = 0u;
    for (int k = tid; k < KL_MAXNODES; k += kThreads) {
        const bool live = k < K;
        const float an = live ? dv.kl_node_a[k] / amax0 : 0.0f;
        L.node_a[k] = live ? dv.kl_node_a[k] : 0.0f;
        L.node_w[k] = live ? dv.kl_node_w[k] : 0.0f;
        L.node_u[k] = (logf(1.0f + an) - logf(1.0f - an)) / 2.0f;       // PolicyNetwork.atanh (reversekl_network.py:384)
        L.node_j[k] = logf(1.0f - an * an + EPS);
    }
    if (tid < 16) L.hbuf[(MBQ > MB ? MBQ : MB) * LDH + tid] = 0.0f;
    __syncthreads();

    // Q at the rows [p0, p0 + 16 MTQ) of the B x K (state, node) grid for every pass p0 this workgroup takes (SPLIT: pass
    // index mod C == c_split): layer 1 = relu(z1s[b] + a_k W1[action row]) written straight into the activation image
    const int NQ = LDH >> 2;                           // column quads of a row of the activation image
    auto node_passes = [&]() {
        const int cq = tid % NQ, r0t = tid / NQ, rstep = kThreads / NQ;
        const bool filler = tid < rstep * NQ;
        const f32x4 wa = *reinterpret_cast<const lds_f32x4*>(&L.w1a[cq << 2]);
        const float qb3 = th[d.qb3];
        for (int p0 = 0; p0 < rows; p0 += MBQ) {
            if (SPLIT && (p0 / MBQ) % sp.C != c_split) continue;
            const int nr = min(MBQ, rows - p0);
            if (filler)
                for (int i = r0t; i < MBQ; i += rstep) {
                    f32x4 o = {0.f, 0.f, 0.f, 0.f};
                    if (i < nr) {
                        const int rho = p0 + i, b = rho / K, k = rho - b * K;
                        f32x4 zz;
                        if constexpr (ZG) zz = *reinterpret_cast<const f32x4*>(&z1g[b * LDH + (cq << 2)]);
                        else zz = *reinterpret_cast<const lds_f32x4*>(&L.z1s[b * LDH + (cq << 2)]);
                        const float ak = L.node_a[k];
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = fmaxf(zz[e] + ak * wa[e], 0.0f);
                    }
                    *reinterpret_cast<lds_f32x4*>(&L.hbuf[i * LDH + (cq << 2)]) = o;
                }
            __syncthreads();
            f32x4 accn[MTQ][NTW];
            uq.fwd_gemm(accn, th + d.qW2, L2C, L1C);
            uq.template bias_relu<0>(accn, th + d.qb2, L2C);
            uq.template row_dot<false, 1>(accn, L2C, [&](int n, int) { return th[d.qW3 + n]; }, L.part_q);
            __syncthreads();
            for (int i = tid; i < nr; i += kThreads) {
                const float qv = uq.template part_sum<1>(L.part_q, i, 0) + qb3;
                iq[p0 + i] = qv;
                dv.kl_tap_iq[(size_t)agent * rows + p0 + i] = qv;
            }
        }
    };
    if constexpr (SPLIT) {
        if (c_split > 0) {
            // helper: per update, the owner's z1s image and the current Q weights -> my passes
            float* zb = sp.zbuf + (size_t)rel_agent * MB * LDH;
            for (int upd = 0; upd < n_updates; upd++) {
                if (!kl_group_barrier(sp.bar + rel_agent, sp.C, bar_gen, sp.err, L.dups + 3)) return;
                if constexpr (!ZG)
                    for (int i = tid; i < (MB * LDH) >> 2; i += kThreads)
                        reinterpret_cast<lds_f32x4*>(L.z1s)[i] = reinterpret_cast<const f32x4*>(zb)[i];
                for (int n = tid; n < 256; n += kThreads) L.w1a[n] = n < L1C ? th[d.qW1 + S * L1C + n] : 0.0f;
                __syncthreads();
                node_passes();
                if (!kl_group_barrier(sp.bar + rel_agent, sp.C, bar_gen, sp.err, L.dups + 3)) return;
            }
            return;
        }
    }

    f32x4 acc[MT][NTW];
    for (int upd = 0; upd < n_updates; upd++) {
        asm volatile("" : "+v"(u.c), "+v"(u.g), "+s"(u.wave));     // see ddpg_mfma_kernel.h
        if (rollout) {
            // on-device experiment loop: one environment step first (hbuf is free here); update when learn() would run
            if (!rlc_sac_train_step_device(rollout, agent, (float*)L.hbuf)) continue;
        }
        // ================= sample + gather =================
        const RlcRingMeta ring = dv.rep.ring[agent];
        if (source == RLC_SRC_REPLAY_DEVICE_SAMPLER) {
            const unsigned long long call = dv.rep.sample_ctr[agent];
            __syncthreads();
            rlc_sample_distinct(ring.size, B, dv.rep.seed[agent], call, L.pool, L.idx, L.dups);
            if (tid == 0) dv.rep.sample_ctr[agent] = call + 1;
        } else if (source == RLC_SRC_REPLAY_HOST_INDICES) {
            for (int b = tid; b < B; b += kThreads) L.idx[b] = host_idx[((size_t)rel_agent * n_updates + upd) * B + b];
        }
        __syncthreads();
        const unsigned long long nctr = dv.noise_ctr[agent];
        const int step = dv.kl_step[agent] + 1;
        for (int b = tid; b < B; b += kThreads) {
            const float *ps, *pa, *ps2;
            if (source == RLC_SRC_STAGING) {
                const size_t slot = (size_t)agent * RLC_MAX_BATCH + b;
                ps = dv.rep.gs + slot * S; pa = dv.rep.ga + slot; ps2 = dv.rep.gs2 + slot * S;
                L.r[b] = (float)dv.rep.gr[slot]; L.g[b] = (float)dv.rep.gg[slot];
            } else {
                const size_t slot = (size_t)agent * dv.rep.cap + ring_slot(ring, dv.rep.cap, L.idx[b]);
                ps = dv.rep.rs + slot * S; pa = dv.rep.ra + slot; ps2 = dv.rep.rs2 + slot * S;
                L.r[b] = (float)dv.rep.rr[slot]; L.g[b] = (float)dv.rep.rg[slot];
            }
            for (int i = 0; i < S; i++) {
                const float sv = ps[i];
                L.x[b * SMAX + i] = sv; L.xq[b * SMAX + i] = sv; L.xn[b * SMAX + i] = sv;
                L.x2[b * SMAX + i] = ps2[i];
            }
            L.a[b] = pa[0];
            L.xq[b * SMAX + S] = pa[0];
            float e;
            if (eps_in) {
                e = eps_in[((size_t)rel_agent * n_updates + upd) * B + b];
            } else {
                const Philox4 p = philox4x32_10(dv.rep.seed[agent] ^ RLC_KEY_SAC_EPS, nctr, (unsigned long long)b >> 1);
                float n0, n1;
                philox_normal2(p, n0, n1);
                e = (b & 1) ? n1 : n0;
            }
            L.eps[b] = e;
        }
        if (tid == 0) {
            // torch's Adam as the TF-form step with alpha = lr * c / (1 - b1^t), epsilon = 1e-8 * c, c = sqrt(1 - b2^t)
            const double c = sqrt(1.0 - pow(0.999, (double)step)), bc1 = 1.0 - pow(0.9, (double)step);
            L.adam[0] = (float)((double)dv.pi_lr[agent] * c / bc1);
            L.adam[1] = (float)((double)dv.qv_lr[agent] * c / bc1);
            L.adam[2] = (float)(1e-8 * c);
        }
        __syncthreads();
        if (tid == 0 && !eps_in) dv.noise_ctr[agent] = nctr + 1;
        const float alpha_p = L.adam[0], alpha_v = L.adam[1];
        u.adam_eps = L.adam[2];

        // ================= 1: V'(s') =================
        u.S = S; u.H1 = L1C;
        u.trunk(tt + d.vW1, tt + d.vb1, L.x2);
        __syncthreads();
        u.fwd_gemm(acc, tt + d.vW2, L2C, L1C);
        u.template bias_relu<0>(acc, tt + d.vb2, L2C);
        u.template row_dot<false, 1>(acc, L2C, [&](int n, int) { return tt[d.vW3 + n]; }, L.part_q);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) L.vt[b] = u.template part_sum<1>(L.part_q, b, 0) + tt[d.vb3];
        // ================= 2: pi forward, the draw and its log-density (reversekl_network.py:332-357) =================
        u.H1 = L1A;
        u.trunk(th + d.pW1, th + d.pb1, L.x);
        for (int i = tid; i < NS * 256; i += kThreads) {      // row 0: mean head, row 1: log_std head
            const int j = i / 256, n = i % 256;
            L.wvec[i] = n < L2A ? (j == 0 ? th[d.pWm + n] : th[d.pWs + n]) : 0.0f;
        }
        __syncthreads();
        u.fwd_gemm(acc, th + d.pW2, L2A, L1A);
        u.template bias_relu<0>(acc, th + d.pb2, L2A);
        u.template row_dot<false, NS>(acc, L2A, [&](int n, int j) { return L.wvec[j * 256 + n]; }, L.part_h);
        u.template store_masks<0, true>(acc, L2A);
        f32x4 acch[MT][NTW];                                  // pi's hidden activation: needed again for the head gradients
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < NTW; i++) acch[mt][i] = acc[mt][i];
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float mu = u.template part_sum<NS>(L.part_h, b, 0) + th[d.pbm];
            const float lsr = u.template part_sum<NS>(L.part_h, b, 1) + th[d.pbs];
            const float ls = fminf(fmaxf(lsr, -20.0f), 2.0f);
            const float sd = expf(ls);
            const float z = mu + sd * L.eps[b];
            const float t = tanhf(z);
            const float dz = z - mu;
            const float lp = -(dz * dz) / (2.0f * (sd * sd)) - logf(sd) - LOG_SQRT_2PI - logf(1.0f - t * t + EPS);
            L.mu[b] = mu; L.lsr[b] = lsr; L.sd[b] = sd; L.z[b] = z; L.lp[b] = lp;
            L.xn[b * SMAX + S] = t * amax0;
            dv.tap_logp[(size_t)agent * RLC_MAX_BATCH + b] = lp;
        }
        // ================= 3: Q(s,a), its seeds and wave-local column reductions =================
        u.S = S + 1; u.H1 = L1C;
        u.trunk(th + d.qW1, th + d.qb1, L.xq);
        __syncthreads();
        f32x4 accq[MT][NTW];
        u.fwd_gemm(accq, th + d.qW2, L2C, L1C);
        u.template bias_relu<0>(accq, th + d.qb2, L2C);
        u.template row_dot<false, 1>(accq, L2C, [&](int n, int) { return th[d.qW3 + n]; }, L.part_q);
        u.template store_masks<1, false>(accq, L2C);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float q = u.template part_sum<1>(L.part_q, b, 0) + th[d.qb3];
            L.q[b] = q;
            dv.tap_q[(size_t)agent * RLC_MAX_BATCH + b] = q;
            L.dq[b] = 2.0f * (q - (L.r[b] + L.g[b] * L.vt[b])) * invB;      // MSELoss over the B x 1 outputs
        }
        __syncthreads();
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
                    const f32x4 d4 = *reinterpret_cast<const lds_f32x4*>(&L.dq[16 * mt + 4 * u.g]);
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
        // ================= 4: Q(s, a_new) =================
        u.trunk(th + d.qW1, th + d.qb1, L.xn);
        __syncthreads();
        u.fwd_gemm(acc, th + d.qW2, L2C, L1C);
        u.template bias_relu<0>(acc, th + d.qb2, L2C);
        u.template row_dot<false, 1>(acc, L2C, [&](int n, int) { return th[d.qW3 + n]; }, L.part_q);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float qn = u.template part_sum<1>(L.part_q, b, 0) + th[d.qb3];
            L.qn[b] = qn;
            dv.tap_qpi[(size_t)agent * RLC_MAX_BATCH + b] = qn;
        }
        // ================= 5: V(s) (values only; the V step repeats this forward for its masks) =================
        u.S = S;
        u.trunk(th + d.vW1, th + d.vb1, L.x);
        __syncthreads();
        u.fwd_gemm(acc, th + d.vW2, L2C, L1C);
        u.template bias_relu<0>(acc, th + d.vb2, L2C);
        u.template row_dot<false, 1>(acc, L2C, [&](int n, int) { return th[d.vW3 + n]; }, L.part_q);
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) {
            const float v = u.template part_sum<1>(L.part_q, b, 0) + th[d.vb3];
            L.v[b] = v;
            dv.tap_v[(size_t)agent * RLC_MAX_BATCH + b] = v;
        }
        __syncthreads();

        float pl = 0.0f;
        if (integral) {
            // ================= 6: Q at the quadrature nodes =================
            // z1s[b] = s_b W1[:S] + b1 (no relu), the action row of W1
            for (int e = tid; e < B * NQ; e += kThreads) {
                const int b = e / NQ, n0 = (e % NQ) << 2;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (n0 < L1C) {
                    o = *reinterpret_cast<const f32x4*>(&th[d.qb1 + n0]);
                    for (int i = 0; i < S; i++)
                        o += L.x[b * SMAX + i] * *reinterpret_cast<const f32x4*>(&th[d.qW1 + i * L1C + n0]);
                }
                if constexpr (ZG) *reinterpret_cast<f32x4*>(&z1g[b * LDH + n0]) = o;      // visible after the __syncthreads below
                else *reinterpret_cast<lds_f32x4*>(&L.z1s[b * LDH + n0]) = o;
                if constexpr (SPLIT) *reinterpret_cast<f32x4*>(&sp.zbuf[(size_t)rel_agent * MB * LDH + b * LDH + n0]) = o;
            }
            for (int n = tid; n < 256; n += kThreads) L.w1a[n] = n < L1C ? th[d.qW1 + S * L1C + n] : 0.0f;
            __syncthreads();
            if constexpr (SPLIT) { if (!kl_group_barrier(sp.bar + rel_agent, sp.C, bar_gen, sp.err, L.dups + 3)) return; }     // z1s published
            node_passes();
            if constexpr (SPLIT) { if (!kl_group_barrier(sp.bar + rel_agent, sp.C, bar_gen, sp.err, L.dups + 3)) return; }     // every pass has landed
            __syncthreads();
            // ================= 7: one wave per state: log pi at the nodes, d loss / d lp, seeds of mean and log_std =================
            for (int b = u.wave; b < B; b += kWaves) {
                const float mu = L.mu[b], sd = L.sd[b], var = sd * sd, lsd = logf(sd), vb = L.v[b];
                float shift = -INFINITY, zsum = 0.0f;
                if (dv.kl_kind == RLC_KL_FORWARD) {
                    for (int k = u.lane; k < K; k += 64) shift = fmaxf(shift, iq[b * K + k] / alpha_ent);
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) shift = fmaxf(shift, __shfl_xor(shift, off, 64));
                    for (int k = u.lane; k < K; k += 64) zsum += expf(iq[b * K + k] / alpha_ent - shift) * L.node_w[k];
                    zsum = kl_wave_sum(zsum);
                }
                float gm = 0.0f, gs = 0.0f, loss = 0.0f;
                for (int k = u.lane; k < K; k += 64) {
                    const float w = L.node_w[k], du = L.node_u[k] - mu;
                    const float lp = -(du * du) / (2.0f * var) - lsd - LOG_SQRT_2PI - L.node_j[k];
                    float coef;
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
                    gm += coef * (du / var);
                    gs += coef * (du * du / var - 1.0f);
                }
                gm = kl_wave_sum(gm); gs = kl_wave_sum(gs); loss = kl_wave_sum(loss);
                if (u.lane == 0) {
                    const bool inside = L.lsr[b] >= -20.0f && L.lsr[b] <= 2.0f;
                    L.dml[b * NS + 0] = gm * invB;
                    L.dml[b * NS + 1] = inside ? gs * invB : 0.0f;
                    L.pls[b] = loss;
                }
            }
        } else {
            // ll / hard_ll: -log_prob * (advantage).detach() on the drawn z
            for (int b = tid; b < B; b += kThreads) {
                const float adv = (L.qn[b] - L.v[b]) - (dv.kl_optim == RLC_KL_OPTIM_LL ? alpha_ent * L.lp[b] : 0.0f);
                const float coef = -adv * invB, dz = L.z[b] - L.mu[b], var = L.sd[b] * L.sd[b];
                const bool inside = L.lsr[b] >= -20.0f && L.lsr[b] <= 2.0f;
                L.pls[b] = -L.lp[b] * adv;
                L.dml[b * NS + 0] = coef * dz / var;
                L.dml[b * NS + 1] = inside ? coef * (dz * dz / var - 1.0f) : 0.0f;
            }
        }
        __syncthreads();
        for (int b = tid; b < B; b += kThreads) pl += L.pls[b];

        // ================= 8: pi step =================
        u.S = S; u.H1 = L1A;
        u.trunk(th + d.pW1, th + d.pb1, L.x);
        __syncthreads();
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
                        const float hv = acch[mt][i][r];
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
        u.template bwd_gemm<NS, 0>(acc, th + d.pW2, L2A, L1A, L.dml, L.wvec);
        __syncthreads();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_p, d.pW1, d.pb1, tapg, nullptr, 0.0f, L.x);
        u.template wgrad_adam<NS, 0, 0, false, true>(L.dml, nullptr, L2A, th + d.pW2, mm + d.pW2, vv + d.pW2, alpha_p,
                                                     tapg ? tapg + d.pW2 : nullptr, nullptr, 0.0f, L.wvec);
        {
            const int NT = (L2A + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2A && u.g <= NS) {
                    // lane group 0 -> pb2[n]; 1 -> Wm[n]; 2 -> Ws[n]
                    const int p = u.g == 0 ? d.pb2 + n : (u.g == 1 ? d.pWm + n : d.pWs + n);
                    const float gr = u.g == 0 ? g_pb2[i] : (u.g == 1 ? g_ph[i][0] : g_ph[i][1]);
                    u.adam_scalar_m(th, mm, vv, nullptr, tapg, p, gr, alpha_p, 0.0f);
                }
            }
            if (u.wave < NS) {            // bm / bs: sum_b of the seeds, wave j
                const int j = u.wave;
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dml[b * NS + j];
                gr = kl_wave_sum(gr);
                if (u.lane == 0) u.adam_scalar_m(th, mm, vv, nullptr, tapg, j == 0 ? d.pbm : d.pbs, gr, alpha_p, 0.0f);
            }
        }
        __syncthreads();
        // ================= 9: Q step =================
        u.S = S + 1; u.H1 = L1C;
        u.trunk(th + d.qW1, th + d.qb1, L.xq);
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < L2C ? th[d.qW3 + n] : 0.0f;
        __syncthreads();
        u.template bwd_gemm<1, 1>(acc, th + d.qW2, L2C, L1C, L.dq, L.wvec);
        __syncthreads();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_v, d.qW1, d.qb1, tapg, nullptr, 0.0f, L.xq);
        u.template wgrad_adam<1, 0, 1, false, true>(L.dq, nullptr, L2C, th + d.qW2, mm + d.qW2, vv + d.qW2, alpha_v,
                                                    tapg ? tapg + d.qW2 : nullptr, nullptr, 0.0f, L.wvec);
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2C && u.g < 2)
                    u.adam_scalar_m(th, mm, vv, nullptr, tapg, u.g == 0 ? d.qW3 + n : d.qb2 + n, u.g == 0 ? g_qw3[i] : g_qb2[i],
                                    alpha_v, 0.0f);
            }
            if (u.wave == 0) {
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dq[b];
                gr = kl_wave_sum(gr);
                if (u.lane == 0) u.adam_scalar_m(th, mm, vv, nullptr, tapg, d.qb3, gr, alpha_v, 0.0f);
            }
        }
        __syncthreads();
        // ================= 10: V forward again (masks, accumulators), V step, update_target_network =================
        u.S = S;
        u.trunk(th + d.vW1, th + d.vb1, L.x);
        for (int n = tid; n < 256; n += kThreads) L.wvec[n] = n < L2C ? th[d.vW3 + n] : 0.0f;
        __syncthreads();
        u.fwd_gemm(acc, th + d.vW2, L2C, L1C);
        u.template bias_relu<0>(acc, th + d.vb2, L2C);
        u.template store_masks<0, true>(acc, L2C);
        float ql = 0.0f, vl = 0.0f;
        for (int b = tid; b < B; b += kThreads) {
            const float tq = L.r[b] + L.g[b] * L.vt[b];
            const float tv = dv.kl_qupdate == RLC_KL_Q_SAC ? L.qn[b] - alpha_ent * L.lp[b]
                                                           : (L.r[b] - alpha_ent * L.lp[b]) + L.g[b] * L.vt[b];
            const float eq = L.q[b] - tq, ev = L.v[b] - tv;
            ql += eq * eq; vl += ev * ev;
            L.dvs[b] = 2.0f * ev * invB;
        }
        ql = kl_blk_sum(ql, L.red); vl = kl_blk_sum(vl, L.red); pl = kl_blk_sum(pl, L.red);
        if (tid == 0) {
            dv.tap_loss[agent * 4 + 0] = pl * invB;
            dv.tap_loss[agent * 4 + 1] = ql * invB;
            dv.tap_loss[agent * 4 + 2] = vl * invB;
        }
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
        u.template bwd_gemm<1, 0>(acc, th + d.vW2, L2C, L1C, L.dvs, L.wvec);
        __syncthreads();
        u.trunk_grad_adam(acc, th, mm, vv, alpha_v, d.vW1, d.vb1, tapg, tt, tau, L.x);
        u.template wgrad_adam<1, 0, 0>(L.dvs, nullptr, L2C, th + d.vW2, mm + d.vW2, vv + d.vW2, alpha_v,
                                       tapg ? tapg + d.vW2 : nullptr, tt + d.vW2, tau, L.wvec);
        {
            const int NT = (L2C + 15) >> 4;
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const int t = u.tile_of(i);
                const int n = 16 * t + u.c;
                if (t < NT && n < L2C && u.g < 2)
                    u.adam_scalar_m(th, mm, vv, tt, tapg, u.g == 0 ? d.vW3 + n : d.vb2 + n, u.g == 0 ? g_vw3[i] : g_vb2[i],
                                    alpha_v, tau);
            }
            if (u.wave == 0) {
                float gr = 0.0f;
                for (int b = u.lane; b < MB; b += 64) gr += L.dvs[b];
                gr = kl_wave_sum(gr);
                if (u.lane == 0) u.adam_scalar_m(th, mm, vv, tt, tapg, d.vb3, gr, alpha_v, tau);
            }
        }
        __syncthreads();
        if (tid == 0) dv.kl_step[agent] = step;
        __syncthreads();
    }
}

// workgroups of the latency-mode grid: agents padded to the 8 XCDs, C per agent
inline int kl_split_grid(int n_agents, int C) { return (n_agents + 7) / 8 * 8 * C; }

template <int MT, int MTQ>
int kl_launch_split_t(const RlcSacDev& dv, const KlSplit& sp, int first_agent, int n_updates, int source,
                      const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st) {
    const size_t lds = ksmem_carve(dv.d, MT, MTQ, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA KL kernel needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = rlc_kl_update_mfma_kernel<MT, MTQ, true>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    RLC_HIP(hipMemsetAsync(sp.bar, 0, sizeof(unsigned int) * sp.n_agents, st));
    hipLaunchKernelGGL(kern, dim3(kl_split_grid(sp.n_agents, sp.C)), dim3(kThreads), lds, st, dv, first_agent, n_updates,
                       source, idx_dev, eps_dev, grad_taps, (const RlcSacRollout*)nullptr, sp);
    RLC_HIP(hipGetLastError());
    return 0;
}

template <int MT, int MTQ>
int kl_launch_t(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source, const long long* idx_dev,
                const float* eps_dev, int grad_taps, hipStream_t st, const RlcSacRollout* rollout) {
    const size_t lds = ksmem_carve(dv.d, MT, MTQ, nullptr, nullptr);
    RLC_REQUIRE(lds <= 160 * 1024, "MFMA KL kernel needs %zu B of LDS (> 160 KiB)", lds);
    auto kern = rlc_kl_update_mfma_kernel<MT, MTQ, false>;
    static bool attr_set = false;
    if (!attr_set) {
        RLC_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    RLC_REQUIRE(!(rollout && eps_dev), "the on-device loop draws its own eps");
    hipLaunchKernelGGL(kern, dim3(n_agents), dim3(kThreads), lds, st, dv, first_agent, n_updates, source, idx_dev, eps_dev,
                       grad_taps, rollout, KlSplit{nullptr, nullptr, nullptr, 1, n_agents});
    RLC_HIP(hipGetLastError());
    return 0;
}

}  // namespace
