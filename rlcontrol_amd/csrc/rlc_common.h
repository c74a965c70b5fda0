// rlc_common.h -- shared host/device declarations of librlcontrol_hip.so (gfx950 only).
// Product code: never includes or links anything from oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/rlcontrol_hip.h"

#define RLC_MAX_BATCH 128      // minibatch rows per update (LDS-resident per-sample vectors)
#define RLC_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing: every ABI entry returns 0 / non-zero and leaves a message for rlc_last_error()
// ---------------------------------------------------------------------------------------------
void rlc_set_error(const char* fmt, ...);

#define RLC_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            rlc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

#define RLC_REQUIRE(cond, ...)        \
    do {                              \
        if (!(cond)) {                \
            rlc_set_error(__VA_ARGS__); \
            return 2;                 \
        }                             \
    } while (0)

// ---------------------------------------------------------------------------------------------
// network geometry.  Blob layout = variable creation order of hydra_ddpg_network.py:100-140.
// ---------------------------------------------------------------------------------------------
// The two big matrices (Wa2 [H1,HA], Wc2 [H1+A,HC]) and everything shaped like them (target copy, Adam m/v,
// gradient taps) can be kept in a TILE-BLOCKED device layout: 16x16 blocks, block (tr, tc) at
// (tr*ceil(cols/16) + tc)*256 floats, and inside a block element (row c, col 4g+r) at ((g*16 + c)*4 + r) --
// i.e. lane l = g*16+c of a wavefront owns the 16 bytes at l*16.  The MFMA kernel's weight-gradient epilogue
// (4 arrays read + written per element) and its backward GEMM then move 1 KB CONTIGUOUS per instruction instead
// of sixteen 64-byte row fragments 800 B apart: 2.7x the per-CU streaming rate (scripts/micro/stream_pattern.hip).
// The layout is private to the library: the ABI blob is row-major and (un)packed on the host.
__host__ __device__ inline int rlc_blk_index(int row, int col, int ncols) {
    const int ntc = (ncols + 15) >> 4;
    const int c = row & 15, cc = col & 15;
    return (((row >> 4) * ntc + (col >> 4)) << 8) + ((((cc >> 2) << 4) + c) << 2) + (cc & 3);
}
__host__ __device__ inline int rlc_blk_floats(int rows, int cols) { return (((rows + 15) >> 4) * ((cols + 15) >> 4)) << 8; }
__host__ __device__ inline size_t rlc_widx(int blocked, int row, int col, int ncols) {
    return blocked ? (size_t)rlc_blk_index(row, col, ncols) : (size_t)row * ncols + col;
}

#define RLC_DDPG_MAX_SEG 20
struct RlcDims {
    int S, A, H1, HA, HC, B;
    int blocked;   // 1: Wa2 / Wc2 segments use the tile-blocked layout (MFMA kernel), 0: row-major (generic kernel)
    int norm;      // 1: norm_type 'layer' -- tf.contrib layer_norm (beta, gamma) after every hidden fully_connected
                   //    (agents/network/base_network.py:53-56); 0: 'none' / 'input_norm' (just the activation)
    int sep;       // 1: separate actor / critic networks (agents/network/actor_network.py:73-96,
                   //    critic_network.py:77-99): the critic has a first layer of its own; 0: the hydra network
    int arow0;     // device row of Wc2's first action row: H1 (row-major) or the next multiple of 16 (blocked), so
                   // that in the blocked layout the trunk rows H1..arow0-1 are zero padding and the k-loops of the
                   // forward GEMMs need no row mask
    // DEVICE offsets of the tensors inside one agent's blob, variable creation order
    //   W1 b1 [l1b l1g] Wa2 ba2 [l2b l2g] Wa3 ba3 | [Wc1 bc1 [lcb lcg]] Wc2 bc2 [l3b l3g] Wc3 bc3
    // ([l..] with layer norm, [Wc1 ..] with separate networks; without them oWc1.. alias oW1..): each tensor starts
    // on a 256-byte boundary so that rows can be fetched with 16-byte vector loads.  The ABI's compact blob
    // (rlc_ddpg_set_blob / get_blob) is packed/unpacked on the host with rlc_pack_blob / rlc_unpack_blob.
    int oW1, ob1, oWa2, oba2, oWa3, oba3, oWc2, obc2, oWc3, obc3;
    int oL1b, oL1g, oL2b, oL2g, oL3b, oL3g, oWc1, obc1, oLcb, oLcg;
    int ocritic0;  // device offset of the critic optimizer's own block (everything before it is the actor optimizer's)
    int iWa2, iWc2;             // segment indices of the two matrices that can be tile-blocked
    int P;      // parameter count of the compact ABI blob
    int Pdev;   // extent of the device layout in use
    int Ppad;   // per-agent stride of every device blob (fits either layout, multiple of 64 floats)
    int nseg;
    int seg_len[RLC_DDPG_MAX_SEG], seg_compact[RLC_DDPG_MAX_SEG], seg_dev[RLC_DDPG_MAX_SEG];
    int seg_rows[RLC_DDPG_MAX_SEG], seg_cols[RLC_DDPG_MAX_SEG], seg_h[RLC_DDPG_MAX_SEG];
    char seg_big[RLC_DDPG_MAX_SEG];
};

template <class D> inline void rlc_layout_segs(D& d);
template <class D> inline void rlc_pack_segs(const D& d, const float* compact, float* padded);
template <class D> inline void rlc_unpack_segs(const D& d, const float* padded, float* compact);

inline RlcDims rlc_make_dims(int S, int A, int H1, int HA, int HC, int B, int blocked, int norm = 0, int sep = 0);

// logical row -> device row of segment i (only Wc2's action rows move, see RlcDims::arow0)
__host__ __device__ inline int rlc_dev_row(const RlcDims& d, int seg, int r) {
    return (seg == d.iWc2 && r >= d.H1) ? d.arow0 + (r - d.H1) : r;
}

// Generic form of the two functions above for dims structs that carry a segment table (SAC, NAF): fields nseg,
// blocked, seg_len / seg_compact / seg_dev / seg_rows / seg_cols, seg_big (1: a matrix that is tile-blocked when
// d.blocked) and seg_h (rows fed by the hidden activation; rows seg_h.. are extra input rows -- a critic's action
// rows -- which the blocked layout moves to a block row of their own, as RlcDims::arow0 does).
template <class D>
inline int rlc_seg_dev_row(const D& d, int i, int r) {
    return (d.blocked && d.seg_big[i] && r >= d.seg_h[i]) ? ((d.seg_h[i] + 15) & ~15) + (r - d.seg_h[i]) : r;
}
template <class D>
inline void rlc_pack_segs(const D& d, const float* compact, float* padded) {
    for (int i = 0; i < d.nseg; i++) {
        const float* src = compact + d.seg_compact[i];
        float* dst = padded + d.seg_dev[i];
        if (d.blocked && d.seg_big[i]) {
            for (int r = 0; r < d.seg_rows[i]; r++)
                for (int c = 0; c < d.seg_cols[i]; c++)
                    dst[rlc_blk_index(rlc_seg_dev_row(d, i, r), c, d.seg_cols[i])] = src[(size_t)r * d.seg_cols[i] + c];
        } else {
            for (int k = 0; k < d.seg_len[i]; k++) dst[k] = src[k];
        }
    }
}
template <class D>
inline void rlc_unpack_segs(const D& d, const float* padded, float* compact) {
    for (int i = 0; i < d.nseg; i++) {
        float* dst = compact + d.seg_compact[i];
        const float* src = padded + d.seg_dev[i];
        if (d.blocked && d.seg_big[i]) {
            for (int r = 0; r < d.seg_rows[i]; r++)
                for (int c = 0; c < d.seg_cols[i]; c++)
                    dst[(size_t)r * d.seg_cols[i] + c] = src[rlc_blk_index(rlc_seg_dev_row(d, i, r), c, d.seg_cols[i])];
        } else {
            for (int k = 0; k < d.seg_len[i]; k++) dst[k] = src[k];
        }
    }
}
// lay the segments out: row-major segments padded to 64 floats; big ones to their block extent when blocked.
// Ppad fits either layout, so a handle can switch kernels (and layouts) without reallocating.
template <class D>
inline void rlc_layout_segs(D& d) {
    int pc = 0, pd = 0, pmax = 0;
    for (int i = 0; i < d.nseg; i++) {
        const int len = d.seg_rows[i] * d.seg_cols[i];
        const int xr = d.seg_rows[i] - d.seg_h[i];
        const int blk = d.seg_big[i] ? rlc_blk_floats(xr > 0 ? ((d.seg_h[i] + 15) & ~15) + xr : d.seg_rows[i], d.seg_cols[i]) : len;
        d.seg_len[i] = len;
        d.seg_compact[i] = pc;
        d.seg_dev[i] = pd;
        pc += len;
        pd += (((d.seg_big[i] && d.blocked) ? blk : len) + 63) & ~63;
        pmax += ((blk > len ? blk : len) + 63) & ~63;
    }
    d.P = pc; d.Pdev = pd; d.Ppad = pmax;
}

inline RlcDims rlc_make_dims(int S, int A, int H1, int HA, int HC, int B, int blocked, int norm, int sep) {
    RlcDims d;
    d.S = S; d.A = A; d.H1 = H1; d.HA = HA; d.HC = HC; d.B = B;
    d.blocked = blocked; d.norm = norm; d.sep = sep;
    d.arow0 = blocked ? ((H1 + 15) & ~15) : H1;
    int n = 0;
    auto seg = [&](int r, int c, int big, int h) {
        d.seg_rows[n] = r; d.seg_cols[n] = c; d.seg_big[n] = (char)big; d.seg_h[n] = h;
        return n++;
    };
    auto vec = [&](int c) { return seg(1, c, 0, 1); };
    const int iW1 = seg(S, H1, 0, S), ib1 = vec(H1);
    const int iL1b = norm ? vec(H1) : -1, iL1g = norm ? vec(H1) : -1;
    d.iWa2 = seg(H1, HA, 1, H1);
    const int iba2 = vec(HA);
    const int iL2b = norm ? vec(HA) : -1, iL2g = norm ? vec(HA) : -1;
    const int iWa3 = seg(HA, A, 0, HA), iba3 = vec(A);
    const int icrit = n;
    const int iWc1 = sep ? seg(S, H1, 0, S) : iW1, ibc1 = sep ? vec(H1) : ib1;
    const int iLcb = (sep && norm) ? vec(H1) : iL1b, iLcg = (sep && norm) ? vec(H1) : iL1g;
    d.iWc2 = seg(H1 + A, HC, 1, H1);
    const int ibc2 = vec(HC);
    const int iL3b = norm ? vec(HC) : -1, iL3g = norm ? vec(HC) : -1;
    const int iWc3 = seg(HC, 1, 0, HC), ibc3 = vec(1);
    d.nseg = n;
    rlc_layout_segs(d);
    auto at = [&](int i) { return i >= 0 ? d.seg_dev[i] : 0; };
    d.oW1 = at(iW1); d.ob1 = at(ib1); d.oWa2 = at(d.iWa2); d.oba2 = at(iba2); d.oWa3 = at(iWa3); d.oba3 = at(iba3);
    d.oWc2 = at(d.iWc2); d.obc2 = at(ibc2); d.oWc3 = at(iWc3); d.obc3 = at(ibc3);
    d.oL1b = at(iL1b); d.oL1g = at(iL1g); d.oL2b = at(iL2b); d.oL2g = at(iL2g); d.oL3b = at(iL3b); d.oL3g = at(iL3g);
    d.oWc1 = at(iWc1); d.obc1 = at(ibc1); d.oLcb = at(iLcb); d.oLcg = at(iLcg);
    d.ocritic0 = d.seg_dev[icrit];
    return d;
}

// compact row-major ABI blob <-> one agent's device blob (padded is Ppad floats, zero-filled by the caller)
inline void rlc_pack_blob(const RlcDims& d, const float* compact, float* padded) { rlc_pack_segs(d, compact, padded); }
inline void rlc_unpack_blob(const RlcDims& d, const float* padded, float* compact) { rlc_unpack_segs(d, padded, compact); }

// Replay ring of ONE agent lives at agent*cap inside each SoA array.  Logical index 0 = oldest.
struct RlcRingMeta {
    long long start;   // physical slot of the oldest transition
    long long size;    // number stored (<= cap)
};

// Device view of the replay of a population (any algorithm): SoA ring per agent + a staging minibatch.
struct RlcReplayDev {
    int S, A, n_agents;
    long long cap;               // capacity per agent
    float* rs; float* ra; double* rr; float* rs2; double* rg;     // [n_agents][cap][*]
    RlcRingMeta* ring;           // [n_agents]
    float* gs; float* ga; double* gr; float* gs2; double* gg;     // staging minibatch [n_agents][RLC_MAX_BATCH][*]
    const unsigned long long* seed;      // [n_agents] Philox keys
    unsigned long long* sample_ctr;      // [n_agents] sampler invocations so far
};

// On-device experiment loop (rollout_kernels.hip): environment state, episode bookkeeping and the logs
// Experiment.run() returns (experiment.py:52-98 of the reference), per agent.
#define RLC_ENV_PENDULUM RLC_ENV_PENDULUM_V0
#define RLC_ENV_STATE 4                  // doubles of simulator state per environment instance
struct RlcEnvDev {
    int env_id;
    int episode_limit;                   // EPISODE_STEPS_LIMIT
    int learn_threshold;                 // max(warmup_steps, batch_size): learn when size > threshold
    int eval_episodes;
    int max_episodes, max_evals;         // log capacities per agent
    double gamma;
    double* sim;                         // [n_agents][RLC_ENV_STATE] training simulator state
    double* obs;                         // [n_agents][S] current observation of the training episode
    int* ep_step;                        // [n_agents]
    double* ep_ret;                      // [n_agents]
    int* need_reset;                     // [n_agents] 1 = next train step starts a new episode
    long long* total_steps;              // [n_agents]
    unsigned long long* reset_ctr;       // [n_agents] training-environment resets so far
    double* train_ret; int* train_len; long long* train_cum; int* n_train_ep;   // [n_agents][max_episodes], [n_agents]
    double* eval_ret; int* eval_len;     // [n_agents][max_evals][eval_episodes]
};

// Everything the DDPG kernels need; passed by value as a kernel argument (all pointers are device memory).
struct RlcDev {
    RlcDims d;
    RlcReplayDev rep;
    int n_agents;
    int clip_state;
    float tau;
    float ou_theta, ou_mu, ou_sigma;
    // networks + optimizer state: [n_agents][Ppad]
    float *theta, *theta_t, *m_a, *v_a, *m_c, *v_c;
    float* pw;                   // [n_agents][4] beta powers {a1,a2,c1,c2}
    const float *actor_lr, *critic_lr;   // [n_agents]
    const float *smin, *smax, *amin, *amax;
    unsigned long long* noise_ctr;       // [n_agents] OU draws so far
    float* ou_state;                     // [n_agents][A]
    // taps of the last update: [n_agents][B], [B], [B*A], [B*A]; grads [n_agents][Ppad] (optional)
    float *tap_q, *tap_y, *tap_aout, *tap_dqda, *tap_gc, *tap_ga;
    // generic-kernel scratch: [n_agents][scratch_stride] floats
    float* scratch;
    long long scratch_stride;
};

// device-resident argument block of a launch that runs training steps (ddpg_rollout_device.h)
struct RlcRollout {
    RlcDev dv;
    RlcEnvDev env;
};

// where a launch takes its minibatch from
enum RlcBatchSource { RLC_SRC_REPLAY_DEVICE_SAMPLER = 0, RLC_SRC_REPLAY_HOST_INDICES = 1, RLC_SRC_STAGING = 2 };

// ---------------------------------------------------------------------------------------------
// host launchers implemented in the kernel translation units
// ---------------------------------------------------------------------------------------------
// generic (any dims) fused update: one workgroup per agent, n_updates sequential updates per launch
// `rollout` (device pointer, may be null): every iteration first takes one environment step of the on-device
// experiment loop and skips the update while the learn gate is closed; q8_first flags the first iteration
// as following an evaluation (OU reset after acting).
int rlc_launch_ddpg_update_generic(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                   const long long* idx_dev, int grad_taps, hipStream_t st,
                                   const RlcRollout* rollout = nullptr, int q8_first = 0);
// MFMA-tiled fused update (dims must satisfy rlc_mfma_supported)
bool rlc_mfma_supported(const RlcDims& d);
// batch-split latency mode (ddpg_split.hip): C workgroups per agent; part [n_agents][C][Ppad] zero-initialised, bar
// [n_agents], err [1]; rlc_split_mt: M tiles per workgroup for (batch, C), 0 if unsupported
int rlc_split_mt(int B, int C);
int rlc_launch_ddpg_update_split(const RlcDev& dv, float* part, unsigned int* bar, int* err, int C, int first_agent,
                                 int n_agents, int n_updates, int source, const long long* idx_dev, int grad_taps,
                                 hipStream_t st);
int rlc_launch_ddpg_update_mfma(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                const long long* idx_dev, int grad_taps, hipStream_t st,
                                const RlcRollout* rollout = nullptr, int q8_first = 0);

// acting / evaluation
int rlc_launch_act(const RlcDev& dv, int first_agent, int n, const float* states_dev, float* out_dev, int explore,
                   hipStream_t st, int* done_flag = nullptr, int done_val = 0);
int rlc_launch_qval(const RlcDev& dv, int agent, int n, const float* states_dev, const float* actions_dev,
                    float* out_dev, hipStream_t st);
int rlc_launch_reset_noise(const RlcDev& dv, int first_agent, int n, hipStream_t st);
// on-device experiment loop (rollout_kernels.hip)
int rlc_launch_ddpg_eval(const RlcDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st);
size_t rlc_generic_scratch_floats(const RlcDims& d);
// replay
int rlc_launch_replay_scatter(const RlcReplayDev& rp, int agent, long long first_slot, long long n, const float* s,
                              const float* a, const double* r, const float* s2, const double* g, hipStream_t st);
int rlc_launch_replay_fill_all(const RlcReplayDev& rp, long long n, const float* s, const float* a, const double* r,
                               const float* s2, const double* g, hipStream_t st);
int rlc_launch_replay_gather(const RlcReplayDev& rp, int agent, const long long* logical_idx_dev, int k, float* s,
                             float* a, double* r, float* s2, double* g, hipStream_t st);
int rlc_launch_sample_indices(const RlcReplayDev& rp, int agent, int k, long long* out_idx_dev, hipStream_t st);
// single transition passed by value (no staging copy): ReplayBuffer.add for one env step
#define RLC_PUT1_MAX_FLOATS 56
struct RlcPut1 {
    float sas[RLC_PUT1_MAX_FLOATS];   // s[S] | s2[S] | a[A]
    double r, g;
    long long slot, new_start, new_size;
};
int rlc_launch_replay_put1(const RlcReplayDev& rp, int agent, const RlcPut1& t, hipStream_t st);
int rlc_launch_set_ring(const RlcReplayDev& rp, int agent, long long start, long long size, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// device-side helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__

// Philox4x32-10 (Salmon et al. 2011): counter-based, one independent stream per (key, counter).
struct Philox4 { unsigned int x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long key, unsigned long long ctr_lo,
                                                 unsigned long long ctr_hi) {
    unsigned int k0 = (unsigned int)key, k1 = (unsigned int)(key >> 32);
    unsigned int c0 = (unsigned int)ctr_lo, c1 = (unsigned int)(ctr_lo >> 32);
    unsigned int c2 = (unsigned int)ctr_hi, c3 = (unsigned int)(ctr_hi >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

// unbiased-enough integer in [0,n): high 64 bits of (64 random bits x n); bias < n / 2^64
__device__ __forceinline__ long long philox_below(const Philox4& p, long long n) {
    const unsigned long long u = ((unsigned long long)p.x << 32) | p.y;
    return (long long)__umul64hi(u, (unsigned long long)n);
}

// two standard normals from one Philox draw (Box-Muller on (0,1] uniforms)
__device__ __forceinline__ void philox_normal2(const Philox4& p, float& n0, float& n1) {
    const float u0 = ((float)(p.x >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    const float u1 = (float)(p.y >> 8) * (1.0f / 16777216.0f);            // [0,1)
    const float rad = sqrtf(-2.0f * logf(u0));
    n0 = rad * cosf(6.28318530717958647692f * u1);
    n1 = rad * sinf(6.28318530717958647692f * u1);
}

// logical (0 = oldest) -> physical slot of an agent's ring
__device__ __forceinline__ long long ring_slot(const RlcRingMeta& m, long long cap, long long logical) {
    long long p = m.start + logical;
    return p >= cap ? p - cap : p;
}

__device__ __forceinline__ float clip_state_val(float v, int do_clip, float lo, float hi) {
    // hydra_ddpg_network.py:86-87 with RunningMeanStd mean 0 / var 1 baked in (quirk Q6)
    return do_clip ? fminf(fmaxf((v - 0.0f) / 1.0f, lo), hi) : v;
}

// TF-1.15 ApplyAdam on one element (core/kernels/training_ops.cc, non-Nesterov; quirk Q2)
// Every product and sum is rounded (and, with denormals flushed, flushed) on its own, as in TF's Eigen expression
// m += (g - m) * (1 - beta1): contracted into an fma the decrement of an idle slot would never flush and m would decay
// to zero, whereas the reference's checkpoints show idle slots resting at 9..10 x FLT_MIN (tests/test_ckpt_pins.py).
__device__ __forceinline__ float adam_step(float var, float g, float& m, float& v, float alpha) {
#pragma clang fp contract(off)
    m += (g - m) * (1.0f - 0.9f);
    v += (g * g - v) * (1.0f - 0.999f);
    return var - (m * alpha) / (sqrtf(v) + 1e-8f);
}

// the same step with the epsilon as an argument (generic kernels: AdamCtx::eps)
__device__ __forceinline__ float adam_step_eps(float var, float g, float& m, float& v, float alpha, float eps) {
#pragma clang fp contract(off)
    m += (g - m) * (1.0f - 0.9f);
    v += (g * g - v) * (1.0f - 0.999f);
    return var - (m * alpha) / (sqrtf(v) + eps);
}

// same update with the hardware sqrt / reciprocal (1 ulp each) instead of the IEEE-exact expansions:
// ~10 VALU instead of ~30 per element; relative deviation ~2e-7, far inside the 1e-5 parity bar
__device__ __forceinline__ float adam_step_fast(float var, float g, float& m, float& v, float alpha) {
#pragma clang fp contract(off)
    m += (g - m) * (1.0f - 0.9f);
    v += (g * g - v) * (1.0f - 0.999f);
    return var - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + 1e-8f);
}

__device__ __forceinline__ float adam_step_fast_eps(float var, float g, float& m, float& v, float alpha, float eps) {
#pragma clang fp contract(off)
    m += (g - m) * (1.0f - 0.9f);
    v += (g * g - v) * (1.0f - 0.999f);
    return var - (m * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
}

__device__ __forceinline__ float adam_alpha(float lr, float b1p, float b2p) {
    return lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
}

// column n of a [rows, ncols] weight matrix in either device layout (row-major or tile-blocked, rlc_common.h):
// at(k0) points at row k0; rows k0+i of the same 16-row block follow at i*step floats.
struct RlcWCol {
    const float* p;
    int step, rowblk, blocked;
    __device__ __forceinline__ const float* at(int k0) const {
        return blocked ? p + (size_t)(k0 >> 4) * rowblk + ((k0 & 15) << 2) : p + (size_t)k0 * step;
    }
};
__device__ __forceinline__ RlcWCol rlc_wcol(const float* W, int blocked, int n, int ncols) {
    RlcWCol w;
    w.blocked = blocked;
    if (blocked) {
        w.p = W + ((n >> 4) << 8) + (((n & 15) >> 2) << 6) + (n & 3);
        w.step = 4;
        w.rowblk = ((ncols + 15) >> 4) << 8;
    } else {
        w.p = W + n;
        w.step = ncols;
        w.rowblk = 0;
    }
    return w;
}

// One row z[0..N) (LDS or global) -> relu(tf.contrib layer_norm(z)) in place, by the whole workgroup (B = 1 acting
// paths).  red: >= 34 floats of LDS.  Biased variance over the features, eps 1e-12 (base_network.py:53-56).
__device__ inline void rlc_row_layernorm_relu(float* z, int N, const float* beta, const float* gamma, float* red) {
    const int tid = threadIdx.x, nthr = blockDim.x, nw = (nthr + 63) >> 6;
    auto block_sum = [&](float v) {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        float t = 0.0f;
        for (int w = 0; w < nw; w++) t += red[w];
        return t;
    };
    float s = 0.0f;
    for (int n = tid; n < N; n += nthr) s += z[n];
    const float mean = block_sum(s) / (float)N;
    float q = 0.0f;
    for (int n = tid; n < N; n += nthr) { const float c = z[n] - mean; q += c * c; }
    const float rs = 1.0f / sqrtf(block_sum(q) / (float)N + 1e-12f);
    for (int n = tid; n < N; n += nthr) z[n] = fmaxf((z[n] - mean) * rs * gamma[n] + beta[n], 0.0f);
    __syncthreads();
}

// h_out[n] = relu(bias[n] + sum_k h_in[k] W[k][n]) for one row (B = 1 acting paths), W in either device layout.
// One thread per output unit, k ascending in one accumulator; the weight column is fetched KC rows at a time so
// that KC loads are in flight per thread instead of one dependent chain.  No barrier inside.
__device__ inline void rlc_hidden_forward_row(const float* W, int blocked, const float* bias, const float* h_in, int K,
                                              int N, float* h_out, bool relu = true) {
    constexpr int KC = 16;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const RlcWCol wcol = rlc_wcol(W, blocked, n, N);
        float acc = 0.0f;
        int k0 = 0;
        for (; k0 + KC <= K; k0 += KC) {     // KC = 16: a chunk never straddles a 16-row block
            const float* wp = wcol.at(k0);
            float w[KC];
#pragma unroll
            for (int i = 0; i < KC; i++) w[i] = wp[(size_t)i * wcol.step];
#pragma unroll
            for (int i = 0; i < KC; i++) acc += h_in[k0 + i] * w[i];
        }
        for (; k0 < K; k0++) acc += h_in[k0] * *wcol.at(k0);
        h_out[n] = relu ? fmaxf(acc + bias[n], 0.0f) : acc + bias[n];
    }
}

// k distinct uniform logical indices in [0, n) for one agent; one workgroup.
// dense regime (3k >= n, as in sample_n_k): partial Fisher-Yates over an LDS copy of range(n);
// sparse regime: one candidate per thread, duplicates (against lower-numbered threads) redrawn
// until none remain -- equivalent in distribution to sequential sampling without replacement.
// Every thread of the workgroup must call it; blockDim.x must be a multiple of RLC_MAX_BATCH (128).
// (PI / PL: int / long long pointers into LDS, generic or address-space-3 typed)
template <class PI, class PL>
__device__ inline void rlc_sample_distinct(long long n, int k, unsigned long long key, unsigned long long call,
                                           PI lds_pool /* >= 3*RLC_MAX_BATCH ints */, PL out /* LDS, k */,
                                           PI lds_dups /* 1 int */) {
    const int tid = threadIdx.x;
    if (3LL * k >= n) {
        for (int i = tid; i < (int)n; i += blockDim.x) lds_pool[i] = i;
        __syncthreads();
        if (tid == 0) {
            for (int i = 0; i < k; i++) {
                const Philox4 p = philox4x32_10(key, call, 0x100000000ull + (unsigned long long)i);
                const int j = i + (int)philox_below(p, n - i);
                const int t = lds_pool[i]; lds_pool[i] = lds_pool[j]; lds_pool[j] = t;
            }
        }
        __syncthreads();
        for (int i = tid; i < k; i += blockDim.x) out[i] = lds_pool[i];
        __syncthreads();
        return;
    }
    // sparse regime.  Candidate t (t < k) is thread t's; the duplicate test "some j < t holds the same value"
    // is spread over the whole workgroup: thread (t + 128*p) scans j in [32p', ...) -- no data-dependent
    // exits, broadcast LDS reads -- and the verdicts are combined through one LDS word.
    long long mine = -1;
    unsigned int round = 0;
    bool need = tid < k;
    PI flag = lds_pool;                       // k ints (the dense regime's pool is free here)
    const int t = tid & (RLC_MAX_BATCH - 1);
    const int part = tid / RLC_MAX_BATCH, nparts = (blockDim.x + RLC_MAX_BATCH - 1) / RLC_MAX_BATCH;
    const int span = (RLC_MAX_BATCH + nparts - 1) / nparts;
    for (;;) {
        if (need) {
            const Philox4 p = philox4x32_10(key, call, ((unsigned long long)round << 32) | (unsigned int)tid);
            mine = philox_below(p, n);
            out[tid] = mine;
        }
        if (tid < k) flag[tid] = 0;
        if (tid == 0) *lds_dups = 0;
        __syncthreads();
        if (t < k) {
            const long long v = out[t];
            const int j0 = part * span, j1 = min(t, j0 + span);
            bool dup = false;
            for (int j = j0; j < j1; j++) dup |= out[j] == v;
            if (dup) { flag[t] = 1; *lds_dups = 1; }      // benign races: every writer stores 1
        }
        __syncthreads();
        need = tid < k && flag[tid] != 0;
        const int dups = *lds_dups;
        __syncthreads();
        if (dups == 0) break;
        round++;
    }
}


#endif  // __HIPCC__
