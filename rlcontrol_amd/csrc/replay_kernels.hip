// replay_kernels.hip -- device-resident replay ring of (s, a, r, s', gamma_i) in SoA layout.
//
// Replaces (reference file:line):
//   utils/replaybuffer.py:25-27  ReplayBuffer.add  -> rlc_replay_scatter_kernel (ring write)
//   utils/replaybuffer.py:32-37  sample_batch's per-item __getitem__ + 5 np.array conversions
//                                (utils/custom_collections.py:37-58,103-104) -> rlc_replay_gather_kernel
//   utils/custom_collections.py:107-131  sample_n_k -> rlc_sample_indices_kernel (Philox; statistical
//                                parity -- exact-sequence parity is the host sampler's job)
// The replay stores the per-transition gamma (gamma or 0.0 at terminals, agents/base_agent.py:56-59),
// not a done flag.  r and gamma stay float64 so that the TD target can be formed in float64 exactly
// as agents/DDPG.py:80-84 does (quirk Q5); s, a, s' are fp32 = what the fp32 placeholders receive.
//
// HBM layout: per field one [n_agents][cap][width] array; logical index 0 is the oldest transition,
// physical slot = (start + logical) mod cap.  A transition is 2S+A floats + 2 doubles (44 B at
// Pendulum sizes); a random gather touches one 64-B sector per field per sample.
#include "rlc_common.h"

__global__ void rlc_replay_scatter_kernel(RlcReplayDev rp, int agent, long long first_slot, long long n,
                                          const float* s, const float* a, const double* r, const float* s2,
                                          const double* g) {
    const int S = rp.S, A = rp.A;
    const long long cap = rp.cap;
    const long long base = (long long)agent * cap;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long slot = first_slot + i;
        slot %= cap;
        for (int j = 0; j < S; j++) {
            rp.rs[(base + slot) * S + j] = s[i * S + j];
            rp.rs2[(base + slot) * S + j] = s2[i * S + j];
        }
        for (int j = 0; j < A; j++) rp.ra[(base + slot) * A + j] = a[i * A + j];
        rp.rr[base + slot] = r[i];
        rp.rg[base + slot] = g[i];
    }
}

// every agent's ring <- the same n transitions (bench set-up; coalesced field-wise copies)
__global__ void rlc_replay_fill_all_kernel(RlcReplayDev rp, long long n, const float* s, const float* a,
                                           const double* r, const float* s2, const double* g) {
    const int S = rp.S, A = rp.A;
    const long long cap = rp.cap;
    const int agent = blockIdx.y;
    const long long base = (long long)agent * cap;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = t0; i < n * S; i += stride) {
        rp.rs[base * S + i] = s[i];
        rp.rs2[base * S + i] = s2[i];
    }
    for (long long i = t0; i < n * A; i += stride) rp.ra[base * A + i] = a[i];
    for (long long i = t0; i < n; i += stride) {
        rp.rr[base + i] = r[i];
        rp.rg[base + i] = g[i];
    }
    if (t0 == 0) {
        rp.ring[agent].start = 0;
        rp.ring[agent].size = n;
    }
}

__global__ void rlc_replay_put1_kernel(RlcReplayDev rp, int agent, RlcPut1 t) {
    const int S = rp.S, A = rp.A;
    const long long slot = (long long)agent * rp.cap + t.slot;
    const int i = threadIdx.x;
    if (i < S) {
        rp.rs[slot * S + i] = t.sas[i];
        rp.rs2[slot * S + i] = t.sas[S + i];
    }
    if (i < A) rp.ra[slot * A + i] = t.sas[2 * S + i];
    if (i == 0) {
        rp.rr[slot] = t.r;
        rp.rg[slot] = t.g;
        rp.ring[agent].start = t.new_start;
        rp.ring[agent].size = t.new_size;
    }
}

__global__ void rlc_set_ring_kernel(RlcReplayDev rp, int agent, long long start, long long size) {
    rp.ring[agent].start = start;
    rp.ring[agent].size = size;
}

__global__ void rlc_replay_gather_kernel(RlcReplayDev rp, int agent, const long long* logical_idx, int k, float* s,
                                         float* a, double* r, float* s2, double* g) {
    const int S = rp.S, A = rp.A;
    const RlcRingMeta m = rp.ring[agent];
    const long long base = (long long)agent * rp.cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k; i += gridDim.x * blockDim.x) {
        const long long slot = base + ring_slot(m, rp.cap, logical_idx[i]);
        for (int j = 0; j < S; j++) {
            s[i * S + j] = rp.rs[slot * S + j];
            s2[i * S + j] = rp.rs2[slot * S + j];
        }
        for (int j = 0; j < A; j++) a[i * A + j] = rp.ra[slot * A + j];
        r[i] = rp.rr[slot];
        g[i] = rp.rg[slot];
    }
}

__global__ void rlc_sample_indices_kernel(RlcReplayDev rp, int agent, int k, long long* out_idx) {
    __shared__ int pool[3 * RLC_MAX_BATCH];
    __shared__ long long picks[RLC_MAX_BATCH];
    __shared__ int dups;
    const long long n = rp.ring[agent].size;
    const unsigned long long call = rp.sample_ctr[agent];
    rlc_sample_distinct(n, k, rp.seed[agent], call, pool, picks, &dups);
    __syncthreads();
    for (int i = threadIdx.x; i < k; i += blockDim.x) out_idx[i] = picks[i];
    if (threadIdx.x == 0) rp.sample_ctr[agent] = call + 1;
}

int rlc_launch_replay_scatter(const RlcReplayDev& rp, int agent, long long first_slot, long long n, const float* s,
                              const float* a, const double* r, const float* s2, const double* g, hipStream_t st) {
    if (n <= 0) return 0;
    const int threads = 256;
    long long blocks = (n + threads - 1) / threads;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rlc_replay_scatter_kernel, dim3((unsigned)blocks), dim3(threads), 0, st, rp, agent,
                       first_slot, n, s, a, r, s2, g);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_replay_fill_all(const RlcReplayDev& rp, long long n, const float* s, const float* a, const double* r,
                               const float* s2, const double* g, hipStream_t st) {
    const int threads = 256;
    long long blocks = (n * rp.S + threads - 1) / threads;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rlc_replay_fill_all_kernel, dim3((unsigned)blocks, rp.n_agents), dim3(threads), 0, st, rp,
                       n, s, a, r, s2, g);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_replay_gather(const RlcReplayDev& rp, int agent, const long long* logical_idx_dev, int k, float* s,
                             float* a, double* r, float* s2, double* g, hipStream_t st) {
    if (k <= 0) return 0;
    const int threads = 128;
    hipLaunchKernelGGL(rlc_replay_gather_kernel, dim3((k + threads - 1) / threads), dim3(threads), 0, st, rp,
                       agent, logical_idx_dev, k, s, a, r, s2, g);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_sample_indices(const RlcReplayDev& rp, int agent, int k, long long* out_idx_dev, hipStream_t st) {
    hipLaunchKernelGGL(rlc_sample_indices_kernel, dim3(1), dim3(256), 0, st, rp, agent, k, out_idx_dev);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_replay_put1(const RlcReplayDev& rp, int agent, const RlcPut1& t, hipStream_t st) {
    hipLaunchKernelGGL(rlc_replay_put1_kernel, dim3(1), dim3(64), 0, st, rp, agent, t);
    RLC_HIP(hipGetLastError());
    return 0;
}

int rlc_launch_set_ring(const RlcReplayDev& rp, int agent, long long start, long long size, hipStream_t st) {
    hipLaunchKernelGGL(rlc_set_ring_kernel, dim3(1), dim3(1), 0, st, rp, agent, start, size);
    RLC_HIP(hipGetLastError());
    return 0;
}
