// rollout_kernels.hip -- evaluation kernel of the on-device experiment loop (experiment.py:137-196 of the
// reference); the training step lives in ddpg_rollout_device.h and runs inside the fused update kernels.
#include "ddpg_rollout_device.h"

namespace {

constexpr int kThreads = RLC_POLICY_THREADS;

// ---- evaluation: run_episode_eval (experiment.py:163-196) for all episodes of one agent in one workgroup -----
// The E test episodes of an evaluation are independent (greedy policy, own reset draw), so they advance in
// lock step and share one pass over the actor weights per environment step (the weights, 160 KB at H=200,
// are the only sizeable traffic).  Per episode the arithmetic -- and its summation order -- is that of
// ddpg_greedy_forward, i.e. of the acting kernel.
#define RLC_EVAL_GROUP 16          // episodes advanced together (accumulators held in registers)

__global__ __launch_bounds__(kThreads) void rlc_ddpg_eval_kernel(RlcDev dv, RlcEnvDev env, int eval_round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const RlcDims d = dv.d;
    const int S = d.S, A = d.A, H1 = d.H1, HA = d.HA;
    const int agent = blockIdx.x, tid = threadIdx.x;
    const float* th = dv.theta + (size_t)agent * d.Ppad;
    // LDS: per episode of the group x[S] | h1[H1] | h2[HA] | act[A]; then the simulators
    const int SP = (S + 3) & ~3, H1P = (H1 + 3) & ~3, HAP = (HA + 3) & ~3, AP = (A + 3) & ~3;
    float* x = (float*)smem;                         // [G][SP]
    float* h1 = x + RLC_EVAL_GROUP * SP;             // [H1P][G]  (episode-minor: one 16-B read = 4 episodes)
    float* h2 = h1 + RLC_EVAL_GROUP * H1P;           // [G][HAP]
    float* act = h2 + RLC_EVAL_GROUP * HAP;          // [G][AP]
    double* sim = (double*)(act + RLC_EVAL_GROUP * AP);          // [G][RLC_ENV_STATE]
    double* obs = sim + RLC_EVAL_GROUP * RLC_ENV_STATE;          // [G][8]
    double* ret = obs + RLC_EVAL_GROUP * 8;                      // [G]
    int* alive = (int*)(ret + RLC_EVAL_GROUP);                   // [G]
    int* nsteps = alive + RLC_EVAL_GROUP;                        // [G]
    __shared__ int any_alive;

    for (int e0 = 0; e0 < env.eval_episodes; e0 += RLC_EVAL_GROUP) {
        const int G = min(RLC_EVAL_GROUP, env.eval_episodes - e0);
        __syncthreads();
        if (tid < G) {
            env_reset(env.env_id, sim + tid * RLC_ENV_STATE, obs + tid * 8, dv.rep.seed[agent] ^ RLC_KEY_ENV_TEST,
                      (unsigned long long)eval_round * env.eval_episodes + e0 + tid);
            ret[tid] = 0.0; alive[tid] = 1; nsteps[tid] = 0;
        }
        __syncthreads();
        for (int step = 0; step < env.episode_limit; step++) {
            for (int i = tid; i < G * S; i += kThreads) {
                const int e = i / S, c = i % S;
                x[e * SP + c] = clip_state_val((float)obs[e * 8 + c], dv.clip_state, dv.smin[c], dv.smax[c]);
            }
            __syncthreads();
            for (int k = tid; k < H1; k += kThreads) {
                float acc[RLC_EVAL_GROUP];
#pragma unroll
                for (int e = 0; e < RLC_EVAL_GROUP; e++) acc[e] = 0.0f;
                for (int i = 0; i < S; i++) {
                    const float w = th[d.oW1 + i * H1 + k];
#pragma unroll
                    for (int e = 0; e < RLC_EVAL_GROUP; e++) acc[e] += x[e * SP + i] * w;
                }
                const float bias = th[d.ob1 + k];
#pragma unroll
                for (int e = 0; e < RLC_EVAL_GROUP; e++) h1[k * RLC_EVAL_GROUP + e] = fmaxf(acc[e] + bias, 0.0f);
            }
            __syncthreads();
            for (int n = tid; n < HA; n += kThreads) {
                float acc[RLC_EVAL_GROUP];
#pragma unroll
                for (int e = 0; e < RLC_EVAL_GROUP; e++) acc[e] = 0.0f;
                const RlcWCol wcol = rlc_wcol(th + d.oWa2, d.blocked, n, HA);
                constexpr int KC = 8;                  // weight rows in flight per thread (divides the 16-row blocks)
                int k0 = 0;
                for (; k0 + KC <= H1; k0 += KC) {
                    const float* wp = wcol.at(k0);
                    float w[KC];
#pragma unroll
                    for (int i = 0; i < KC; i++) w[i] = wp[(size_t)i * wcol.step];
#pragma unroll
                    for (int i = 0; i < KC; i++) {
                        const float4* hk = reinterpret_cast<const float4*>(h1 + (k0 + i) * RLC_EVAL_GROUP);
#pragma unroll
                        for (int q = 0; q < RLC_EVAL_GROUP / 4; q++) {
                            const float4 v = hk[q];
                            acc[4 * q] += v.x * w[i]; acc[4 * q + 1] += v.y * w[i];
                            acc[4 * q + 2] += v.z * w[i]; acc[4 * q + 3] += v.w * w[i];
                        }
                    }
                }
                for (; k0 < H1; k0++) {
                    const float w = *wcol.at(k0);
                    const float4* hk = reinterpret_cast<const float4*>(h1 + k0 * RLC_EVAL_GROUP);
#pragma unroll
                    for (int q = 0; q < RLC_EVAL_GROUP / 4; q++) {
                        const float4 v = hk[q];
                        acc[4 * q] += v.x * w; acc[4 * q + 1] += v.y * w; acc[4 * q + 2] += v.z * w; acc[4 * q + 3] += v.w * w;
                    }
                }
                const float bias = th[d.oba2 + n];
#pragma unroll
                for (int e = 0; e < RLC_EVAL_GROUP; e++) h2[e * HAP + n] = fmaxf(acc[e] + bias, 0.0f);
            }
            __syncthreads();
            // output layer: one wave per (episode, action) pair, 64-lane shuffle reduction over HA
            const int wave = tid / RLC_WAVE, lane = tid % RLC_WAVE;
            for (int p = wave; p < G * A; p += kThreads / RLC_WAVE) {
                const int e = p / A, j = p % A;
                float a = 0.0f;
                for (int n = lane; n < HA; n += RLC_WAVE) a += h2[e * HAP + n] * th[d.oWa3 + n * A + j];
                for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, RLC_WAVE);
                if (lane == 0) act[e * AP + j] = tanhf(a + th[d.oba3 + j]) * dv.amax[j];
            }
            __syncthreads();
            if (tid == 0) any_alive = 0;
            __syncthreads();
            if (tid < G && alive[tid]) {
                double reward;
                const int done = env_step(env.env_id, sim + tid * RLC_ENV_STATE, act + tid * AP, obs + tid * 8, &reward,
                                          step + 1, env.episode_limit);
                ret[tid] += reward;
                nsteps[tid] = step + 1;
                if (done) alive[tid] = 0; else any_alive = 1;
            }
            __syncthreads();
            if (!any_alive) break;
        }
        if (tid < G && eval_round < env.max_evals) {
            const size_t at = ((size_t)agent * env.max_evals + eval_round) * env.eval_episodes + e0 + tid;
            env.eval_ret[at] = ret[tid];
            env.eval_len[at] = nsteps[tid];
        }
    }
}

// The variants (norm_type 'layer', separate networks): one greedy test episode per workgroup through
// ddpg_greedy_forward -- the acting kernel's own forward pass, layer norms included -- as the SAC / NAF loops do.
__global__ __launch_bounds__(kThreads) void rlc_ddpg_eval_rows_kernel(RlcDev dv, RlcEnvDev env, int eval_round) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double sim[RLC_ENV_STATE];
    __shared__ double obs[8];
    __shared__ int s_done;
    const RlcDims& d = dv.d;
    const int S = d.S;
    const int agent = blockIdx.x / env.eval_episodes, ep = blockIdx.x % env.eval_episodes;
    const int tid = threadIdx.x;
    const DdpgPolicyLds L = ddpg_policy_carve(d, (float*)smem);
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
        ddpg_greedy_forward(d, th, L, dv.amax);
        if (tid == 0) {
            double reward;
            s_done = env_step(env.env_id, sim, L.act, obs, &reward, steps + 1, env.episode_limit);
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

static size_t eval_lds_bytes(const RlcDims& d) {
    const size_t fl = (size_t)RLC_EVAL_GROUP * (((d.S + 3) & ~3) + ((d.H1 + 3) & ~3) + ((d.HA + 3) & ~3) + ((d.A + 3) & ~3));
    return sizeof(float) * fl + sizeof(double) * RLC_EVAL_GROUP * (RLC_ENV_STATE + 8 + 1) + sizeof(int) * RLC_EVAL_GROUP * 2;
}

}  // namespace

int rlc_launch_ddpg_eval(const RlcDev& dv, const RlcEnvDev& env, int eval_round, hipStream_t st) {
    if (dv.d.norm || dv.d.sep) {
        hipLaunchKernelGGL(rlc_ddpg_eval_rows_kernel, dim3(dv.n_agents * env.eval_episodes), dim3(kThreads),
                           sizeof(float) * ddpg_policy_lds_floats(dv.d), st, dv, env, eval_round);
        RLC_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = eval_lds_bytes(dv.d);
    RLC_REQUIRE(lds <= 160 * 1024, "evaluation kernel needs %zu B of LDS (> 160 KiB)", lds);
    hipLaunchKernelGGL(rlc_ddpg_eval_kernel, dim3(dv.n_agents), dim3(kThreads), lds, st, dv, env, eval_round);
    RLC_HIP(hipGetLastError());
    return 0;
}
