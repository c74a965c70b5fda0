// ddpg_mfma_inst.hip -- one instantiation of the MFMA DDPG kernel per translation unit
// (compiled once per (RLC_MT, RLC_AD) pair by rlcontrol_amd/build.py so the variants build in parallel).
#include "ddpg_mfma_kernel.h"

#ifndef RLC_MT
#error "compile with -DRLC_MT=<M tiles> -DRLC_AD=<action dim>"
#endif

#ifndef RLC_T4
#define RLC_T4 0            // 1: the tail-of-four variant (mfma_blocks.h, Blk's T4), entry point rlc_mfma_launch_t4_<MT>_<AD>
#endif
#if RLC_T4
#define RLC_CAT_(a, b, c) rlc_mfma_launch_t4_##a##_##b
#else
#define RLC_CAT_(a, b, c) rlc_mfma_launch_##a##_##b
#endif
#define RLC_CAT(a, b) RLC_CAT_(a, b, 0)

int RLC_CAT(RLC_MT, RLC_AD)(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                            const long long* idx_dev, int grad_taps, hipStream_t st, const RlcRollout* rollout,
                            int q8_first) {
    return launch_t<RLC_MT, RLC_AD, RLC_T4 != 0>(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, rollout,
                                    q8_first);
}
