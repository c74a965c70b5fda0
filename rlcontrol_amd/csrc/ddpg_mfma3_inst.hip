// ddpg_mfma3_inst.hip -- one instantiation of the trunk-on-the-fly MFMA DDPG kernel per translation unit
// (compiled once per (RLC_MT, RLC_AD, RLC_SP) by rlcontrol_amd/build.py so the variants build in parallel).
#include "ddpg_mfma3_kernel.h"

#ifndef RLC_MT
#error "compile with -DRLC_MT=<M tiles> -DRLC_AD=<action dim> -DRLC_SP=<state row stride 4|8>"
#endif

#define RLC_CAT_(a, b, c) rlc_mfma3_launch_##a##_##b##_##c
#define RLC_CAT(a, b, c) RLC_CAT_(a, b, c)

int RLC_CAT(RLC_MT, RLC_AD, RLC_SP)(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                    const long long* idx_dev, int grad_taps, int stagger, hipStream_t st) {
    return mf3::launch_t<RLC_MT, RLC_AD, RLC_SP>(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps,
                                                 stagger, st);
}
