// ddpg_split_inst.hip -- one instantiation of the batch-split DDPG kernel per translation unit
// (compiled once per (RLC_MT, RLC_AD) pair by rlcontrol_amd/build.py).
#include "ddpg_split_kernel.h"

#ifndef RLC_MT
#error "compile with -DRLC_MT=<M tiles per workgroup> -DRLC_AD=<action dim>"
#endif

#define RLC_CAT_(a, b, c) rlc_split_launch_##a##_##b
#define RLC_CAT(a, b) RLC_CAT_(a, b, 0)

int RLC_CAT(RLC_MT, RLC_AD)(const RlcDev& dv, float* part, unsigned int* bar, int* err, int C, int first_agent,
                            int n_agents, int n_updates, int source, const long long* idx_dev, int grad_taps,
                            hipStream_t st) {
    RlcSplit sp;
    sp.part = part; sp.bar = bar; sp.err = err; sp.C = C;
    return launch_split_t<RLC_MT, RLC_AD>(dv, sp, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st);
}
