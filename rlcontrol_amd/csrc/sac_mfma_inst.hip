// sac_mfma_inst.hip -- one instantiation of the MFMA SAC kernel per translation unit
// (compiled once per (RLC_MT, RLC_NTW, RLC_AD) triple by rlcontrol_amd/build.py so the variants build in parallel).
#ifndef RLC_WG_NO_EXACT
#define RLC_WG_EXACT 1      // mfma_blocks.h wgrad_adam: paired activation reads in exact items (+4 % at widths <= 128)
#endif
#ifndef RLC_WG_NO_LATE_ISSUE
#define RLC_WG_LATE_ISSUE 1 // mfma_blocks.h wgrad_adam: second prefetch issued after the first k-loop (+4.5 % at widths <= 128;
#endif                      // DDPG +-0, NAF -1.6 %: profiles/r03_variant_timings_s27.txt)
#include "sac_mfma_kernel.h"

#ifndef RLC_MT
#error "compile with -DRLC_MT=<M tiles> -DRLC_NTW=<N tiles per wave> -DRLC_AD=<action dim>"
#endif

#ifndef RLC_T4
#define RLC_T4 0            // 1: the tail-of-four variant (mfma_blocks.h, Blk's T4), entry point rlc_sac_mfma_launch_t4_<MT>_<NTW>_<AD>
#endif
#if RLC_T4
#define RLC_CAT_(a, b, c) rlc_sac_mfma_launch_t4_##a##_##b##_##c
#else
#define RLC_CAT_(a, b, c) rlc_sac_mfma_launch_##a##_##b##_##c
#endif
#define RLC_CAT(a, b, c) RLC_CAT_(a, b, c)

int RLC_CAT(RLC_MT, RLC_NTW, RLC_AD)(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                     const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                                     const RlcSacRollout* rollout) {
    return sac_launch_t<RLC_MT, RLC_NTW, RLC_AD, RLC_T4 != 0>(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, grad_taps,
                                                 st, rollout);
}
