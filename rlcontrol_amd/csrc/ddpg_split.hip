// ddpg_split.hip -- dispatch to the per-shape instantiations of the batch-split (latency mode) DDPG kernel
// (kernel: ddpg_split_kernel.h; instantiations: ddpg_split_inst.hip compiled per (MT, AD)).
#include "ddpg_split_kernel.h"

#ifdef RLC_ONLY_7_1   // developer loop (RLC_FAST_BUILD=1)
#define RLC_FOR_SPLIT(X) X(2, 1)
#else
#define RLC_FOR_SPLIT(X) X(1, 1) X(2, 1) X(4, 1) X(1, 2) X(2, 2) X(4, 2)
#endif

#define RLC_DECLS(M, A_) \
    int rlc_split_launch_##M##_##A_(const RlcDev&, float*, unsigned int*, int*, int, int, int, int, int, const long long*, int, hipStream_t);
RLC_FOR_SPLIT(RLC_DECLS)

// M tiles per workgroup when a minibatch of B rows is split over C workgroups: the smallest of {1, 2, 4} that covers it
int rlc_split_mt(int B, int C) {
    for (int mt : {1, 2, 4})
        if (mt * 16 * C >= B) return mt;
    return 0;
}

int rlc_launch_ddpg_update_split(const RlcDev& dv, float* part, unsigned int* bar, int* err, int C, int first_agent,
                                 int n_agents, int n_updates, int source, const long long* idx_dev, int grad_taps,
                                 hipStream_t st) {
    RLC_REQUIRE(rlc_mfma_supported(dv.d) && dv.d.blocked, "the split kernel runs on the MFMA kernel's shapes and layout");
    const int mt = rlc_split_mt(dv.d.B, C);
    RLC_REQUIRE(mt > 0, "batch_size %d does not fit %d workgroups of at most 64 rows", dv.d.B, C);
    RLC_HIP(hipMemsetAsync(bar + first_agent, 0, sizeof(unsigned int) * n_agents, st));
#define RLC_CASES(M, A_)          \
    if (mt == M && dv.d.A == A_)  \
        return rlc_split_launch_##M##_##A_(dv, part, bar, err, C, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st);
    RLC_FOR_SPLIT(RLC_CASES)
#undef RLC_CASES
    rlc_set_error("no split-kernel instantiation for MT=%d A=%d in this build", mt, dv.d.A);
    return 3;
}
