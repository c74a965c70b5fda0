// sac_mfma.hip -- shape check + dispatch to the per-shape instantiations of the MFMA SAC kernel
// (kernel: sac_mfma_kernel.h; instantiations: sac_mfma_inst.hip compiled per (MT, NTW, AD)).
#include "sac_mfma_kernel.h"

#ifdef RLC_ONLY_7_1   // developer loop (RLC_FAST_BUILD=1): only the BASELINE shape is compiled
#define RLC_FOR_SAC(X) X(7, 1, 1)
#else
#define RLC_FOR_SAC(X)                                                                  \
    X(2, 1, 1) X(4, 1, 1) X(7, 1, 1) X(8, 1, 1) X(2, 2, 1) X(4, 2, 1) X(7, 2, 1) X(8, 2, 1) \
    X(2, 1, 2) X(4, 1, 2) X(7, 1, 2) X(8, 1, 2) X(2, 2, 2) X(4, 2, 2) X(7, 2, 2) X(8, 2, 2)
#endif

#define RLC_DECL3(M, N_, A_)                                                                                        \
    int rlc_sac_mfma_launch_##M##_##N_##_##A_(const RlcSacDev&, int, int, int, int, const long long*, const float*, int, \
                                              hipStream_t, const RlcSacRollout*);
RLC_FOR_SAC(RLC_DECL3)
// tail-of-four variants (compiled for the seven-tile shapes only: batch 97..100)
#ifdef RLC_ONLY_7_1
#define RLC_FOR_SAC_T4(X) X(7, 1, 1)
#else
#define RLC_FOR_SAC_T4(X) X(7, 1, 1) X(7, 2, 1) X(7, 1, 2) X(7, 2, 2)
#endif
#define RLC_DECLT4(M, N_, A_)                                                                                          \
    int rlc_sac_mfma_launch_t4_##M##_##N_##_##A_(const RlcSacDev&, int, int, int, int, const long long*, const float*, int, \
                                                 hipStream_t, const RlcSacRollout*);
RLC_FOR_SAC_T4(RLC_DECLT4)

static inline int sac_mt_for(int B) { return B <= 32 ? 2 : (B <= 64 ? 4 : (B <= 112 ? 7 : 8)); }
static inline int sac_ntw_for(const RlcSacDims& d) {
    const int w = d.L2A > d.L2C ? d.L2A : d.L2C, k = d.L1A > d.L1C ? d.L1A : d.L1C;
    return (w <= 128 && k <= 128) ? 1 : 2;
}

bool rlc_sac_mfma_supported(const RlcSacDims& d) {
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (!(okdim(d.L1A) && okdim(d.L2A) && okdim(d.L1C) && okdim(d.L2C))) return false;
    if (d.qcat) return false;   // the KL agents' input-concatenated Q network is not an SAC shape
    if (d.norm) return false;   // layer norm: the any-shape kernel (sac_generic.hip)
    if (d.S < 1 || d.S > SMAX) return false;
    if (d.A != 1 && d.A != 2) return false;
    if (d.B < 1 || d.B > 128) return false;
    const int mt = sac_mt_for(d.B);
    const size_t lds = sac_ntw_for(d) == 1 ? ssmem_carve<mask_stride(8)>(d, mt, nullptr, nullptr)
                                           : ssmem_carve<mask_stride(16)>(d, mt, nullptr, nullptr);
    return lds <= 160 * 1024;
}

int rlc_launch_sac_update_mfma(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                               const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                               const RlcSacRollout* rollout) {
    RLC_REQUIRE(rlc_sac_mfma_supported(dv.d), "MFMA SAC kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_sac_set_kernel re-packs them)");
    RLC_REQUIRE(!(rollout && eps_dev), "the on-device loop draws its own eps");
    const int mt = sac_mt_for(dv.d.B), ntw = sac_ntw_for(dv.d);
#define RLC_CASET4(M, N_, A_)                                                  \
    if (mt == M && ntw == N_ && dv.d.A == A_ && rlc_tail4_enabled(dv.d.B, M))  \
        return rlc_sac_mfma_launch_t4_##M##_##N_##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, \
                                                        grad_taps, st, rollout);
    RLC_FOR_SAC_T4(RLC_CASET4)
#undef RLC_CASET4
#define RLC_CASE3(M, N_, A_)                       \
    if (mt == M && ntw == N_ && dv.d.A == A_)      \
        return rlc_sac_mfma_launch_##M##_##N_##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, \
                                                     grad_taps, st, rollout);
    RLC_FOR_SAC(RLC_CASE3)
#undef RLC_CASE3
    rlc_set_error("no MFMA SAC instantiation for MT=%d NTW=%d A=%d in this build", mt, ntw, dv.d.A);
    return 3;
}
