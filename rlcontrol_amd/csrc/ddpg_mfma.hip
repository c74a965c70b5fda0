// ddpg_mfma.hip -- shape check + dispatch to the per-shape instantiations of the MFMA DDPG kernel
// (kernel: ddpg_mfma_kernel.h; instantiations: ddpg_mfma_inst.hip compiled per (MT, AD)).
#include "ddpg_mfma_kernel.h"

#ifdef RLC_ONLY_7_1   // developer loop (RLC_FAST_BUILD=1): only the headline shape is compiled
#define RLC_FOR_V2(X) X(7, 1)
#else
#define RLC_FOR_V2(X) X(2, 1) X(4, 1) X(7, 1) X(8, 1) X(2, 2) X(4, 2) X(7, 2) X(8, 2)
#endif

#define RLC_DECL2(M, A_) \
    int rlc_mfma_launch_##M##_##A_(const RlcDev&, int, int, int, int, const long long*, int, hipStream_t, const RlcRollout*, int);
RLC_FOR_V2(RLC_DECL2)
// tail-of-four variants (compiled for the seven-tile shapes only: batch 97..100)
#ifdef RLC_ONLY_7_1
#define RLC_FOR_T4(X) X(7, 1)
#else
#define RLC_FOR_T4(X) X(7, 1) X(7, 2)
#endif
#define RLC_DECLT4(M, A_) \
    int rlc_mfma_launch_t4_##M##_##A_(const RlcDev&, int, int, int, int, const long long*, int, hipStream_t, const RlcRollout*, int);
RLC_FOR_T4(RLC_DECLT4)

static inline int mt_for(int B) { return B <= 32 ? 2 : (B <= 64 ? 4 : (B <= 112 ? 7 : 8)); }

bool rlc_mfma_supported(const RlcDims& d) {
    if (d.norm) return false;               // layer norm: the any-shape kernel (ddpg_generic.hip)
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (!(okdim(d.H1) && okdim(d.HA) && okdim(d.HC))) return false;
    if (d.S < 1 || d.S > SMAX) return false;
    if (d.A != 1 && d.A != 2) return false;
    if (d.B < 1 || d.B > 128) return false;
    return smem_carve(d, mt_for(d.B), nullptr, nullptr) <= 160 * 1024;
}

int rlc_launch_ddpg_update_mfma(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                const long long* idx_dev, int grad_taps, hipStream_t st, const RlcRollout* rollout,
                                int q8_first) {
    RLC_REQUIRE(rlc_mfma_supported(dv.d), "MFMA kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_ddpg_set_kernel re-packs them)");
    const int mt = mt_for(dv.d.B);
#define RLC_CASET4(M, A_)                                           \
    if (mt == M && dv.d.A == A_ && rlc_tail4_enabled(dv.d.B, M))    \
        return rlc_mfma_launch_t4_##M##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, rollout, \
                                             q8_first);
    RLC_FOR_T4(RLC_CASET4)
#undef RLC_CASET4
#define RLC_CASE2(M, A_)          \
    if (mt == M && dv.d.A == A_)  \
        return rlc_mfma_launch_##M##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, rollout, \
                                          q8_first);
    RLC_FOR_V2(RLC_CASE2)
#undef RLC_CASE2
    rlc_set_error("no MFMA instantiation for MT=%d A=%d in this build", mt, dv.d.A);
    return 3;
}
