// naf_mfma.hip -- shape check + dispatch to the per-shape instantiations of the MFMA NAF kernel
// (kernel: naf_mfma_kernel.h; instantiations: naf_mfma_inst.hip compiled per (MT, NTW, AD)).
#include "naf_mfma_kernel.h"

#ifdef RLC_ONLY_7_1   // developer loop (RLC_FAST_BUILD=1): only the BASELINE shape is compiled
#define RLC_FOR_NAF(X) X(7, 2, 2)
#else
#define RLC_FOR_NAF(X)                                                                  \
    X(2, 1, 1) X(4, 1, 1) X(7, 1, 1) X(8, 1, 1) X(2, 2, 1) X(4, 2, 1) X(7, 2, 1) X(8, 2, 1) \
    X(2, 1, 2) X(4, 1, 2) X(7, 1, 2) X(8, 1, 2) X(2, 2, 2) X(4, 2, 2) X(7, 2, 2) X(8, 2, 2)
#endif

#define RLC_DECL3(M, N_, A_)                                                                                  \
    int rlc_naf_mfma_launch_##M##_##N_##_##A_(const RlcNafDev&, int, int, int, int, const long long*, int, hipStream_t, \
                                              const RlcNafRollout*);
RLC_FOR_NAF(RLC_DECL3)
// tail-of-four variants (compiled for the seven-tile shapes only: batch 97..100)
#ifdef RLC_ONLY_7_1
#define RLC_FOR_NAF_T4(X) X(7, 2, 2)
#else
#define RLC_FOR_NAF_T4(X) X(7, 1, 1) X(7, 2, 1) X(7, 1, 2) X(7, 2, 2)
#endif
#define RLC_DECLT4(M, N_, A_)                                                                                    \
    int rlc_naf_mfma_launch_t4_##M##_##N_##_##A_(const RlcNafDev&, int, int, int, int, const long long*, int, hipStream_t, \
                                                 const RlcNafRollout*);
RLC_FOR_NAF_T4(RLC_DECLT4)

static inline int naf_mt_for(int B) { return B <= 32 ? 2 : (B <= 64 ? 4 : (B <= 112 ? 7 : 8)); }
static inline int naf_ntw_for(const RlcNafDims& d) { return (d.L1 <= 128 && d.L2 <= 128) ? 1 : 2; }

bool rlc_naf_mfma_supported(const RlcNafDims& d) {
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (d.norm) return false;       // layer norm: any-shape kernel only
    if (!(okdim(d.L1) && okdim(d.L2))) return false;
    if (d.S < 1 || d.S > SMAX) return false;
    if (d.A != 1 && d.A != 2) return false;
    if (d.B < 1 || d.B > 128) return false;
    const int mt = naf_mt_for(d.B);
    const size_t lds = naf_ntw_for(d) == 1 ? nsmem_carve<mask_stride(8)>(d, mt, nullptr, nullptr)
                                           : nsmem_carve<mask_stride(16)>(d, mt, nullptr, nullptr);
    return lds <= 160 * 1024;
}

int rlc_launch_naf_update_mfma(const RlcNafDev& dv, int first_agent, int n_agents, int n_updates, int source,
                               const long long* idx_dev, int grad_taps, hipStream_t st, const RlcNafRollout* rollout) {
    RLC_REQUIRE(rlc_naf_mfma_supported(dv.d), "MFMA NAF kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_naf_set_kernel re-packs them)");
    const int mt = naf_mt_for(dv.d.B), ntw = naf_ntw_for(dv.d);
#define RLC_CASET4(M, N_, A_)                                                  \
    if (mt == M && ntw == N_ && dv.d.A == A_ && rlc_tail4_enabled(dv.d.B, M))  \
        return rlc_naf_mfma_launch_t4_##M##_##N_##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, \
                                                        rollout);
    RLC_FOR_NAF_T4(RLC_CASET4)
#undef RLC_CASET4
#define RLC_CASE3(M, N_, A_)                       \
    if (mt == M && ntw == N_ && dv.d.A == A_)      \
        return rlc_naf_mfma_launch_##M##_##N_##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st, \
                                                     rollout);
    RLC_FOR_NAF(RLC_CASE3)
#undef RLC_CASE3
    rlc_set_error("no MFMA NAF instantiation for MT=%d NTW=%d A=%d in this build", mt, ntw, dv.d.A);
    return 3;
}
