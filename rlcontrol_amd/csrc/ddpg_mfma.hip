// ddpg_mfma.hip -- shape checks + dispatch to the per-shape instantiations of the MFMA DDPG kernels.
//   variant 2 "mfma"      : ddpg_mfma3_kernel.h, trunk recomputed on the fly, 4 waves, two agents per CU
//   variant 3 "mfma_hbuf" : ddpg_mfma_kernel.h, trunk activation kept in LDS, 8 waves, one agent per CU
#include "ddpg_mfma3_kernel.h"
#include "ddpg_mfma_kernel.h"

#ifdef RLC_ONLY_7_1   // developer loop (RLC_FAST_BUILD=1): only the headline shape is compiled
#define RLC_FOR_V2(X) X(7, 1)
#define RLC_FOR_V3(X) X(7, 1, 4)
#else
#define RLC_FOR_V2(X) X(2, 1) X(4, 1) X(7, 1) X(8, 1) X(2, 2) X(4, 2) X(7, 2) X(8, 2)
#define RLC_FOR_V3(X)                                                              \
    X(2, 1, 4) X(4, 1, 4) X(7, 1, 4) X(8, 1, 4) X(2, 2, 4) X(4, 2, 4) X(7, 2, 4) X(8, 2, 4) \
    X(2, 1, 8) X(4, 1, 8) X(7, 1, 8) X(8, 1, 8) X(2, 2, 8) X(4, 2, 8) X(7, 2, 8) X(8, 2, 8)
#endif

#define RLC_DECL2(M, A_) \
    int rlc_mfma_launch_##M##_##A_(const RlcDev&, int, int, int, int, const long long*, int, hipStream_t);
#define RLC_DECL3(M, A_, S_) \
    int rlc_mfma3_launch_##M##_##A_##_##S_(const RlcDev&, int, int, int, int, const long long*, int, int, hipStream_t);
RLC_FOR_V2(RLC_DECL2)
RLC_FOR_V3(RLC_DECL3)

static inline int mt_for(int B) { return B <= 32 ? 2 : (B <= 64 ? 4 : (B <= 112 ? 7 : 8)); }
static inline int sp_for(int S) { return S <= 4 ? 4 : 8; }

static bool dims_ok(const RlcDims& d) {
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (!(okdim(d.H1) && okdim(d.HA) && okdim(d.HC))) return false;
    if (d.S < 1 || d.S > 8) return false;
    if (d.A != 1 && d.A != 2) return false;
    return d.B >= 1 && d.B <= 128;
}

bool rlc_mfma_supported(const RlcDims& d) {
    return dims_ok(d) && mf3::smem_carve(d, mt_for(d.B), sp_for(d.S), nullptr, nullptr) <= 64 * 1024;
}

bool rlc_mfma_hbuf_supported(const RlcDims& d) {
    return dims_ok(d) && smem_carve(d, mt_for(d.B), nullptr, nullptr) <= 160 * 1024;
}

size_t rlc_mfma_scratch_floats(const RlcDims& d) { return mf3::park_floats(mt_for(d.B)); }

int rlc_launch_ddpg_update_mfma(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                const long long* idx_dev, int grad_taps, hipStream_t st) {
    RLC_REQUIRE(rlc_mfma_supported(dv.d), "MFMA kernel does not support these dimensions");
    const int mt = mt_for(dv.d.B), sp = sp_for(dv.d.S);
#define RLC_CASE3(M, A_, S_)                                 \
    if (mt == M && dv.d.A == A_ && sp == S_)                 \
        return rlc_mfma3_launch_##M##_##A_##_##S_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, \
                                                  dv.stagger, st);
    RLC_FOR_V3(RLC_CASE3)
#undef RLC_CASE3
    rlc_set_error("no MFMA instantiation for MT=%d A=%d SP=%d in this build", mt, dv.d.A, sp);
    return 3;
}

int rlc_launch_ddpg_update_mfma_hbuf(const RlcDev& dv, int first_agent, int n_agents, int n_updates, int source,
                                     const long long* idx_dev, int grad_taps, hipStream_t st) {
    RLC_REQUIRE(rlc_mfma_hbuf_supported(dv.d), "MFMA (hbuf) kernel does not support these dimensions");
    const int mt = mt_for(dv.d.B);
#define RLC_CASE2(M, A_)          \
    if (mt == M && dv.d.A == A_)  \
        return rlc_mfma_launch_##M##_##A_(dv, first_agent, n_agents, n_updates, source, idx_dev, grad_taps, st);
    RLC_FOR_V2(RLC_CASE2)
#undef RLC_CASE2
    rlc_set_error("no MFMA (hbuf) instantiation for MT=%d A=%d in this build", mt, dv.d.A);
    return 3;
}
