// kl_mfma.hip -- shape check + launch of the MFMA ReverseKL / ForwardKL update kernel (kl_mfma_kernel.h).
// One instantiation: 2 batch tiles for the three small networks (batch_size <= 32, the reference's default),
// 7 batch tiles per pass of the action integral.
#include "kl_mfma_kernel.h"

// batch tiles of the three small networks: 2 (the reference's default batch 32), 7 (BASELINE's 100) or 8; the node passes
// always run at 7 (8) tiles
static inline int kl_mt_for(int B) { return B <= 32 ? 2 : (B <= 112 ? 7 : 8); }

bool rlc_kl_mfma_supported(const RlcSacDims& d, int nodes) {
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (!d.qcat || d.A != 1) return false;
    if (!(okdim(d.L1A) && okdim(d.L2A) && okdim(d.L1C) && okdim(d.L2C))) return false;
    if (d.S < 1 || d.S + 1 > SMAX) return false;
    if (d.B < 1 || d.B > 128) return false;
    if (nodes < 0 || nodes > KL_MAXNODES) return false;
    const int mt = kl_mt_for(d.B);
    return ksmem_carve(d, mt, mt == 8 ? 8 : 7, nullptr, nullptr) <= 160 * 1024;
}

size_t rlc_kl_mfma_scratch_floats(const RlcSacDims& d, int nodes) { return kl_mfma_scratch_floats(d, nodes, kl_mt_for(d.B)); }

int rlc_launch_kl_update_mfma(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                              const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                              const RlcSacRollout* rollout) {
    RLC_REQUIRE(rlc_kl_mfma_supported(dv.d, dv.kl_nodes), "MFMA KL kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_kl_set_kernel re-packs them)");
    RLC_REQUIRE((size_t)dv.scratch_stride >= rlc_kl_mfma_scratch_floats(dv.d, dv.kl_nodes), "KL scratch row too short for the MFMA kernel");
    const int mt = kl_mt_for(dv.d.B);
    if (mt == 2) return kl_launch_t<2, 7>(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, grad_taps, st, rollout);
    if (mt == 7) return kl_launch_t<7, 7>(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, grad_taps, st, rollout);
    return kl_launch_t<8, 8>(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, grad_taps, st, rollout);
}

// latency mode: the node passes of every agent's action integral dealt over C workgroups (kl_mfma_kernel.h)
size_t rlc_kl_split_zbuf_floats(const RlcSacDims& d) { return (size_t)32 * kl_mfma_ldh(d); }
int rlc_kl_split_grid(int n_agents, int C) { return kl_split_grid(n_agents, C); }

int rlc_launch_kl_update_mfma_split(const RlcSacDev& dv, float* zbuf, unsigned int* bar, int* err, int C, int first_agent,
                                    int n_agents, int n_updates, int source, const long long* idx_dev, const float* eps_dev,
                                    int grad_taps, hipStream_t st) {
    RLC_REQUIRE(rlc_kl_mfma_supported(dv.d, dv.kl_nodes), "MFMA KL kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_kl_set_kernel re-packs them)");
    RLC_REQUIRE(dv.kl_optim == RLC_KL_OPTIM_INTG || dv.kl_optim == RLC_KL_OPTIM_HARD_INTG,
                "latency mode splits the action integral; the 'll' updates have none");
    RLC_REQUIRE(dv.d.B <= 32, "latency mode of the KL agents is built for batch sizes up to 32 (got %d)", dv.d.B);
    const KlSplit sp = {zbuf, bar, err, C, n_agents};
    return kl_launch_split_t<2, 7>(dv, sp, first_agent, n_updates, source, idx_dev, eps_dev, grad_taps, st);
}
