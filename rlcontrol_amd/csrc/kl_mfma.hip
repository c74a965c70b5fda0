// kl_mfma.hip -- shape check + launch of the MFMA ReverseKL / ForwardKL update kernel (kl_mfma_kernel.h).
// One instantiation: 2 batch tiles for the three small networks (batch_size <= 32, the reference's default),
// 7 batch tiles per pass of the action integral.
#include "kl_mfma_kernel.h"

bool rlc_kl_mfma_supported(const RlcSacDims& d, int nodes) {
    auto okdim = [](int h) { return h >= 16 && h <= 256 && (h % 4) == 0; };
    if (!d.qcat || d.A != 1) return false;
    if (!(okdim(d.L1A) && okdim(d.L2A) && okdim(d.L1C) && okdim(d.L2C))) return false;
    if (d.S < 1 || d.S + 1 > SMAX) return false;
    if (d.B < 1 || d.B > 32) return false;
    if (nodes < 0 || nodes > KL_MAXNODES) return false;
    return ksmem_carve(d, 2, 7, nullptr, nullptr) <= 160 * 1024;
}

int rlc_launch_kl_update_mfma(const RlcSacDev& dv, int first_agent, int n_agents, int n_updates, int source,
                              const long long* idx_dev, const float* eps_dev, int grad_taps, hipStream_t st,
                              const RlcSacRollout* rollout) {
    RLC_REQUIRE(rlc_kl_mfma_supported(dv.d, dv.kl_nodes), "MFMA KL kernel does not support these dimensions");
    RLC_REQUIRE(dv.d.blocked, "the MFMA kernel reads tile-blocked weights (rlc_kl_set_kernel re-packs them)");
    return kl_launch_t<2, 7>(dv, first_agent, n_agents, n_updates, source, idx_dev, eps_dev, grad_taps, st, rollout);
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
    const KlSplit sp = {zbuf, bar, err, C, n_agents};
    return kl_launch_split_t<2, 7>(dv, sp, first_agent, n_updates, source, idx_dev, eps_dev, grad_taps, st);
}
