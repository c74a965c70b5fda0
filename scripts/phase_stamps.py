#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of the MFMA DDPG kernel (needs the RLC_STAMPS=1 build).
   RLC_STAMPS=1 python rlcontrol_amd/build.py ; RLCONTROL_HIP_LIB=rlcontrol_amd/librlcontrol_hip_stamps.so python scripts/phase_stamps.py
Shares only; the stamped build's absolute run time is not a performance number."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params  # noqa: E402

NAMES_HBUF = ["sample+gather", "trunk(target)", "G1 actor' fwd", "a' tanh", "G2 critic' fwd", "TD target", "trunk(online)",
         "G3 critic fwd", "dq", "dWc3/dbc2+masks", "G5 dh1 (bwd)", "trunk grad+Adam", "G4 dWc2+Adam+small", "trunk(new)",
         "G6 actor fwd", "mu tanh", "G7 critic(mu) fwd", "dz", "G9 dh1 (bwd)", "trunk grad+Adam (a)", "G8 dWa2+Adam+small"]


def main():
    KERN = "mfma"
    NAMES = NAMES_HBUF
    NA, U, N = int(os.environ.get("NA", "256")), 8, 200000
    BATCH = int(os.environ.get("B", "100"))              # B=32: the shipped jsons' batch size
    pop = DDPGPopulation(NA, 3, 1, 200, 200, 200, BATCH, N, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0], 1e-3, 1e-2,
                         seeds=np.arange(NA) + 1)
    pop.set_kernel(KERN)
    pop.enable_grad_taps(True)
    rng = np.random.RandomState(0)
    data = (rng.randn(N, 3), rng.randn(N, 1), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    for i in range(NA):
        pop.set_params(i, init_params(3, 1, 200, 200, 200, i))
        pop.replay_add_batch(i, *data) if i < 1 else None
    # only agent 0 has data in this diagnostic unless the fill helper is used; fill all via per-agent adds
    for i in range(1, NA):
        pop.replay_add_batch(i, *[x[:2000] for x in data])
    pop.update(U)
    pop.timer_begin()
    pop.update(U)
    ms = pop.timer_end()
    tot = np.zeros(len(NAMES))
    for ag in (0, NA // 2, NA - 1):
        full = pop.last_tap(ag, "grads_c")
        st = full[:len(NAMES)]
        tot += st / U
        print("agent %d: clock64 total %.0f, wall_clock64 total %.0f (100 MHz -> %.1f us) => clock64 at %.0f MHz" % (
            ag, full[40], full[41], full[41] / 100.0, full[40] / (full[41] / 100.0)))
    tot /= 3
    print("kernel %.3f ms for %d updates/agent -> %.1f us/update" % (ms, U, ms * 1e3 / U))
    print("stamped cycles/update %.0f (%.1f us at 2.4 GHz)" % (tot.sum(), tot.sum() / 2400))
    for n, c in zip(NAMES, tot):
        print("  %-24s %9.0f cyc  %5.1f %%" % (n, c, 100 * c / tot.sum()))
    pw = pop.last_tap(0, "grads_c")[48:56] / U / 5
    print("  forward-GEMM k-loop cycles per wave (mean of the 5 GEMMs): " + " ".join("%.0f" % v for v in pw))
    pb = pop.last_tap(0, "grads_c")[56:64] / U / 2
    print("  backward-GEMM k-loop cycles per wave (mean of the 2 GEMMs): " + " ".join("%.0f" % v for v in pb))
    sg = pop.last_tap(0, "grads_c")[25:28] / U
    print("  thread 0 inside sample+gather: ring/counter loads %.0f, sampler %.0f, gather %.0f" % tuple(sg))
    sub = pop.last_tap(0, "grads_c")[21:25] / U
    print("  wave-0 inside both wgrad GEMMs: prefetch-issue %.0f, k-loop %.0f, epilogue(+next prefetch wait) %.0f, action rows %.0f" % tuple(sub))
    g = pop.last_tap(0, "grads_c") / U
    print("  wave-0 between items (next prefetch issue + geometry) %.0f; critic phase: small tensors %.0f, barrier wait %.0f" % (g[28], g[29], g[30]))
    print("  per-wave cycles inside both wgrad calls: " + " ".join("%.0f" % v for v in g[32:40]))


if __name__ == "__main__":
    main()
