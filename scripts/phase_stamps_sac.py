#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of the MFMA SoftActorCritic kernel (needs the RLC_STAMPS=1 build).
   RLC_STAMPS=1 RLC_FAST_BUILD=1 python -c "from rlcontrol_amd import build as B; B.build()"
   RLCONTROL_HIP_LIB=rlcontrol_amd/librlcontrol_hip_stamps.so python scripts/phase_stamps_sac.py
Shares only (thread 0's view); the stamped build's absolute run time is not a performance number."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcontrol_amd.hip_sac import SACPopulation, init_params  # noqa: E402

NAMES = ["sample+gather", "1 V'(s'): trunk+fwd+head", "2 Q hidden: trunk+fwd", "3 pi forward", "4 Q(s,pi), Q(s,a), sums, seeds",
         "5 pi: small grads + bwd", "5 pi: trunk grad+Adam", "5 pi: wgrad+Adam+small", "6 Q: trunk + bwd", "6 Q: trunk grad+Adam",
         "6 Q: wgrad+Adam+small", "7 V: fwd + sums + bwd", "7 V: trunk grad+Adam", "7 V: wgrad+Adam+small"]


def main():
    NA, U, N, B = int(os.environ.get("NA", "256")), 8, 100000, int(os.environ.get("B", "100"))
    pop = SACPopulation(NA, 3, 1, 128, 128, 128, 128, B, N, 0.01, -8.0, 8.0, 2.0, 1e-3, 1e-3, 0.1, seeds=np.arange(NA) + 1)
    pop.set_kernel("mfma")
    pop.enable_grad_taps(True)
    rng = np.random.RandomState(0)
    data = (rng.randn(2000, 3), rng.randn(2000, 1), rng.randn(2000), rng.randn(2000, 3), np.full(2000, 0.99))
    for i in range(NA):
        pop.set_params(i, init_params(3, 1, 128, 128, 128, 128, i))
        pop.replay_add_batch(i, *data)
    pop.update(U)
    pop.timer_begin()
    pop.update(U)
    ms = pop.timer_end()
    tot = np.zeros(len(NAMES))
    for ag in (0, NA // 2, NA - 1):
        tot += pop.last_tap(ag, "grads")[:len(NAMES)] / U
    tot /= 3
    print("kernel %.3f ms for %d updates/agent -> %.1f us/update" % (ms, U, ms * 1e3 / U))
    print("stamped cycles/update %.0f" % tot.sum())
    for n, c in zip(NAMES, tot):
        print("  %-34s %9.0f cyc  %5.1f %%" % (n, c, 100 * c / tot.sum()))
    g = pop.last_tap(0, "grads") / U
    print("  wave-0 inside the three wgrad calls: prefetch-issue %.0f, k-loop %.0f, epilogue %.0f, action rows %.0f, between items %.0f"
          % (g[21], g[22], g[23], g[24], g[28]))
    print("  per-wave cycles inside the wgrad calls: " + " ".join("%.0f" % v for v in g[32:40]))


if __name__ == "__main__":
    main()
