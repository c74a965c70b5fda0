#!/usr/bin/env python3
"""Diagnostic: per-phase shader cycles of the MFMA NAF kernel (needs the RLC_STAMPS=1 build; see phase_stamps_sac.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlcontrol_amd.hip_naf import NAFPopulation, init_params  # noqa: E402

NAMES = ["sample+gather", "1 target V'(s') + TD glue", "2 online forward (trunk, mu | V pair, heads)", "3 per-sample L columns, Q, seeds",
         "4 output-layer / bias gradients", "5 backward (V branch + mu branch)", "5 trunk grad+Adam (+heads)", "5b small tensors",
         "6 two wgrad calls + small"]


def main():
    NA, U, N, B = int(os.environ.get("NA", "256")), 8, 100000, int(os.environ.get("B", "100"))
    S, A = 8, 2
    pop = NAFPopulation(NA, S, A, 200, 200, B, N, 0.01, [-1.0] * S, [1.0] * S, [1.0] * A, 1e-3, seeds=list(range(1, NA + 1)))
    pop.set_kernel("mfma")
    pop.enable_grad_taps(True)
    rng = np.random.RandomState(0)
    data = (rng.randn(2000, S), rng.uniform(-1, 1, (2000, A)), rng.randn(2000), rng.randn(2000, S), np.full(2000, 0.99))
    for i in range(NA):
        pop.set_params(i, init_params(S, A, 200, 200, i))
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
        print("  %-46s %9.0f cyc  %5.1f %%" % (n, c, 100 * c / tot.sum()))
    g = pop.last_tap(0, "grads") / U
    print("  wave-0 inside the two wgrad calls: prefetch-issue %.0f, k-loop %.0f, epilogue %.0f, action rows %.0f, between items %.0f"
          % (g[21], g[22], g[23], g[24], g[28]))
    print("  per-wave cycles inside the wgrad calls: " + " ".join("%.0f" % v for v in g[32:40]))


if __name__ == "__main__":
    main()
