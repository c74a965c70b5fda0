"""Child process of tests/test_gpu_tail4.py: a few updates of one algorithm at a batch size of the tail-of-four range on
fixed weights and data; dumps the parameters.  RLC_NO_TAIL4=1 in the environment selects the padded kernels."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    algo, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rng = np.random.RandomState(11)
    N = 600
    if algo == "ddpg":
        from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params
        pop = DDPGPopulation(2, 3, 1, 200, 200, 200, B, N, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0], 1e-3, 1e-2, seeds=[1, 2])
        pop.set_kernel("mfma")
        for i in range(2):
            pop.set_params(i, init_params(3, 1, 200, 200, 200, 5 + i))
        data = (rng.randn(N, 3), rng.uniform(-2, 2, (N, 1)), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    elif algo == "sac":
        from rlcontrol_amd.hip_sac import SACPopulation, init_params
        pop = SACPopulation(2, 3, 1, 128, 128, 128, 128, B, N, 0.01, -8.0, 8.0, 2.0, 1e-3, 1e-3, 0.1, seeds=[1, 2])
        pop.set_kernel("mfma")
        for i in range(2):
            pop.set_params(i, init_params(3, 1, 128, 128, 128, 128, 5 + i))
        data = (rng.randn(N, 3), rng.uniform(-2, 2, (N, 1)), rng.randn(N), rng.randn(N, 3), np.full(N, 0.99))
    else:
        from rlcontrol_amd.hip_naf import NAFPopulation, init_params
        pop = NAFPopulation(2, 8, 2, 200, 200, B, N, 0.01, [-1.0] * 8, [1.0] * 8, [1.0] * 2, 1e-3, seeds=[1, 2])
        pop.set_kernel("mfma")
        for i in range(2):
            pop.set_params(i, init_params(8, 2, 200, 200, 5 + i))
        data = (rng.randn(N, 8), rng.uniform(-1, 1, (N, 2)), rng.randn(N), rng.randn(N, 8), np.full(N, 0.99))
    for i in range(2):
        pop.replay_add_batch(i, *data)
    idx = np.stack([np.stack([rng.choice(N, B, replace=False) for _ in range(3)]) for _ in range(2)]).astype(np.int64)
    kw = {}
    if algo == "sac":
        kw["eps"] = rng.standard_normal((2, 3, B, 1)).astype(np.float32)
    pop.update(3, host_indices=idx, **kw)
    np.save(out, np.stack([pop.get_blob(i, "theta") for i in range(2)] + [pop.get_blob(i, "theta_target") for i in range(2)]))
    pop.close()


if __name__ == "__main__":
    main()
