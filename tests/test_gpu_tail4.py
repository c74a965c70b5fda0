"""The tail-of-four kernels (mfma_blocks.h, Blk's T4: the four-row last batch tile on v_mfma_f32_4x4x1) against the padded
kernels they replace: same weights, same minibatches, three updates -- the parameters agree to rounding (only the
summation order of rows 96..99 differs).  Both run in fresh processes (the switch RLC_NO_TAIL4 is read once)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "tail4_dump.py")


@pytest.mark.gpu
@pytest.mark.parametrize("algo,B", [("ddpg", 100), ("ddpg", 98), ("sac", 100), ("naf", 97)])
def test_tail_of_four_kernels_match_the_padded_kernels(clean_launcher, tmp_path, algo, B):
    outs = []
    for no_t4 in ("0", "1"):
        out = str(tmp_path / ("theta_%s.npy" % no_t4))
        rep = clean_launcher([sys.executable, HELPER, algo, str(B), out], env={"RLC_NO_TAIL4": no_t4}, timeout=300)
        assert rep["rc"] == 0, rep["stderr"][-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert a.shape == b.shape and np.all(np.isfinite(a))
    assert not np.array_equal(a, b)                       # two different kernels did run
    # Adam normalises the step: a parameter whose gradient is rounding noise may step +lr in one run and -lr in the
    # other, so a handful of elements differ by up to 2 * lr per update; everything else agrees to rounding
    diff, scale = np.abs(a - b), np.abs(b).max()
    assert np.mean(diff <= 2e-5 * scale) > 0.999, float(np.mean(diff <= 2e-5 * scale))
    assert diff.max() <= 3 * 2 * 1e-2
