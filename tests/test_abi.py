"""The C-ABI library loads on a CPU-only box, exports every symbol include/rlcontrol_hip.h declares, and
the product fails LOUDLY (no CPU fallback) when no MI355X is present.  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "rlcontrol_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rlc_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib):
    from rlcontrol_amd import _lib
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(hip_lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names          # the Python binding lists exactly the header's functions
    assert hip_lib.rlc_version() >= 100


def test_config_struct_matches_header_layout():
    """ctypes mirror of rlc_ddpg_config: field order and 8-byte pointer alignment as in the header."""
    from rlcontrol_amd._lib import rlc_ddpg_config
    with open(os.path.join(ROOT, "include", "rlcontrol_hip.h")) as f:
        text = f.read()
    body = text[text.index("typedef struct rlc_ddpg_config {"):text.index("} rlc_ddpg_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for nm in decl.split(","):
            fields.append(re.findall(r"([A-Za-z_0-9]+)\s*$", nm.strip())[0])
    assert [f[0] for f in rlc_ddpg_config._fields_] == fields
    assert ctypes.sizeof(rlc_ddpg_config) % 8 == 0
    assert rlc_ddpg_config.state_min.offset % 8 == 0


def test_no_gpu_means_loud_failure_not_fallback(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is for CPU-only boxes")
    from rlcontrol_amd._lib import RlcError
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    with pytest.raises(RlcError):
        DDPGPopulation(1, 3, 1, 200, 200, 200, 100, 1000, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0],
                       1e-3, 1e-2, seeds=[0])
    n = ctypes.c_int(0)
    assert hip_lib.rlc_device_count(ctypes.byref(n)) != 0
    assert len(hip_lib.rlc_last_error()) > 0


def test_product_never_imports_the_oracle():
    """rlcontrol_amd/, main.py must not reference oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "rlcontrol_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                with open(os.path.join(base, fn)) as f:
                    src = f.read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "liboracle" in src:
                    bad.append(os.path.join(base, fn))
    with open(os.path.join(ROOT, "main.py")) as f:
        if re.search(r"^\s*(from|import)\s+oracle\b", f.read(), flags=re.M):
            bad.append("main.py")
    assert not bad, bad


def test_param_layout_matches_oracle_layout():
    from oracle.ddpg import Dims
    from rlcontrol_amd.hip_ddpg import param_layout, init_params
    for dims in ((3, 1, 200, 200, 200), (8, 2, 64, 48, 40)):
        lay_o, P_o = Dims(*dims).layout()
        lay_p, P_p = param_layout(*dims)
        assert P_o == P_p and list(lay_o.items()) == list(lay_p.items())
    assert param_layout(3, 1, 200, 200, 200)[1] == 81802
    th = init_params(3, 1, 200, 200, 200, 0)
    lay, _ = param_layout(3, 1, 200, 200, 200)
    off, shp = lay["Wc3"]
    assert np.max(np.abs(th[off:off + 200])) <= 3e-3               # output layers U(+-3e-3)
    off, shp = lay["W1"]
    assert np.max(np.abs(th[off:off + 600])) <= 1.0 and np.max(np.abs(th[off:off + 600])) > 0.9   # sqrt(3/3)
