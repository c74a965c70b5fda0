import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_LAUNCHER = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # GPU runs: start the GPU-free launcher process NOW, before any test initialises the GPU in this process (a process
    # that holds the GPU must not exec; tests/clean_launcher.py).  The multi-process GPU tests (torch.distributed.run
    # children over RCCL) are started through it.
    global _LAUNCHER
    expr = (getattr(config.option, "markexpr", "") or "").strip()
    if "gpu" in expr and "not gpu" not in expr:
        import subprocess
        _LAUNCHER = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "clean_launcher.py")],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT)


def pytest_unconfigure(config):
    global _LAUNCHER
    if _LAUNCHER is not None:
        try:
            _LAUNCHER.stdin.close()
            _LAUNCHER.wait(timeout=10)
        except Exception:
            _LAUNCHER.kill()
        _LAUNCHER = None


@pytest.fixture(scope="session")
def clean_launcher():
    """run(argv, env=None, timeout=600) -> {"rc", "stdout", "stderr"} in a process tree that never held the GPU"""
    import json
    if _LAUNCHER is None:
        pytest.skip("the GPU-free launcher is started only for `-m gpu` runs (tests/conftest.py)")

    def run(argv, env=None, timeout=600):
        _LAUNCHER.stdin.write(json.dumps({"argv": list(argv), "env": env or {}, "cwd": ROOT, "timeout": timeout}) + "\n")
        _LAUNCHER.stdin.flush()
        line = _LAUNCHER.stdout.readline()
        if not line:
            raise RuntimeError("clean launcher died")
        return json.loads(line)
    return run


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library; building is __graft_entry__.build()'s job."""
    from rlcontrol_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()
