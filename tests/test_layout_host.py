"""CPU test of the device weight layouts: compiles tests/host_cpp/layout_check.cpp (host code only, against
rlcontrol_amd/csrc/rlc_common.h) with hipcc and runs it -- block index bijection, pack / unpack round trip under
the row-major and the tile-blocked layout, placement of Wc2's action rows, zero padding."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_weight_layout_pack_unpack_and_block_index(tmp_path):
    exe = str(tmp_path / "layout_check")
    src = os.path.join(ROOT, "tests", "host_cpp", "layout_check.cpp")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O1", "-std=c++17",
                           "-I", os.path.join(ROOT, "rlcontrol_amd", "csrc"), src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
