"""Build librlcontrol_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with the tree).

Every translation unit is compiled to an object under csrc/_obj/ (in parallel, skipped when newer than
all sources/headers) and linked into rlcontrol_amd/librlcontrol_hip.so.  The MFMA kernel template is
instantiated once per (M tiles, action dim) pair, each in its own object.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# RLC_STAMPS=1: diagnostic build with in-kernel phase stamps (never the shipped library)
STAMPS = os.environ.get("RLC_STAMPS", "0") == "1"
OBJ = os.path.join(CSRC, ("_obj_stamps" if STAMPS else "_obj") + ("_fast" if os.environ.get("RLC_FAST_BUILD", "0") == "1" else ""))
OUT = os.path.join(_HERE, "librlcontrol_hip_stamps.so" if STAMPS else "librlcontrol_hip.so")
PLAIN = ("rlc_api.hip", "rlc_api_sac.hip", "rlc_api_naf.hip", "replay_kernels.hip", "ddpg_generic.hip", "ddpg_mfma.hip", "sac_generic.hip", "sac_mfma.hip", "naf_generic.hip", "naf_mfma.hip", "rollout_kernels.hip", "rlc_api_rollout.hip", "kl_generic.hip", "rlc_api_kl.hip", "ddpg_split.hip", "kl_mfma.hip")
MFMA_VARIANTS = [(mt, ad) for ad in (1, 2) for mt in (2, 4, 7, 8)]
FAST = os.environ.get("RLC_FAST_BUILD", "0") == "1"     # developer loop: only the headline shape
SAC_VARIANTS = [(mt, ntw, ad) for ad in (1, 2) for ntw in (1, 2) for mt in (2, 4, 7, 8)]
if FAST:
    MFMA_VARIANTS = [(7, 1)]
    SAC_VARIANTS = [(7, 1, 1)]
SPLIT_VARIANTS = [(2, 1)] if FAST else [(mt, ad) for ad in (1, 2) for mt in (1, 2, 4)]
NAF_VARIANTS = [(7, 2, 2)] if FAST else [(mt, ntw, ad) for ad in (1, 2) for ntw in (1, 2) for mt in (2, 4, 7, 8)]
# -fgpu-flush-denormals-to-zero: TF-1.15's CPU kernels flush denormals (the reference's checkpoints show it: beta1
# power exactly 0, idle Adam m slots resting at 9..10 x FLT_MIN; tests/test_ckpt_pins.py) -- the kernels' fp32 VALU
# arithmetic runs in the same mode
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-Wno-unused-result",
          "-fgpu-flush-denormals-to-zero"] + (["-DRLC_STAMPS"] if STAMPS else [])
if os.environ.get("RLC_FAST_BUILD", "0") == "1":
    CFLAGS.append("-DRLC_ONLY_7_1")
CFLAGS += os.environ.get("RLC_EXTRA_CFLAGS", "").split()      # developer loop: A/B switches (-DRLC_...)


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(_HERE, "..", "include", "rlcontrol_hip.h"))
    return hs


def _units():
    units = [(os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o")), []) for s in PLAIN]
    for mt, ad in MFMA_VARIANTS:
        units.append((os.path.join(CSRC, "ddpg_mfma_inst.hip"), os.path.join(OBJ, "ddpg_mfma_%d_%d.o" % (mt, ad)),
                      ["-DRLC_MT=%d" % mt, "-DRLC_AD=%d" % ad]))
    for mt, ad in SPLIT_VARIANTS:
        units.append((os.path.join(CSRC, "ddpg_split_inst.hip"), os.path.join(OBJ, "ddpg_split_%d_%d.o" % (mt, ad)),
                      ["-DRLC_MT=%d" % mt, "-DRLC_AD=%d" % ad]))
    for mt, ntw, ad in SAC_VARIANTS:
        units.append((os.path.join(CSRC, "sac_mfma_inst.hip"), os.path.join(OBJ, "sac_mfma_%d_%d_%d.o" % (mt, ntw, ad)),
                      ["-DRLC_MT=%d" % mt, "-DRLC_NTW=%d" % ntw, "-DRLC_AD=%d" % ad]))
    for mt, ntw, ad in NAF_VARIANTS:
        units.append((os.path.join(CSRC, "naf_mfma_inst.hip"), os.path.join(OBJ, "naf_mfma_%d_%d_%d.o" % (mt, ntw, ad)),
                      ["-DRLC_MT=%d" % mt, "-DRLC_NTW=%d" % ntw, "-DRLC_AD=%d" % ad]))
    return units


def _stale(src, obj, hdr_time):
    return (not os.path.exists(obj)) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time)


VARIANT_TAG = OUT + ".variant"
VARIANT = "fast" if FAST else "full"


def needs_build():
    if not os.path.exists(OUT):
        return True
    try:
        if open(VARIANT_TAG).read().strip() != VARIANT:
            return True
    except OSError:
        return True
    t = os.path.getmtime(OUT)
    srcs = [u[0] for u in _units()] + _headers()
    return any(os.path.getmtime(f) > t for f in srcs)


def build(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    todo = [u for u in _units() if force or _stale(u[0], u[1], hdr_time)]

    def compile_one(u):
        src, obj, defs = u
        cmd = [_hipcc()] + CFLAGS + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [u[1] for u in _units()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(VARIANT_TAG, "w") as f:
        f.write(VARIANT)
    return OUT


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
