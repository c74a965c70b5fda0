#!/usr/bin/env python3
"""Developer loop (CPU box): build ab/<name>.so variants of the BASELINE-shape library that differ in ONE translation
unit's flags -- the timing-only ablations of mfma_blocks.h (RLC_ABLATE bit masks) or any -D switch -- without touching
rlcontrol_amd/librlcontrol_hip.so.  The variants are benched on one GPU box by scripts/ab_run.sh.
    python scripts/ab_ablate.py base abl1=-DRLC_ABLATE=1 wg32=-DRLC_WG_B128 ...
`name` alone = the unmodified tree.  Units recompiled per variant: ddpg_mfma_inst (7,1), sac_mfma_inst (7,1,1),
naf_mfma_inst (7,2,2)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RLC_FAST_BUILD"] = "1"
from rlcontrol_amd import build as B  # noqa: E402

AB = os.path.join(ROOT, "ab")
os.makedirs(os.path.join(AB, "obj"), exist_ok=True)
B.OUT = os.path.join(AB, "_fastbase.so")
B.VARIANT_TAG = B.OUT + ".variant"
B.build()                                   # base objects under csrc/_obj_fast (incremental)

VAR_UNITS = ("ddpg_mfma_7_1.o", "sac_mfma_7_1_1.o", "naf_mfma_7_2_2.o")


def make(spec):
    name, _, flags = spec.partition("=")
    flags, _, units = flags.partition("@")          # name=-DX,-DY@unit1.o+unit2.o  (default units: VAR_UNITS)
    flags = flags.split(",") if flags else []
    units = tuple(units.split("+")) if units else VAR_UNITS
    objs, jobs = [], []
    for src, obj, defs in B._units():
        base = os.path.basename(obj)
        if flags and base in units:
            vobj = os.path.join(AB, "obj", name + "_" + base)
            jobs.append([B._hipcc()] + B.CFLAGS + defs + flags + ["-c", src, "-o", vobj])
            objs.append(vobj)
        else:
            objs.append(obj)
    return name, jobs, objs


specs = [make(s) for s in sys.argv[1:]]
alljobs = [j for _, jobs, _ in specs for j in jobs]
with ThreadPoolExecutor(max_workers=8) as ex:
    list(ex.map(subprocess.check_call, alljobs))
for name, _, objs in specs:
    out = os.path.join(AB, name + ".so")
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)
