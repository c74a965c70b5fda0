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
    # seven-tile shapes also in their tail-of-four form (mfma_blocks.h, Blk's T4: batch 97..100)
    t4 = lambda mt: ((0, ""), (1, "_t4")) if mt == 7 else ((0, ""),)
    for mt, ad in MFMA_VARIANTS:
        for flag, tag in t4(mt):
            units.append((os.path.join(CSRC, "ddpg_mfma_inst.hip"), os.path.join(OBJ, "ddpg_mfma_%d_%d%s.o" % (mt, ad, tag)),
                          ["-DRLC_MT=%d" % mt, "-DRLC_AD=%d" % ad, "-DRLC_T4=%d" % flag]))
    for mt, ad in SPLIT_VARIANTS:
        units.append((os.path.join(CSRC, "ddpg_split_inst.hip"), os.path.join(OBJ, "ddpg_split_%d_%d.o" % (mt, ad)),
                      ["-DRLC_MT=%d" % mt, "-DRLC_AD=%d" % ad]))
    for mt, ntw, ad in SAC_VARIANTS:
        for flag, tag in t4(mt):
            units.append((os.path.join(CSRC, "sac_mfma_inst.hip"), os.path.join(OBJ, "sac_mfma_%d_%d_%d%s.o" % (mt, ntw, ad, tag)),
                          ["-DRLC_MT=%d" % mt, "-DRLC_NTW=%d" % ntw, "-DRLC_AD=%d" % ad, "-DRLC_T4=%d" % flag]))
    for mt, ntw, ad in NAF_VARIANTS:
        for flag, tag in t4(mt):
            units.append((os.path.join(CSRC, "naf_mfma_inst.hip"), os.path.join(OBJ, "naf_mfma_%d_%d_%d%s.o" % (mt, ntw, ad, tag)),
                          ["-DRLC_MT=%d" % mt, "-DRLC_NTW=%d" % ntw, "-DRLC_AD=%d" % ad, "-DRLC_T4=%d" % flag]))
    return units


_LLVM = "/opt/rocm/lib/llvm/bin"
# Units whose kernels must not carry whole-wave spills (see audit_object); the MFMA kernels are reported, not refused:
# their SGPR spills sit at phase boundaries and they run two waves per SIMD with the full 256-register budget.
GUARDED_UNITS = ("ddpg_generic.o", "sac_generic.o", "naf_generic.o", "kl_generic.o")
USAGE_JSON = os.path.join(_HERE, "kernel_resource_usage.json")


def audit_object(obj):
    """Per-kernel register / spill report of one compiled unit, read back from the gfx950 code object inside `obj`:
    vgpr_count, sgpr_spill_count, vgpr_spill_count, scratch bytes (the code object's metadata = what
    -Rpass-analysis=kernel-resource-usage prints), `wwm_spills`: the number of whole-wave scratch stores / loads
    (`s_or_saveexec_b64 s[..], -1` + `scratch_store/load`), i.e. VGPRs that hold spilled SGPRs in their lanes and are
    themselves spilled to scratch, and `exec0_copies`: register-allocator split copies placed in front of the exec
    restore of a block that a divergent loop leaves with exec == 0.  The second is what hipcc 7.2 did to naf_generic.hip
    in round 2 (a 64-bit index kept a stale high word -> HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION), the first is the
    register-pressure regime it happened in (profiles/r03_naf_spill_fault.md).  Nothing in the source shows either: the
    build has to watch them."""
    import re
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, "x.hipfb"), os.path.join(tmp, "x.co")
        # (an explicit output file: with one positional argument llvm-objcopy rewrites `obj` in place, and the library
        # then looks older than its objects)
        if subprocess.call([os.path.join(_LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj,
                            os.path.join(tmp, "x.o")], stderr=subprocess.DEVNULL) != 0:
            return {}                     # a host-only unit: no device code in it
        subprocess.check_call([os.path.join(_LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fb,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(_LLVM, "llvm-readelf"), "--notes", co], text=True)
        dis = subprocess.check_output([os.path.join(_LLVM, "llvm-objdump"), "-d", "--symbolize-operands",
                                       "--no-show-raw-insn", co], text=True)
    kernels, cur = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.match(r"\s*\.(private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count|vgpr_count|sgpr_count|"
                     r"group_segment_fixed_size):\s+(\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    lines = dis.splitlines()
    execz_targets = set(m.group(1) for m in (re.search(r"s_cbranch_execz\s+(L\d+)", l) for l in lines) if m)
    cur_k, prev_saveexec = None, False
    for i, line in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m and not re.match(r"L\d+$", m.group(1)):
            cur_k = m.group(1) if m.group(1) in kernels else None
            prev_saveexec = False
            continue
        if cur_k is None:
            continue
        if m:
            # a local label: if an s_cbranch_execz lands here (exec == 0 on that edge), are there VGPR moves in front of
            # the block's `s_or_b64 exec, exec, ...`?  (the round-2 NAF miscompile: profiles/r03_naf_spill_fault.md)
            if m.group(1) in execz_targets:
                moves, j = 0, i + 1
                while j < len(lines):
                    ins = lines[j].strip().split("//")[0].strip()
                    if re.match(r"v_mov_b(32|64)|v_accvgpr", ins):
                        moves += 1
                    elif re.match(r"s_or_b64 exec, exec,", ins):
                        if moves:
                            kernels[cur_k]["exec0_copies"] = kernels[cur_k].get("exec0_copies", 0) + 1
                        break
                    elif not re.match(r"s_(waitcnt|nop|mov_b32|mov_b64 s)", ins):
                        break                     # anything else: an ordinary block, not a bare restore-then-reconverge
                    j += 1
            continue
        ins = line.strip()
        if prev_saveexec and ins.startswith("scratch_"):
            kernels[cur_k]["wwm_spills"] = kernels[cur_k].get("wwm_spills", 0) + 1
        prev_saveexec = bool(re.match(r"s_or_saveexec_b64 s\[\d+:\d+\], -1", ins))
    for k in kernels.values():
        k.setdefault("wwm_spills", 0)
        k.setdefault("exec0_copies", 0)
    return kernels


def check_spill_policy(usage):
    """usage: {unit basename: audit_object(...)}.  Refuse whole-wave spills and exec-0 restore copies in the guarded
    units."""
    bad = ["%s: %s (%d whole-wave spill ops, %d exec-0 restore copies; %d SGPR + %d VGPR spills)" % (
               u, k, r["wwm_spills"], r.get("exec0_copies", 0), r.get("sgpr_spill_count", 0), r.get("vgpr_spill_count", 0))
           for u, ks in sorted(usage.items()) if u in GUARDED_UNITS for k, r in sorted(ks.items())
           if r["wwm_spills"] > 0 or r.get("exec0_copies", 0) > 0]
    if bad and os.environ.get("RLC_ALLOW_WWM_SPILLS", "0") != "1":
        raise RuntimeError("kernels that spill SGPR-spill VGPRs to scratch (the combination hipcc 7.2 miscompiled, "
                           "DESIGN.md 5.5) -- reduce their register pressure or set RLC_ALLOW_WWM_SPILLS=1 to look at "
                           "them anyway:\n  " + "\n  ".join(bad))
    return bad


_INC_CACHE = {}


def _deps(path):
    """`path` and every header under csrc/ or include/ it includes, transitively (plain `#include "x.h"` scan)"""
    import re
    path = os.path.abspath(path)
    if path in _INC_CACHE:
        return _INC_CACHE[path]
    _INC_CACHE[path] = out = {path}
    try:
        text = open(path).read()
    except OSError:
        return out
    for name in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
        for d in (os.path.dirname(path), CSRC, os.path.join(_HERE, "..", "include")):
            cand = os.path.abspath(os.path.join(d, name))
            if os.path.exists(cand):
                out |= _deps(cand)
                break
    return out


def _stale(src, obj, hdr_time=None):
    """an object is stale when it is older than its source or any header that source includes (transitively)"""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(f) > t for f in _deps(src))


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
    # an object older than one of its sources, or the library older than an object (an edit made while a build ran)
    return any(_stale(u[0], u[1]) or os.path.getmtime(u[1]) > t for u in _units())


def build(force=False, verbose=False, jobs=None):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    todo = [u for u in _units() if force or _stale(u[0], u[1], hdr_time)]

    def compile_one(u):
        import json
        src, obj, defs = u
        cmd = [_hipcc()] + CFLAGS + defs + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        with open(obj + ".usage.json", "w") as f:
            json.dump(audit_object(obj), f)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    import json
    usage = {}
    for _, obj, _ in _units():
        if not os.path.exists(obj + ".usage.json"):          # an object of an older build
            with open(obj + ".usage.json", "w") as f:
                json.dump(audit_object(obj), f)
        with open(obj + ".usage.json") as f:
            usage[os.path.basename(obj)] = json.load(f)
    check_spill_policy(usage)                                # before the link: a refused build leaves no library
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [u[1] for u in _units()]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    if not STAMPS and not FAST:
        with open(USAGE_JSON, "w") as f:
            json.dump(usage, f, indent=1, sort_keys=True)
    with open(VARIANT_TAG, "w") as f:
        f.write(VARIANT)
    return OUT


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
