"""The build's spill audit (rlcontrol_amd/build.py, profiles/r03_naf_spill_fault.md): the any-shape units must carry
neither whole-wave spills nor exec-0 restore copies, and the policy check refuses a unit that does."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_policy_refuses_the_faulting_signature(monkeypatch):
    from rlcontrol_amd import build as B
    clean = {"k": {"wwm_spills": 0, "exec0_copies": 0, "sgpr_spill_count": 500, "vgpr_spill_count": 40}}
    assert B.check_spill_policy({"naf_generic.o": clean, "ddpg_mfma_7_1.o": clean}) == []
    for bad in ({"wwm_spills": 10, "exec0_copies": 0}, {"wwm_spills": 0, "exec0_copies": 1}):
        usage = {"naf_generic.o": {"rlc_naf_update_kernel": dict(bad, sgpr_spill_count=286, vgpr_spill_count=89)}}
        monkeypatch.delenv("RLC_ALLOW_WWM_SPILLS", raising=False)
        with pytest.raises(RuntimeError, match="rlc_naf_update_kernel"):
            B.check_spill_policy(usage)
        monkeypatch.setenv("RLC_ALLOW_WWM_SPILLS", "1")
        assert len(B.check_spill_policy(usage)) == 1           # reported, not refused, when asked to look anyway
    # the MFMA units are reported, not guarded
    monkeypatch.delenv("RLC_ALLOW_WWM_SPILLS", raising=False)
    assert B.check_spill_policy({"ddpg_mfma_7_1.o": {"k": {"wwm_spills": 3, "exec0_copies": 0}}}) == []


def test_shipped_objects_are_clean():
    """audit of the objects behind the library in the tree (built by __graft_entry__.build())"""
    from rlcontrol_amd import build as B
    checked = 0
    for unit in B.GUARDED_UNITS:
        obj = os.path.join(B.OBJ, unit)
        if not os.path.exists(obj):
            pytest.skip("objects not built here")
        for name, k in B.audit_object(obj).items():
            assert k["wwm_spills"] == 0 and k["exec0_copies"] == 0, (unit, name, k)
            checked += 1
    assert checked >= 8
    usage_json = os.path.join(ROOT, "rlcontrol_amd", "kernel_resource_usage.json")
    if os.path.exists(usage_json):
        with open(usage_json) as f:
            usage = json.load(f)
        assert set(B.GUARDED_UNITS) <= set(usage)


def test_stale_tracking_follows_includes():
    from rlcontrol_amd import build as B
    deps = B._deps(os.path.join(B.CSRC, "ddpg_mfma_inst.hip"))
    names = {os.path.basename(p) for p in deps}
    assert {"ddpg_mfma_inst.hip", "ddpg_mfma_kernel.h", "mfma_blocks.h", "rlc_common.h"} <= names
    assert "kl_mfma_kernel.h" not in names and "naf_mfma_kernel.h" not in names
