"""Host-side mirror vs golden vectors generated FROM THE REFERENCE (tests/golden/make_golden.py).
CPU only; pins quirks Q6, Q7, Q12 and the RNG streams."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

from rlcontrol_amd.utils.config import Config
from rlcontrol_amd.utils.custom_collections import DistinctIndexSampler
from rlcontrol_amd.utils.exploration_policy import OrnsteinUhlenbeckProcess
from rlcontrol_amd.utils.main_utils import get_sweep_parameters
from rlcontrol_amd.utils.running_mean_std import RunningMeanStd


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_sample_n_k_matches_reference_stream(golden_dir):
    cases = _load(golden_dir, "sample_n_k.json")
    assert len(cases) > 50
    for c in cases:
        s = DistinctIndexSampler(c["seed"])
        for call in c["calls"]:
            got = s.sample_n_k(c["n"], c["k"])
            assert got.dtype == np.int64
            assert got.tolist() == call, (c["seed"], c["n"], c["k"])


def test_sample_n_k_errors_like_reference():
    s = DistinctIndexSampler(0)
    with pytest.raises(ValueError):          # custom_collections.py:110-111
        s.sample_n_k(5, 6)
    with pytest.raises(ValueError):
        s.sample_n_k(5, -1)
    assert s.sample_n_k(7, 0).shape == (0,)


def test_sample_n_k_distinct_and_in_range():
    s = DistinctIndexSampler(123)
    for n, k in ((101, 100), (10 ** 6, 100), (400, 128), (33, 32)):
        for _ in range(20):
            idx = s.sample_n_k(n, k)
            assert len(set(idx.tolist())) == k and idx.min() >= 0 and idx.max() < n


def test_ou_noise_matches_reference_stream(golden_dir):
    for c in _load(golden_dir, "ou_noise.json"):
        ou = OrnsteinUhlenbeckProcess(c["seed"], c["dim"], np.array(c["min"]), np.array(c["max"]),
                                      c["theta"], c["mu"], c["sigma"])
        greedy = np.zeros(c["dim"])
        for t, want in enumerate(c["actions"]):
            if t in c["resets_before"]:
                ou.reset()
            got = ou.generate(greedy + 0.5 * np.sin(0.01 * t), t)
            assert np.array_equal(got, np.array(want)), t      # bit-exact float64


def test_ou_first_three_values_quoted_in_survey():
    ou = OrnsteinUhlenbeckProcess(0, 1, [-2.0], [2.0], 0.15, 0.0, 0.2)
    got = [float(ou.generate(np.zeros(1), i)[0]) for i in range(3)]
    assert np.allclose(got, [0.35281047, 0.37992034, 0.51867989], atol=1e-8)


def test_sweep_indexing_first_key_fastest(golden_dir):
    g = _load(golden_dir, "sweep_params.json")
    for name, blk in g.items():
        sweeps = OrderedDict(blk["sweeps"])
        for row in blk["rows"]:
            params, total = get_sweep_parameters(sweeps, row["index"])
            assert total == row["total"]
            assert list(params.items()) == [tuple(kv) for kv in row["params"]]
    ddpg = OrderedDict(g["ddpg"]["sweeps"])
    p0, total = get_sweep_parameters(ddpg, 0)
    assert total == 49 and p0["actor_lr"] == 0.001 and p0["critic_lr"] == 0.01
    p8, _ = get_sweep_parameters(ddpg, 8)
    assert p8["actor_lr"] == 0.005 and p8["critic_lr"] == 0.5


@pytest.mark.parametrize("name", ["ddpg", "sac", "naf"])
def test_shipped_agent_json_equals_reference_sweep(golden_dir, name):
    g = _load(golden_dir, "sweep_params.json")[name]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "jsonfiles", "agent", name + ".json")) as f:
        mine = json.load(f, object_pairs_hook=OrderedDict)
    assert mine["agent"] == g["agent"]
    assert [list(kv) for kv in mine["sweeps"].items()] == g["sweeps"]


def test_config_defaults_and_merge(golden_dir):
    want = _load(golden_dir, "config_defaults.json")
    c = Config()
    assert vars(c) == want
    c.merge_config({"batch_size": 100, "brand_new_key": [1, 2]})
    assert c.batch_size == 100 and c.brand_new_key == [1, 2]


def test_running_mean_std_is_inert_scalar_state(golden_dir):
    """Q6: RunningMeanStd(state_dim) puts state_dim into epsilon; mean/var are scalars 0/1."""
    want = _load(golden_dir, "running_mean_std.json")
    r = RunningMeanStd(3)
    assert float(r.mean) == want["init"]["mean"] and float(r.var) == want["init"]["var"]
    assert float(r.count) == want["init"]["count"] == 3.0
    x = np.array([[0.5, -0.25, 4.0]])
    assert r.normalize(x).tolist() == want["normalize_before_update"]
    r.update(x)
    assert np.allclose(r.mean, want["after"]["mean"]) and np.allclose(r.var, want["after"]["var"])
    assert r.count == want["after"]["count"]


class _FakeStore(object):
    """records what BaseAgent sends over the device boundary (no GPU needed)"""

    def __init__(self, cap):
        self.cap, self.rows = cap, []

    def replay_add(self, agent, s, a, r, s2, g):
        self.rows.append((np.array(s), np.array(a), r, np.array(s2), g))
        if len(self.rows) > self.cap:
            self.rows.pop(0)

    def replay_size(self, agent):
        return len(self.rows)


class _StubManager(object):
    def __init__(self, store):
        from rlcontrol_amd.utils.running_mean_std import RunningMeanStd as R
        self.input_norm = R(3)
        self.store, self.updates, self.norm_updates = store, [], 0
        upd = self.input_norm.update

        def counting(x):
            self.norm_updates += 1
            upd(x)
        self.input_norm.update = counting

    def device_replay(self):
        return (self.store, 0)

    def take_action(self, state, is_train, is_start):
        return np.array([0.25])

    def update_from_replay(self, idx):
        self.updates.append(np.array(idx))

    def reset(self):
        pass


def test_base_agent_insert_rule_and_learn_gate(golden_dir):
    from rlcontrol_amd.agents.base_agent import BaseAgent
    for case in _load(golden_dir, "base_agent_gating.json"):
        cfg = Config()
        cfg.merge_config({"norm_type": "input_norm", "state_dim": 3, "state_min": -np.ones(3),
                          "state_max": np.ones(3), "action_dim": 1, "action_min": [-2.0], "action_max": [2.0],
                          "random_seed": 2, "write_log": False, "write_plot": False, "writer": None,
                          "batch_size": case["batch"], "warmup_steps": case["warmup"], "buffer_size": 16})
        store = _FakeStore(16)
        mgr = _StubManager(store)
        agent = BaseAgent(cfg, mgr)
        rng = np.random.RandomState(9)
        for st in case["steps"]:
            s, s2, a = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), rng.uniform(-2, 2, 1)
            n0 = len(mgr.updates)
            agent.update(s, s2, st["reward"], a, st["terminal"], st["truncated"])
            assert store.replay_size(0) == st["size_after"]
            assert (len(mgr.updates) > n0) == st["learned"]
            assert mgr.norm_updates == st["norm_updates"]
        assert [row[4] for row in store.rows] == case["stored_gamma"]      # gamma or 0.0 at terminals
        assert [row[2] for row in store.rows] == case["stored_reward"]
        # every learn() sampled batch_size distinct in-range indices with the reference's RNG stream
        for idx in mgr.updates:
            assert len(set(idx.tolist())) == case["batch"]


def test_base_agent_warmup_guard_raises():
    from rlcontrol_amd.agents.base_agent import BaseAgent
    cfg = Config()
    cfg.merge_config({"norm_type": "none", "state_dim": 3, "state_min": -np.ones(3), "state_max": np.ones(3),
                      "action_dim": 1, "action_min": [-2.0], "action_max": [2.0], "random_seed": 0,
                      "write_log": False, "write_plot": False, "writer": None, "warmup_steps": 5, "buffer_size": 8})
    agent = BaseAgent(cfg, _StubManager(_FakeStore(8)))
    with pytest.raises(NotImplementedError):       # agents/base_agent.py:42-46
        agent.start(np.zeros(3), True)
