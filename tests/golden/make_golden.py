#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
numpy-only modules (imported from /root/reference, which exists only in the build
container -- never on the GPU box).  Only inputs and outputs are recorded; no
reference source text is copied.

Run once in the build container:  python tests/golden/make_golden.py

Modules exercised (reference file:line):
  utils/main_utils.py:92-99          get_sweep_parameters
  utils/custom_collections.py:6-131  RandomAccessQueue (+ sample_n_k)
  utils/replaybuffer.py:14-42        ReplayBuffer
  utils/exploration_policy.py:4-24   OrnsteinUhlenbeckProcess
  agents/base_agent.py:7-74          BaseAgent insert rule / learn() gate
  utils/running_mean_std.py:2-35     RunningMeanStd (quirk Q6)
  utils/config.py:1-27               Config defaults
  Bimodal1DEnv_trueQ_ckpt/*uneq_var1*.data-00000-of-00001  critic known-answer weights

TensorFlow-dependent modules cannot be imported (tensorflow 1.15 absent), so the
network math has no reference-run vectors: see DESIGN.md "parity unpinned at the
TF boundary".
"""
import json
import os
import sys
from collections import OrderedDict

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

from utils.custom_collections import RandomAccessQueue  # noqa: E402
from utils.replaybuffer import ReplayBuffer  # noqa: E402
from utils.exploration_policy import OrnsteinUhlenbeckProcess  # noqa: E402
from utils.main_utils import get_sweep_parameters  # noqa: E402
from utils.config import Config  # noqa: E402
from utils.running_mean_std import RunningMeanStd  # noqa: E402
from agents.base_agent import BaseAgent  # noqa: E402


def sweep_vectors():
    out = {}
    for name in ("ddpg", "sac", "naf"):
        with open(os.path.join(REF, "jsonfiles/agent/%s.json" % name)) as f:
            js = json.load(f, object_pairs_hook=OrderedDict)
        rows = []
        for idx in (0, 1, 2, 6, 7, 8, 13, 48, 49, 50, 97, 98, 343):
            params, total = get_sweep_parameters(js["sweeps"], idx)
            rows.append({"index": idx, "total": total, "params": list(params.items())})
        out[name] = {"agent": js["agent"], "sweeps": list(js["sweeps"].items()), "rows": rows}
    return out


def sample_n_k_vectors():
    cases = []
    for seed in (0, 1, 7):
        for (n, k) in ((101, 100), (250, 100), (300, 100), (301, 100), (1000, 100), (10 ** 6, 100),
                       (33, 32), (96, 32), (97, 32), (5000, 32), (5, 5), (7, 0), (40, 13)):
            q = RandomAccessQueue(maxlen=None, seed=seed)
            calls = [q.sample_n_k(n, k).astype(np.int64).tolist() for _ in range(5)]
            cases.append({"seed": seed, "n": n, "k": k, "calls": calls})
    # a case that forces the "ran out of spares -> refill" branch (custom_collections.py:124-127)
    # small n just above 3k so collisions are frequent
    for seed in range(40):
        q = RandomAccessQueue(maxlen=None, seed=seed)
        calls = [q.sample_n_k(31, 10).astype(np.int64).tolist() for _ in range(20)]
        cases.append({"seed": seed, "n": 31, "k": 10, "calls": calls})
    return cases


def replay_roundtrip():
    """capacity-8 buffer, 21 adds (two flips of the two-list queue), sample after each phase"""
    rb = ReplayBuffer(8, 3)
    rng = np.random.RandomState(123)
    log = []
    for t in range(21):
        s = rng.uniform(-1, 1, 3)
        a = rng.uniform(-2, 2, 1)
        r = float(rng.uniform(-16, 0))
        s2 = rng.uniform(-1, 1, 3)
        g = 0.0 if t % 5 == 4 else 0.99
        rb.add(s, a, r, s2, g)
        entry = {"t": t, "add": [s.tolist(), a.tolist(), r, s2.tolist(), g], "size": rb.get_size()}
        # logical content oldest -> newest after the add
        entry["content_reward"] = [tr.reward for tr in rb.buffer]
        if rb.get_size() >= 4 and t % 3 == 0:
            st, ac, rw, ns, gm = rb.sample_batch(4)
            entry["sample"] = {
                "state": st.tolist(), "action": ac.tolist(), "reward": rw.tolist(),
                "next_state": ns.tolist(), "gamma": gm.tolist(),
                "dtypes": [str(x.dtype) for x in (st, ac, rw, ns, gm)],
                "shapes": [list(x.shape) for x in (st, ac, rw, ns, gm)],
            }
        log.append(entry)
    return log


def ou_vectors():
    out = []
    for seed, dim, amin, amax in ((0, 1, [-2.0], [2.0]), (5, 2, [-1.0, -1.0], [1.0, 1.0])):
        ou = OrnsteinUhlenbeckProcess(seed, dim, np.array(amin), np.array(amax), 0.15, 0.0, 0.2)
        greedy = np.zeros(dim)
        acts = []
        for t in range(1000):
            if t in (200, 201, 640):
                ou.reset()
            g = greedy + 0.5 * np.sin(0.01 * t)
            acts.append(ou.generate(g, t).tolist())
        out.append({"seed": seed, "dim": dim, "min": amin, "max": amax, "theta": 0.15, "mu": 0.0,
                    "sigma": 0.2, "resets_before": [200, 201, 640], "greedy": "0.5*sin(0.01*t)",
                    "actions": acts})
    return out


class _StubNorm(object):
    def __init__(self):
        self.n = 0

    def update(self, x):
        self.n += 1


class _StubManager(object):
    def __init__(self):
        self.input_norm = _StubNorm()
        self.calls = []
        self.resets = 0

    def take_action(self, state, is_train, is_start):
        return np.array([0.25])

    def update_network(self, s, a, s2, r, g):
        self.calls.append({"shapes": [list(np.shape(x)) for x in (s, a, s2, r, g)],
                           "reward": np.asarray(r).tolist(), "gamma": np.asarray(g).tolist()})

    def reset(self):
        self.resets += 1


def base_agent_gating():
    out = []
    for warmup, batch in ((0, 4), (6, 4), (0, 1)):
        cfg = Config()
        cfg.merge_config({"norm_type": "input_norm", "state_dim": 3, "state_min": -np.ones(3),
                          "state_max": np.ones(3), "action_dim": 1, "action_min": [-2.0],
                          "action_max": [2.0], "random_seed": 2, "write_log": False,
                          "write_plot": False, "writer": None, "batch_size": batch,
                          "warmup_steps": warmup, "buffer_size": 16})
        mgr = _StubManager()
        agent = BaseAgent(cfg, mgr)
        rng = np.random.RandomState(9)
        script = [(False, False)] * 3 + [(True, False), (False, False), (True, True), (False, False),
                                         (False, False), (True, False), (False, False), (True, True)]
        steps = []
        for t, (term, trunc) in enumerate(script):
            s = rng.uniform(-1, 1, 3)
            s2 = rng.uniform(-1, 1, 3)
            a = rng.uniform(-2, 2, 1)
            r = float(-t - 0.5)
            ncalls = len(mgr.calls)
            agent.update(s, s2, r, a, term, trunc)
            steps.append({"t": t, "terminal": term, "truncated": trunc, "reward": r,
                          "size_after": agent.replay_buffer.get_size(),
                          "learned": len(mgr.calls) > ncalls,
                          "norm_updates": mgr.input_norm.n})
        out.append({"warmup": warmup, "batch": batch, "steps": steps, "calls": mgr.calls,
                    "stored_gamma": [tr.transition_gamma for tr in agent.replay_buffer.buffer],
                    "stored_reward": [tr.reward for tr in agent.replay_buffer.buffer]})
    return out


def config_defaults():
    c = Config()
    return {k: v for k, v in vars(c).items()}


def rms_state():
    r = RunningMeanStd(3)
    before = {"mean": float(r.mean), "var": float(r.var), "count": float(r.count)}
    x = np.array([[0.5, -0.25, 4.0]])
    norm0 = r.normalize(x).tolist()
    r.update(x)
    return {"init": before, "normalize_before_update": norm0,
            "after": {"mean": np.asarray(r.mean).tolist(), "var": np.asarray(r.var).tolist(),
                      "count": float(r.count)}}


def bimodal_ckpt():
    """main/qf critic weights of the uneq_var1 checkpoint.  The .data file is raw little-endian
    fp32, tensors concatenated in lexicographic key order:
    beta1_power, beta2_power, then for fully_connected{,_1,_2}/{biases,weights}: value, Adam, Adam_1."""
    d = np.fromfile(os.path.join(REF, "Bimodal1DEnv_trueQ_ckpt",
                                 "Bimodal1DEnv_uneq_var1_trueQ_learned.data-00000-of-00001"), dtype="<f4")
    assert d.size == 123005
    pos = 2
    shapes = [("b1", (200,)), ("W1", (1, 200)), ("b2", (200,)), ("W2", (201, 200)),
              ("b3", (1,)), ("W3", (200, 1))]
    out = {"beta1_power": d[0], "beta2_power": d[1]}
    for name, shp in shapes:
        n = int(np.prod(shp))
        # order inside one variable: value, Adam (m), Adam_1 (v)
        out[name] = d[pos:pos + n].reshape(shp).copy()
        out[name + "_m"] = d[pos + n:pos + 2 * n].reshape(shp).copy()
        out[name + "_v"] = d[pos + 2 * n:pos + 3 * n].reshape(shp).copy()
        pos += 3 * n
    assert pos == d.size
    return out


# reward functions of the five Bimodal1D environments the checkpoints were fitted to: (maximum 1, maximum 2, height 1,
# height 2, stddev 1, stddev 2), restated from environments/environments.py (eq_var1 :573-587, eq_var2 :660-673,
# eq_var3 :747-760, uneq_var1 :312-325, uneq_var2 :399-412); reward(a) = h1 exp(-.5((a-m1)/s1)^2) + h2 exp(-.5((a-m2)/s2)^2)
BIMODAL_REWARDS = {
    "eq_var1": (-0.6, 0.6, 1.0, 1.0, 0.2, 0.2),
    "eq_var2": (-0.8, 0.8, 1.0, 1.0, 0.2, 0.2),
    "eq_var3": (-1.0, 1.0, 1.0, 1.0, 0.2, 0.2),
    "uneq_var1": (-1.0, 1.0, 1.0, 1.5, 0.4, 0.2),
    "uneq_var2": (-1.0, 1.0, 1.0, 1.5, 0.3, 0.1),
}
CKPT_SHAPES = [("b1", (200,)), ("W1", (1, 200)), ("b2", (200,)), ("W2", (201, 200)), ("b3", (1,)), ("W3", (200, 1))]


def bimodal_ckpts_all():
    """Every Bimodal1DEnv_trueQ_ckpt/*.data file (SAC main/qf, 10 000 Adam steps at batch 32): weights of all five,
    the optimizer's beta powers, and the Adam slots m / v in full (they pin the value | Adam | Adam_1 slot order and the
    [in, out] / action-as-last-row layout through their exact-zero pattern).  Weights-only parse of raw fp32."""
    out = {}
    for name, rew in BIMODAL_REWARDS.items():
        d = np.fromfile(os.path.join(REF, "Bimodal1DEnv_trueQ_ckpt",
                                     "Bimodal1DEnv_%s_trueQ_learned.data-00000-of-00001" % name), dtype="<f4")
        assert d.size == 123005
        out[name + "/beta_powers"] = d[:2].copy()
        out[name + "/reward"] = np.asarray(rew, np.float64)
        pos = 2
        for tn, shp in CKPT_SHAPES:
            n = int(np.prod(shp))
            out["%s/%s" % (name, tn)] = d[pos:pos + n].reshape(shp).copy()
            out["%s/%s_m" % (name, tn)] = d[pos + n:pos + 2 * n].reshape(shp).copy()
            out["%s/%s_v" % (name, tn)] = d[pos + 2 * n:pos + 3 * n].reshape(shp).copy()
            pos += 3 * n
        assert pos == d.size
    return out


def main():
    with open(os.path.join(HERE, "sweep_params.json"), "w") as f:
        json.dump(sweep_vectors(), f, indent=1)
    with open(os.path.join(HERE, "sample_n_k.json"), "w") as f:
        json.dump(sample_n_k_vectors(), f)
    with open(os.path.join(HERE, "replay_roundtrip.json"), "w") as f:
        json.dump(replay_roundtrip(), f)
    with open(os.path.join(HERE, "ou_noise.json"), "w") as f:
        json.dump(ou_vectors(), f)
    with open(os.path.join(HERE, "base_agent_gating.json"), "w") as f:
        json.dump(base_agent_gating(), f)
    with open(os.path.join(HERE, "config_defaults.json"), "w") as f:
        json.dump(config_defaults(), f, indent=1)
    with open(os.path.join(HERE, "running_mean_std.json"), "w") as f:
        json.dump(rms_state(), f, indent=1)
    ck = bimodal_ckpt()
    np.savez_compressed(os.path.join(HERE, "bimodal_uneq_var1_qf.npz"), **ck)
    np.savez_compressed(os.path.join(HERE, "bimodal_qf_ckpts.npz"), **bimodal_ckpts_all())
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
