#!/usr/bin/env python3
"""The reference's own deployment unit -- ONE drop-in agent driven step by step from the host (main.py without
--device_rollout) -- on one MI355X: wall time per environment step, split into env.step, agent.update (replay add +
index draw + fused update launch) and agent.step (acting kernel round trip), for the five agents at the shipped batch
size 32 and at BASELINE's 100, DDPG and the KL agents also in latency mode (json key hip_split).  Writes gpurun_out/<tag>_dropin_latency.json.
    python scripts/dropin_latency.py --tag r02c"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

COMMON = {"norm_type": "input_norm", "buffer_size": 100000, "writer": None, "write_log": False, "write_plot": False,
          "random_seed": 0}
AGENTS = {
    "DDPG": {"exploration_policy": "ou_noise", "shared_l1_dim": 200, "actor_l2_dim": 200, "critic_l2_dim": 200,
             "actor_lr": 1e-3, "critic_lr": 1e-2},
    "SoftActorCritic": {"exploration_policy": "none", "actor_l1_dim": 128, "actor_l2_dim": 128, "critic_l1_dim": 128,
                        "critic_l2_dim": 128, "pi_lr": 1e-3, "qf_vf_lr": 1e-3, "entropy_scale": 0.1,
                        "sample_for_eval": "False", "use_true_q": "False"},
    "NAF": {"exploration_policy": "none", "l1_dim": 200, "l2_dim": 200, "noise_scale": 0.3, "learning_rate": 1e-3},
    "ReverseKL": {"exploration_policy": "none", "actor_l1_dim": 200, "actor_l2_dim": 200, "critic_l1_dim": 200,
                  "critic_l2_dim": 200, "pi_lr": 1e-3, "qf_vf_lr": 1e-3, "entropy_scale": 0.1, "sample_for_eval": "False",
                  "use_true_q": "False", "l_param": 6, "N_param": 64, "optim_type": "intg", "q_update_type": "non_sac"},
}
AGENTS["ForwardKL"] = AGENTS["ReverseKL"]


def measure(name, batch, extra, steps, skip):
    from rlcontrol_amd.environments.environments import create_environment
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.utils.main_utils import create_agent
    env = create_environment({"environment": "Pendulum-v0", "TotalMilSteps": 0.1, "EpisodeSteps": -1,
                              "EvalIntervalMilSteps": 0.0005, "EvalEpisodes": 2})
    cfg = Config()
    cfg.merge_config({"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                      "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                      "action_max": env.action_max})
    cfg.merge_config(dict(COMMON, batch_size=batch, **AGENTS[name]))
    cfg.merge_config(extra)
    agent = create_agent(name, cfg)
    env.set_random_seed(0)
    obs = env.reset()
    agent.reset()
    a = agent.start(obs, True)
    acc = {"env": 0.0, "update": 0.0, "act": 0.0}
    n = 0
    for t in range(steps):
        t0 = time.perf_counter()
        obs_n, r, done, _ = env.step(a)
        t1 = time.perf_counter()
        agent.update(obs, obs_n, float(r), a, done, bool(done))
        t2 = time.perf_counter()
        if done:
            obs = env.reset()
            agent.reset()
            a = agent.start(obs, True)
        else:
            a = agent.step(obs_n, True)
            obs = obs_n
        t3 = time.perf_counter()
        if t >= skip:
            acc["env"] += t1 - t0
            acc["update"] += t2 - t1
            acc["act"] += t3 - t2
            n += 1
    out = {k + "_us": 1e6 * v / n for k, v in acc.items()}
    out["step_us"] = sum(out.values())
    out["s_per_100k_steps"] = out["step_us"] * 0.1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--skip", type=int, default=500)
    ap.add_argument("--tag", default="r02c")
    ap.add_argument("--only", default="", help="comma-separated agent names")
    ap.add_argument("--batches", default="", help="comma-separated batch sizes (default: 32 and 100)")
    ap.add_argument("--no-split", action="store_true", help="skip the latency-mode rows")
    a = ap.parse_args()
    res = {"steps_timed": a.steps - a.skip, "environment": "Pendulum-v0 (host numpy simulator)", "rows": []}
    cases = [(n, b, {}) for n in AGENTS for b in (32, 100)] + [("DDPG", b, {"hip_split": c}) for b, c in ((32, 2), (100, 7))] + \
            [(n, 32, {"hip_split": 8}) for n in ("ReverseKL", "ForwardKL")]
    if a.only:
        cases = [c for c in cases if c[0] in a.only.split(",")]
    if a.batches:
        cases = [c for c in cases if str(c[1]) in a.batches.split(",")]
    if a.no_split:
        cases = [c for c in cases if not c[2]]
    for name, batch, extra in cases:
        r = measure(name, batch, extra, a.steps, a.skip)
        r.update(agent=name, batch=batch, **extra)
        res["rows"].append(r)
        print("%-16s batch %3d %-16s env %5.1f  update %6.1f  act %6.1f  = %6.1f us/step  (%.1f s per 100k steps)" % (
            name, batch, json.dumps(extra) if extra else "", r["env_us"], r["update_us"], r["act_us"], r["step_us"],
            r["s_per_100k_steps"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "%s_dropin_latency.json" % a.tag)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
