"""Agent-level parity on the GPU: the drop-in DDPG agent (start/step/update/reset through the C ABI) against
the reference-structured CPU agent (oracle/cpu_baseline.py) driven by the SAME Pendulum episodes, seeds and
initial weights.  Same replay contents, same sampled indices (reference RNG stream), same OU noise stream;
actions must agree while rounding differences have not yet compounded through Adam."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _config(seed, batch, sampler="reference"):
    from rlcontrol_amd.utils.config import Config
    from rlcontrol_amd.environments.environments import create_environment
    env = create_environment({"environment": "Pendulum-v0", "TotalMilSteps": 0.001, "EpisodeSteps": -1,
                              "EvalIntervalMilSteps": 0.0005, "EvalEpisodes": 2})
    cfg = Config()
    cfg.merge_config({"env_name": env.name, "state_dim": env.state_dim, "state_min": env.state_min,
                      "state_max": env.state_max, "action_dim": env.action_dim, "action_min": env.action_min,
                      "action_max": env.action_max})
    cfg.merge_config({"norm_type": "input_norm", "exploration_policy": "ou_noise", "shared_l1_dim": 200,
                      "actor_l2_dim": 200, "critic_l2_dim": 200, "actor_lr": 0.001, "critic_lr": 0.01,
                      "batch_size": batch, "buffer_size": 5000, "writer": None, "replay_sampler": sampler})
    cfg.merge_config({"write_log": False, "write_plot": False, "random_seed": seed})
    return cfg, env


@pytest.mark.parametrize("batch", [32, 100])
def test_dropin_agent_tracks_reference_structured_agent(hip_lib, batch):
    from rlcontrol_amd.utils.main_utils import create_agent
    from oracle.cpu_baseline import CpuDDPGAgent
    seed = 1
    cfg, env = _config(seed, batch)
    gpu = create_agent("DDPG", cfg)
    cpu = CpuDDPGAgent(3, 1, 200, 200, 200, batch, 5000, cfg.gamma, cfg.tau, 0.001, 0.01, env.state_min,
                       env.state_max, env.action_min, env.action_max, seed)
    assert np.array_equal(gpu.network_manager.population.get_blob(0, "theta"), cpu.net.theta)
    env.set_random_seed(seed)
    steps, n_upd, worst = 0, 0, 0.0
    for ep in range(2):
        obs = env.reset()
        gpu.reset(); cpu.reset()
        a_g, a_c = gpu.start(obs, True), cpu.start(obs, True)
        for t in range(200):
            worst = max(worst, float(np.max(np.abs(a_g - a_c))))
            assert np.allclose(a_g, a_c, atol=2e-4), (ep, t, a_g, a_c)
            obs_n, r, done, _ = env.step(a_c)           # both agents see the oracle agent's trajectory
            trunc = bool(done and t == 199)
            gpu.update(obs, obs_n, float(r), a_c, done, trunc)
            cpu.update(obs, obs_n, float(r), a_c, done, trunc)
            steps += 1
            if cpu.last_idx is not None:
                n_upd = cpu.n_updates
            if done:
                break
            a_g, a_c = gpu.step(obs_n, True), cpu.step(obs_n, True)
            obs = obs_n
        assert gpu.replay_buffer.get_size() == len(cpu.replay) == min(steps - (ep + 1), 5000)   # Q7: 1 truncated/ep
    assert n_upd == steps - batch                         # first update at size batch+1 (Q12); truncated steps still learn (Q7)
    # greedy (eval) actions after ~300 updates still agree
    probe = np.array([[1.0, 0.0, 0.5], [-1.0, 0.0, -2.0], [0.0, 1.0, 4.0]])
    got = np.stack([gpu.start(p, False) for p in probe])
    want = np.stack([cpu.start(p, False) for p in probe])
    assert np.allclose(got, want, atol=5e-4)


def test_dropin_agent_manager_level_update_network(hip_lib):
    """update_network(state, action, next_state, reward, gamma) with host arrays (the manager boundary)."""
    from rlcontrol_amd.utils.main_utils import create_agent
    from oracle.ddpg import DDPGOracle, Dims
    cfg, env = _config(4, 32)
    agent = create_agent("DDPG", cfg)
    mgr = agent.network_manager
    th = mgr.population.get_blob(0, "theta")
    o = DDPGOracle(Dims(3, 1, 200, 200, 200), th, 0.001, 0.01, cfg.tau, env.state_min, env.state_max, env.action_max)
    rng = np.random.RandomState(0)
    for _ in range(3):
        s, s2 = rng.uniform(-1, 1, (32, 3)), rng.uniform(-1, 1, (32, 3))
        a, r, g = rng.uniform(-2, 2, (32, 1)), rng.uniform(-16, 0, 32), np.full(32, 0.99)
        mgr.update_network(s, a, s2, r, g)
        taps = o.update(s, a, s2, r, g, taps=True)
        assert np.allclose(mgr.population.last_tap(0, "q"), taps["q"], rtol=0, atol=2e-5 * np.abs(taps["q"]).max())
    assert mgr.input_norm is not None and float(mgr.input_norm.mean) == 0.0     # Q6: inert


def test_device_sampler_mode_runs(hip_lib):
    from rlcontrol_amd.utils.main_utils import create_agent
    cfg, env = _config(2, 32, sampler="device")
    agent = create_agent("DDPG", cfg)
    env.set_random_seed(2)
    obs = env.reset()
    agent.reset()
    a = agent.start(obs, True)
    for t in range(60):
        obs_n, r, done, _ = env.step(a)
        agent.update(obs, obs_n, float(r), a, done, False)
        a = agent.step(obs_n, True)
        obs = obs_n
    assert agent.replay_buffer.get_size() == 60 and np.all(np.isfinite(a))


def _drive(agent, env, seed, steps, queued):
    """the step loop of experiment.py:105-135; returns every action the agent chose"""
    agent.network_manager.queues_next_action = queued
    env.set_random_seed(seed)
    obs = env.reset()
    agent.reset()
    a = agent.start(obs, True)
    out, fetched = [a.copy()], 0
    for t in range(steps):
        obs_n, r, done, _ = env.step(a)
        agent.update(obs, obs_n, float(r), a, done, False)
        if queued and agent.network_manager._queued_state is not None:
            fetched += 1
        if t % 17 == 5:                                  # an evaluation episode's first action between update and step
            agent.start(np.array([0.3, -0.2, 1.0]), False)   # (drops the queued forward: a different state)
        a = agent.step(obs_n, True)
        out.append(a.copy())
        obs = obs_n
    return np.stack(out), fetched


def test_queued_next_action_equals_plain_acting(hip_lib):
    """rlc_ddpg_act_queue / rlc_ddpg_act_fetch (one synchronisation per environment step): the action step() returns is
    bit for bit the one a separate rlc_ddpg_act call computes after the update, exploration noise drawn in the same
    order (agents/DDPG.py:36-48); a forward queued for another state is dropped, not returned."""
    from rlcontrol_amd.utils.main_utils import create_agent
    outs = []
    for queued in (True, False):
        cfg, env = _config(3, 32)
        agent = create_agent("DDPG", cfg)
        acts, fetched = _drive(agent, env, 3, 120, queued)
        outs.append(acts)
        if queued:
            assert fetched == 120 - 32                   # every step after the first update queued its forward (Q12)
    assert np.array_equal(outs[0], outs[1])
    assert np.all(np.isfinite(outs[0]))


_OTHER_AGENTS = {
    "SoftActorCritic": {"exploration_policy": "none", "actor_l1_dim": 128, "actor_l2_dim": 128, "critic_l1_dim": 128,
                        "critic_l2_dim": 128, "pi_lr": 1e-3, "qf_vf_lr": 1e-3, "entropy_scale": 0.1,
                        "sample_for_eval": "True", "use_true_q": "False"},
    "NAF": {"exploration_policy": "none", "l1_dim": 200, "l2_dim": 200, "noise_scale": 0.3, "learning_rate": 1e-3},
    "ReverseKL": {"exploration_policy": "none", "actor_l1_dim": 64, "actor_l2_dim": 64, "critic_l1_dim": 64,
                  "critic_l2_dim": 64, "pi_lr": 1e-3, "qf_vf_lr": 1e-3, "entropy_scale": 0.1, "sample_for_eval": "True",
                  "use_true_q": "False", "l_param": 6, "N_param": 32, "optim_type": "intg", "q_update_type": "non_sac"},
}


@pytest.mark.parametrize("name", sorted(_OTHER_AGENTS))
def test_queued_next_action_of_the_other_agents(hip_lib, name):
    """rlc_sac_/rlc_kl_/rlc_naf_act_queue + _fetch: same actions, step for step, as the un-queued agent -- including the
    host random streams (the training sample's eps is drawn at queue time and put back when the forward is dropped:
    _drive interleaves evaluation calls that sample too, sample_for_eval = "True")."""
    from rlcontrol_amd.utils.main_utils import create_agent
    outs = []
    for queued in (True, False):
        cfg, env = _config(5, 32)
        cfg.merge_config(_OTHER_AGENTS[name])
        agent = create_agent(name, cfg)
        assert agent.network_manager.queues_next_action
        acts, fetched = _drive(agent, env, 5, 90, queued)
        outs.append(acts)
        if queued:
            assert fetched == 90 - 32
    assert np.array_equal(outs[0], outs[1])
    assert np.all(np.isfinite(outs[0]))


def test_act_fetch_without_queue_is_refused(hip_lib):
    from rlcontrol_amd._lib import RlcError
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    pop = DDPGPopulation(2, 3, 1, 32, 32, 32, 16, 100, 0.01, [-1, -1, -8], [1, 1, 8], [-2.0], [2.0], 1e-3, 1e-2,
                         seeds=[1, 2])
    with pytest.raises(RlcError, match="no acting forward is queued"):
        pop.act_fetch(1)
    s = np.array([[0.5, 0.1, -1.0], [0.2, -0.3, 2.0]])
    assert pop.act_queue(s) == 2
    with pytest.raises(RlcError, match="queued forward is for agents"):
        pop.act_fetch(1)
    assert np.array_equal(pop.act_fetch(2), pop.act(s))
    pop.close()
