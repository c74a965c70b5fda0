"""GPU tests of the on-device experiment loop through the C ABI:
device sampler / environment / OU / learn gate / evaluation schedule against the CPU restatement
(oracle/rollout.py, oracle/philox.py) on the same Philox streams, plus size-independent properties at the
BASELINE shapes.

Tolerances: integer bookkeeping (episode lengths, cumulative steps, replay size, sample indices) is exact.
Trajectories are compared with a tolerance that grows along the run: the loop is a closed feedback system
(weights -> action -> state -> minibatch -> weights), so fp32 rounding differences between the fused kernels
and the oracle (1e-6 relative per update, tests/test_gpu_ddpg.py) are amplified by the pendulum dynamics.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMIN, SMAX, AMIN, AMAX = [-1, -1, -8], [1, 1, 8], [-2.0], [2.0]


def _pop(dims, B, n_agents, lr_a, lr_c, seeds, kernel, cap=4096):
    from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params
    S, A, H1, HA, HC = dims
    pop = DDPGPopulation(n_agents, S, A, H1, HA, HC, B, cap, 0.01, SMIN, SMAX, AMIN, AMAX, lr_a, lr_c, seeds=seeds)
    if kernel == "mfma":
        try:
            pop.set_kernel("mfma")
        except Exception:
            pytest.skip("MFMA kernel does not support these dimensions")
    else:
        pop.set_kernel(kernel)
    thetas = [init_params(S, A, H1, HA, HC, 100 + i) for i in range(n_agents)]
    for i, th in enumerate(thetas):
        pop.set_params(i, th, init_target=True)
    return pop, thetas


def test_device_sampler_equals_philox_restatement(hip_lib):
    """rlc_replay_sample_indices == oracle.philox.sample_distinct bit for bit, dense and sparse regimes."""
    from oracle import philox
    pop, _ = _pop((3, 1, 16, 16, 16), 16, 1, 1e-3, 1e-2, [987654321], "generic")
    rng = np.random.RandomState(0)
    n_total, call = 0, 0
    for grow, k in ((20, 16), (30, 16), (400, 16), (3000, 100)):
        n = grow
        pop.replay_add_batch(0, rng.randn(n, 3), rng.randn(n, 1), rng.randn(n), rng.randn(n, 3), np.ones(n))
        n_total = min(n_total + n, 4096)
        for _ in range(3):
            got = pop.replay_sample_indices(0, k)
            want = philox.sample_distinct(n_total, k, 987654321, call)
            call += 1
            assert np.array_equal(got, want), (n_total, k)


@pytest.mark.parametrize("kernel", ["generic", "mfma"])
def test_rollout_matches_cpu_restatement(hip_lib, kernel):
    from oracle.ddpg import Dims
    from oracle.rollout import RolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    dims, B = (3, 1, 32, 32, 32), 16
    seeds, lr_a, lr_c = [11, 7777777777], [1e-3, 5e-4], [1e-2, 2e-3]
    pop, thetas = _pop(dims, B, 2, lr_a, lr_c, seeds, kernel)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00011, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0)
    assert exp.advance(60) == 60                 # two calls: the schedule must survive the split
    exp.advance(1000)
    assert exp.total_steps == 110
    res = exp.results()
    for a in range(2):
        orc = RolloutOracle(Dims(*dims), thetas[a], lr_a[a], lr_c[a], 0.01, SMIN, SMAX, AMIN, AMAX, seeds[a], B, 4096,
                            0.99, 0, 25, 110, 40, 2).run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        # --- exact bookkeeping
        assert tl == orc.train_len == [25] * 4 and tc == orc.train_cum == [25, 50, 75, 100]
        assert ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len and n_started == 5
        assert pop.replay_size(a) == len(orc.replay) == 110 - 4
        obs, ep_step = exp.observation(a)
        assert ep_step == orc.last_step == 10
        # --- trajectory: replay contents in insertion order
        s, act, r, s2, g = pop.replay_gather(a, np.arange(106))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        orr = np.array([t[2] for t in orc.replay]); os2 = np.array([t[3] for t in orc.replay])
        assert np.array_equal(g, np.array([t[4] for t in orc.replay]))
        pre = B + 1                               # no update has touched the weights yet: float rounding only
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=2e-6)
        assert np.allclose(r[:pre], orr[:pre], atol=1e-5) and np.allclose(s2[:pre], os2[:pre], atol=2e-6)
        assert np.allclose(s, os_, atol=2e-3) and np.allclose(act, oa, atol=2e-3)
        assert np.allclose(r, orr, atol=2e-3) and np.allclose(s2, os2, atol=2e-3)
        assert np.allclose(obs, orc.last_obs, atol=2e-3)
        # --- returns and weights
        assert np.allclose(tr, orc.train_ret, rtol=1e-3, atol=1e-2)
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)      # evaluation 0: initial weights
        assert np.allclose(er, orc.eval_ret, rtol=2e-3, atol=2e-2)
        th = pop.get_blob(a, "theta")
        assert np.max(np.abs(th - orc.net.theta)) < 2e-3 * np.max(np.abs(orc.net.theta))
        pw = pop.get_beta_powers(a)
        assert np.allclose(pw, orc.net.pw, rtol=1e-6)                          # same number of Adam steps


@pytest.mark.parametrize("norm,sep", [(True, False), (False, True), (True, True)])
def test_ddpg_variant_rollout_matches_cpu_restatement(hip_lib, norm, sep):
    """norm_type 'layer' / separate networks in the on-device loop (train step, fused any-shape update, per-episode
    evaluation kernel through ddpg_greedy_forward) against the loop around oracle/ddpg_variants.py"""
    from oracle.ddpg_variants import VDims, init_params
    from oracle.rollout import VariantRolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_ddpg import DDPGPopulation
    dims, B = (3, 1, 32, 32, 32), 16
    seeds, lr_a, lr_c = [11, 7777777777], [1e-3, 5e-4], [1e-2, 2e-3]
    d = VDims(*dims, norm=norm, separate=sep)
    pop = DDPGPopulation(2, *dims, B, 4096, 0.01, SMIN, SMAX, AMIN, AMAX, lr_a, lr_c, seeds=seeds,
                         norm_type="layer" if norm else "input_norm", separate_networks=sep)
    thetas = [init_params(d, 100 + i) for i in range(2)]
    for i, th in enumerate(thetas):
        pop.set_params(i, th, init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00011, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0)
    assert exp.advance(60) == 60
    exp.advance(1000)
    assert exp.total_steps == 110
    res = exp.results()
    for a in range(2):
        orc = VariantRolloutOracle(d, thetas[a], lr_a[a], lr_c[a], 0.01, SMIN, SMAX, AMIN, AMAX, seeds[a], B, 4096,
                                   0.99, 0, 25, 110, 40, 2).run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == orc.train_len == [25] * 4 and tc == orc.train_cum == [25, 50, 75, 100]
        assert ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len and n_started == 5
        assert pop.replay_size(a) == len(orc.replay) == 110 - 4
        s, act, r, s2, g = pop.replay_gather(a, np.arange(106))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        pre = B + 1
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=2e-6)
        assert np.allclose(s, os_, atol=3e-3) and np.allclose(act, oa, atol=3e-3)
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)      # evaluation 0: initial weights
        assert np.allclose(er, orc.eval_ret, rtol=3e-3, atol=3e-2)
        th = pop.get_blob(a, "theta")
        assert np.max(np.abs(th - orc.net.theta)) < 3e-3 * np.max(np.abs(orc.net.theta))
        assert np.allclose(pop.get_beta_powers(a), orc.net.pw, rtol=1e-6)
    pop.close()


def test_rollout_quirk_q8_noise_reset_after_eval(hip_lib):
    """An evaluation in the middle of a training episode resets the OU state AFTER the pending action was
    drawn (experiment.py:121-133): with eval_episodes > 0 the noise restarts from mu, with 0 it does not.
    lr = 0 keeps the policy fixed so that the actions differ only through the noise."""
    from rlcontrol_amd.device_experiment import DeviceExperiment
    acts = {}
    for ev in (0, 1):
        pop, _ = _pop((3, 1, 16, 16, 16), 4, 1, 0.0, 0.0, [5], "generic")
        env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00002, "EpisodeSteps": 50,
               "EvalIntervalMilSteps": 0.00001, "EvalEpisodes": ev}
        DeviceExperiment(pop, env).run()
        acts[ev] = pop.replay_gather(0, np.arange(20))[1].ravel()
    # steps 1..11 identical (the action of step 11 was drawn before the evaluation at total=10) ...
    assert np.array_equal(acts[0][:11], acts[1][:11])
    # ... from step 12 on the OU state differs
    assert not np.allclose(acts[0][11:], acts[1][11:])


def test_rollout_full_size_properties(hip_lib):
    """BASELINE shapes (H=200, B=100, MFMA kernel, several agents): every stored transition obeys the
    Pendulum dynamics (recomputed on the host in float64), logs are consistent, everything is finite."""
    from rlcontrol_amd.device_experiment import DeviceExperiment
    n_agents = 6
    pop, thetas = _pop((3, 1, 200, 200, 200), 100, n_agents, 1e-3, 1e-2, list(range(50, 50 + n_agents)), "mfma",
                       cap=2048)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00065, "EpisodeSteps": -1,
           "EvalIntervalMilSteps": 0.0003, "EvalEpisodes": 3}
    res = DeviceExperiment(pop, env).run(chunk=200)
    for a in range(n_agents):
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == [200, 200, 200] and tc == [200, 400, 600] and ts == [0, 300, 600] and n_started == 4
        assert np.array(el).tolist() == [[200] * 3] * 3
        assert np.isfinite(tr).all() and np.isfinite(er).all() and (np.array(tr) <= 0).all()
        n = pop.replay_size(a)
        assert n == 650 - 3
        s, act, r, s2, g = pop.replay_gather(a, np.arange(n))
        assert (g == 0.99).all() and (np.abs(act) <= 2.0).all()
        th, thd, u = np.arctan2(s[:, 1], s[:, 0]), s[:, 2], act[:, 0]
        assert np.allclose(r, -(th ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2), atol=2e-5)
        thd2 = thd + (-15.0 * np.sin(th + np.pi) + 3.0 * u) * 0.05
        th2 = th + thd2 * 0.05
        thd2 = np.clip(thd2, -8, 8)
        assert np.allclose(s2, np.stack([np.cos(th2), np.sin(th2), thd2], 1), atol=2e-5)
        # consecutive transitions chain (s'_t == s_{t+1}) except across episode boundaries
        same = np.all(s2[:-1] == s[1:], axis=1)
        assert same.sum() == n - 1 - 3
        # learning happened: the weights moved, the Adam step counters advanced 650-100 times
        assert not np.array_equal(pop.get_blob(a, "theta"), thetas[a])
        assert np.isclose(pop.get_beta_powers(a)[1], 0.999 ** (1 + 650 - 100 - 0), rtol=1e-4)
    # agents are independent: different seeds -> different trajectories
    assert not np.array_equal(pop.replay_gather(0, np.arange(10))[0], pop.replay_gather(1, np.arange(10))[0])


def test_main_device_rollout_pickle(hip_lib, tmp_path):
    """main.py --device_rollout: all indices of the range in one population, reference pickle schema
    (main.py:80-95,188-209), runs grouped under their setting in index order."""
    import json
    import pickle
    import main as drv
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00025, "EpisodeSteps": 100,
           "EvalIntervalMilSteps": 0.0001, "EvalEpisodes": 2}
    agent = {"agent": "DDPG", "sweeps": {"shared_l1_dim": [32], "actor_l2_dim": [32], "critic_l2_dim": [32],
                                         "actor_lr": [1e-3, 1e-4], "critic_lr": [1e-2], "norm_type": ["input_norm"],
                                         "exploration_policy": ["ou_noise"], "batch_size": [16],
                                         "buffer_size": [1000]}}
    ej, aj = tmp_path / "Pendulum-v0.json", tmp_path / "ddpg.json"
    ej.write_text(json.dumps(env)); aj.write_text(json.dumps(agent))
    drv.main(["--env_json", str(ej), "--agent_json", str(aj), "--indices", "0", "1", "5", "--save_dir", str(tmp_path),
              "--device_rollout", "--quiet"])
    with open(tmp_path / "Pendulum-v0_ddpgresults" / "data_0_1_5.pkl", "rb") as f:
        data = pickle.load(f)
    assert sorted(data["experiment_data"]) == [0, 1]
    assert [r["random_seed"] for r in data["experiment_data"][0]["runs"]] == [0, 1, 2]     # indices 0, 2, 4
    assert [r["random_seed"] for r in data["experiment_data"][1]["runs"]] == [0, 1]        # indices 1, 3
    assert data["experiment_data"][1]["agent_params"]["actor_lr"] == 1e-4
    run = data["experiment_data"][0]["runs"][0]
    assert run["eval_episode_rewards"].shape == (3, 2) and run["timesteps_at_eval"].tolist() == [0, 100, 200]
    assert run["train_episode_steps"].tolist() == [100, 100] and run["total_train_episodes"] == 3
    assert set(run) == {"random_seed", "total_timesteps", "eval_interval_timesteps", "episodes_per_eval",
                        "eval_episode_rewards", "eval_episode_steps", "timesteps_at_eval", "train_episode_steps",
                        "train_episode_rewards", "total_train_episodes", "eval_time", "train_time"}


def test_rollout_ring_eviction_matches_fifo(hip_lib):
    """Replay capacity smaller than the run: the device ring evicts the oldest transition exactly like the
    reference's RandomAccessQueue (utils/custom_collections.py:83-101); logical order = insertion order."""
    from oracle.ddpg import Dims
    from oracle.rollout import RolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    dims, B, cap = (3, 1, 32, 32, 32), 16, 64
    pop, thetas = _pop(dims, B, 1, [1e-3], [1e-2], [99], "generic", cap=cap)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00015, "EpisodeSteps": 40,
           "EvalIntervalMilSteps": 0.0001, "EvalEpisodes": 1}
    DeviceExperiment(pop, env).run()
    orc = RolloutOracle(Dims(*dims), thetas[0], 1e-3, 1e-2, 0.01, SMIN, SMAX, AMIN, AMAX, 99, B, cap, 0.99, 0, 40, 150,
                        100, 1).run()
    assert pop.replay_size(0) == len(orc.replay) == cap
    s, act, r, s2, g = pop.replay_gather(0, np.arange(cap))
    assert np.allclose(s, np.array([t[0] for t in orc.replay]), atol=5e-3)
    assert np.allclose(act, np.array([t[1] for t in orc.replay]), atol=5e-3)
    # structure independent of float drift: consecutive transitions chain except over the episode boundaries
    same = np.all(s2[:-1] == s[1:], axis=1)
    osame = np.array([np.array_equal(orc.replay[i][3], orc.replay[i + 1][0]) for i in range(cap - 1)])
    assert np.array_equal(same, osame)
    with pytest.raises(Exception):
        pop.replay_gather(0, [cap])          # index out of range (custom_collections.py:48,58)


def test_sac_rollout_matches_cpu_restatement(hip_lib):
    """The on-device loop for a SoftActorCritic population (sac_rollout_device.h) against oracle/rollout.py's
    SacRolloutOracle on the same Philox streams: exact bookkeeping, trajectory within the drift the closed loop
    amplifies (see the module docstring)."""
    from oracle.rollout import SacRolloutOracle
    from oracle.sac import SacDims, init_params
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_sac import SACPopulation
    dims, B = (3, 1, 32, 32, 32, 32), 16
    seeds, pi_lr, qv_lr, alpha = [21, 99999999999], [1e-3, 5e-4], [1e-3, 2e-3], [0.2, 0.05]
    pop = SACPopulation(2, *dims, B, 4096, 0.01, -8.0, 8.0, 2.0, pi_lr, qv_lr, alpha, seeds=seeds)
    d = SacDims(*dims)
    thetas = []
    for i in range(2):
        th = init_params(d, 300 + i)
        lay, _ = d.layout()
        off, shp = lay["pWs"]
        th[off:off + int(np.prod(shp))] *= 0.02          # well-conditioned log-std head (tests/test_sac.py)
        thetas.append(th)
        pop.set_params(i, th, init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00009, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0)
    assert exp.advance(37) == 37
    exp.advance(1000)
    assert exp.total_steps == 90
    res = exp.results()
    for a in range(2):
        orc = SacRolloutOracle(d, thetas[a], pi_lr[a], qv_lr[a], alpha[a], 0.01, -8.0, 8.0, 2.0, seeds[a], B, 4096, 0.99,
                               0, 25, 90, 40, 2).run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == orc.train_len == [25] * 3 and tc == orc.train_cum == [25, 50, 75]
        assert ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len and n_started == 4
        assert pop.replay_size(a) == len(orc.replay) == 90 - 3
        s, act, r, s2, g = pop.replay_gather(a, np.arange(87))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        pre = B + 1
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=5e-6)
        assert np.allclose(s, os_, atol=5e-3) and np.allclose(act, oa, atol=5e-3)
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)      # evaluation 0: initial weights, mean action
        assert np.allclose(er, orc.eval_ret, rtol=5e-3, atol=5e-2)
        th = pop.get_blob(a, "theta")
        assert np.max(np.abs(th - orc.net.theta)) < 5e-3 * np.max(np.abs(orc.net.theta))
        assert np.allclose(pop.get_beta_powers(a), orc.net.pw, rtol=1e-6)
    pop.close()


def test_main_device_rollout_sac_pickle(hip_lib, tmp_path):
    """main.py --device_rollout with the SoftActorCritic agent: per-agent pi_lr / entropy_scale from the sweep."""
    import json
    import pickle
    import main as drv
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00012, "EpisodeSteps": 50,
           "EvalIntervalMilSteps": 0.00005, "EvalEpisodes": 2}
    agent = {"agent": "SoftActorCritic",
             "sweeps": {"norm_type": ["input_norm"], "exploration_policy": ["none"], "actor_l1_dim": [32],
                        "actor_l2_dim": [32], "critic_l1_dim": [32], "critic_l2_dim": [32], "pi_lr": [1e-3],
                        "qf_vf_lr": [1e-3], "sample_for_eval": ["False"], "use_true_q": ["False"],
                        "entropy_scale": [0.1, 0.01], "batch_size": [16], "buffer_size": [1000]}}
    ej, aj = tmp_path / "Pendulum-v0.json", tmp_path / "sac.json"
    ej.write_text(json.dumps(env)); aj.write_text(json.dumps(agent))
    drv.main(["--env_json", str(ej), "--agent_json", str(aj), "--indices", "0", "1", "4", "--save_dir", str(tmp_path),
              "--device_rollout", "--quiet"])
    with open(tmp_path / "Pendulum-v0_sacresults" / "data_0_1_4.pkl", "rb") as f:
        data = pickle.load(f)
    assert sorted(data["experiment_data"]) == [0, 1]
    assert [r["random_seed"] for r in data["experiment_data"][0]["runs"]] == [0, 1]
    assert data["experiment_data"][1]["agent_params"]["entropy_scale"] == 0.01
    run = data["experiment_data"][1]["runs"][1]
    assert run["eval_episode_rewards"].shape == (3, 2) and run["timesteps_at_eval"].tolist() == [0, 50, 100]
    assert run["train_episode_steps"].tolist() == [50, 50] and np.isfinite(run["eval_episode_rewards"]).all()


@pytest.mark.parametrize("noise", [0.3, 1.0])
def test_naf_rollout_matches_cpu_restatement(hip_lib, noise):
    """The on-device loop for a NAF population (naf_rollout_device.h) against oracle/rollout.py's NafRolloutOracle
    on the same Philox streams."""
    from oracle.naf import NafDims, init_params
    from oracle.rollout import NafRolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_naf import NAFPopulation
    dims, B, seeds, lr = (3, 1, 32, 32), 16, [8, 123456789012], [1e-3, 3e-4]
    pop = NAFPopulation(2, *dims, B, 4096, 0.01, SMIN, SMAX, AMAX, lr, seeds=seeds)
    d = NafDims(*dims)
    thetas = [init_params(d, 40 + i) for i in range(2)]
    for i in range(2):
        pop.set_params(i, thetas[i], init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00009, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0, noise_scale=noise)
    exp.advance(50)
    exp.advance(1000)
    assert exp.total_steps == 90
    res = exp.results()
    for a in range(2):
        orc = NafRolloutOracle(d, thetas[a], lr[a], 0.01, SMIN, SMAX, AMAX, noise, seeds[a], B, 4096, 0.99, 0, 25, 90, 40,
                               2).run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == orc.train_len == [25] * 3 and tc == orc.train_cum == [25, 50, 75]
        assert ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len and n_started == 4
        assert pop.replay_size(a) == len(orc.replay) == 87
        s, act, r, s2, g = pop.replay_gather(a, np.arange(87))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        pre = B + 1
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=1e-5)
        assert np.allclose(s, os_, atol=5e-3) and np.allclose(act, oa, atol=5e-3)
        assert (np.abs(act) <= 2.0).all() and np.std(act - np.clip(oa, -2, 2)) < 1e-2
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)
        assert np.allclose(er, orc.eval_ret, rtol=5e-3, atol=5e-2)
        th = pop.get_blob(a, "theta")
        assert np.max(np.abs(th - orc.net.theta)) < 5e-3 * np.max(np.abs(orc.net.theta))
    pop.close()


def test_naf_rollout_clips_to_an_asymmetric_action_box(hip_lib):
    """naf_network.py:176 clips the exploration draw to [action_min, action_max]; rlc_naf_config::action_min carries the
    lower bound to the on-device loop (default -action_max).  A large noise scale so that both bounds bite."""
    from oracle.naf import NafDims, init_params
    from oracle.rollout import NafRolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_naf import NAFPopulation
    dims, B, seed, lr, amin = (3, 1, 32, 32), 16, 77, 1e-3, [-0.6]
    pop = NAFPopulation(1, *dims, B, 4096, 0.01, SMIN, SMAX, AMAX, [lr], seeds=[seed], action_min=amin)
    d = NafDims(*dims)
    theta = init_params(d, 41)
    pop.set_params(0, theta, init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00006, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 1}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0, noise_scale=3.0)
    exp.advance(1000)
    orc = NafRolloutOracle(d, theta, lr, 0.01, SMIN, SMAX, AMAX, 3.0, seed, B, 4096, 0.99, 0, 25, 60, 40, 1,
                           action_min=amin).run()
    n = pop.replay_size(0)
    assert n == len(orc.replay)
    _, act, _, _, _ = pop.replay_gather(0, np.arange(n))
    oa = np.array([t[1] for t in orc.replay])
    assert act.min() == np.float32(-0.6) and act.max() == np.float32(2.0)        # both bounds reached, neither exceeded
    assert np.allclose(act[:B + 1], oa[:B + 1], atol=1e-5)                         # before the first update: same draws
    assert np.mean(np.isclose(act, oa, atol=5e-3)) > 0.9
    pop.close()


def test_naf_layer_norm_rollout_matches_cpu_restatement(hip_lib):
    """norm_type 'layer' in the on-device loop: the training step, the fused update and the evaluation kernel all go
    through the three layer norms (naf_policy.h, naf_generic.hip) -- against the torch restatement on the same streams"""
    from oracle.naf import NafDims
    from oracle.naf_variants import init_params
    from oracle.rollout import NafRolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_naf import NAFPopulation
    dims, B, seeds, lr, noise = (3, 1, 32, 32), 16, [8, 123456789012], [1e-3, 3e-4], 0.3
    pop = NAFPopulation(2, *dims, B, 4096, 0.01, SMIN, SMAX, AMAX, lr, seeds=seeds, norm_type="layer")
    thetas = [init_params(dims, 40 + i, True) for i in range(2)]
    for i in range(2):
        pop.set_params(i, thetas[i], init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00009, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0, noise_scale=noise)
    exp.advance(1000)
    assert exp.total_steps == 90
    res = exp.results()
    for a in range(2):
        orc = NafRolloutOracle(NafDims(*dims), thetas[a], lr[a], 0.01, SMIN, SMAX, AMAX, noise, seeds[a], B, 4096, 0.99, 0,
                               25, 90, 40, 2, norm_type="layer").run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == orc.train_len == [25] * 3 and ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len
        assert pop.replay_size(a) == len(orc.replay) == 87
        s, act, r, s2, g = pop.replay_gather(a, np.arange(87))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        pre = B + 1
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=1e-5)
        assert np.allclose(s, os_, atol=5e-3) and np.allclose(act, oa, atol=5e-3)
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)
        assert np.allclose(er, orc.eval_ret, rtol=5e-3, atol=5e-2)
        th, want = pop.get_blob(a, "theta"), orc.net.theta.numpy()
        assert np.max(np.abs(th - want)) < 5e-3 * np.max(np.abs(want))
    pop.close()


@pytest.mark.parametrize("kind,kernel", [("reverse", "mfma"), ("forward", "mfma"), ("reverse", "generic")])
def test_kl_rollout_matches_cpu_restatement(hip_lib, kind, kernel):
    """The on-device loop for a ReverseKL / ForwardKL population (the SoftActorCritic train step on the KL kernels)
    against oracle/rollout.py's KlRolloutOracle (torch restatement) on the same Philox streams."""
    from oracle.kl_torch import KlDims, init_params
    from oracle.rollout import KlRolloutOracle
    from rlcontrol_amd.device_experiment import DeviceExperiment
    from rlcontrol_amd.hip_kl import KLPopulation
    dims, B, n_param = (3, 1, 32, 32, 32, 32), 16, 18
    seeds, pi_lr, qv_lr, alpha = [21, 99999999999], [1e-3, 5e-4], [1e-3, 2e-3], [0.2, 0.05]
    pop = KLPopulation(kind, 2, *dims, B, 4096, 0.01, 2.0, pi_lr, qv_lr, alpha, seeds=seeds, n_param=n_param)
    pop.set_kernel(kernel)
    d = KlDims(*dims)
    thetas = [init_params(d, 300 + i) for i in range(2)]
    for i in range(2):
        pop.set_params(i, thetas[i], init_target=True)
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00009, "EpisodeSteps": 25,
           "EvalIntervalMilSteps": 0.00004, "EvalEpisodes": 2}
    exp = DeviceExperiment(pop, env, gamma=0.99, warmup_steps=0)
    assert exp.advance(37) == 37
    exp.advance(1000)
    assert exp.total_steps == 90
    res = exp.results()
    for a in range(2):
        orc = KlRolloutOracle(kind, d, thetas[a], pi_lr[a], qv_lr[a], alpha[a], 0.01, 2.0, n_param, seeds[a], B, 4096, 0.99,
                              0, 25, 90, 40, 2).run()
        tr, er, tl, el, ts, _, _, n_started, tc = res[a]
        assert tl == orc.train_len == [25] * 3 and tc == orc.train_cum == [25, 50, 75]
        assert ts == orc.timesteps_at_eval == [0, 40, 80] and el == orc.eval_len and n_started == 4
        assert pop.replay_size(a) == len(orc.replay) == 90 - 3
        assert pop.get_step(a) == orc.net.step == orc.n_updates
        s, act, r, s2, g = pop.replay_gather(a, np.arange(87))
        os_ = np.array([t[0] for t in orc.replay]); oa = np.array([t[1] for t in orc.replay])
        pre = B + 1
        assert np.allclose(s[:pre], os_[:pre], atol=2e-6) and np.allclose(act[:pre], oa[:pre], atol=5e-6)
        assert np.allclose(s, os_, atol=5e-3) and np.allclose(act, oa, atol=5e-3)
        assert np.allclose(er[0], orc.eval_ret[0], rtol=1e-5, atol=1e-4)      # evaluation 0: initial weights, mean action
        assert np.allclose(er, orc.eval_ret, rtol=5e-3, atol=5e-2)
        th = pop.get_blob(a, "theta")
        assert np.max(np.abs(th - orc.net.theta.numpy())) < 5e-3 * np.max(np.abs(orc.net.theta.numpy()))
    pop.close()


def test_main_device_rollout_kl_pickle(hip_lib, tmp_path):
    """main.py --device_rollout with the ForwardKL agent: per-agent learning rates / entropy scale from the sweep."""
    import json
    import pickle
    import main as drv
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00012, "EpisodeSteps": 50,
           "EvalIntervalMilSteps": 0.00005, "EvalEpisodes": 2}
    agent = {"agent": "ForwardKL",
             "sweeps": {"norm_type": ["input_norm"], "exploration_policy": ["none"], "actor_l1_dim": [32],
                        "actor_l2_dim": [32], "critic_l1_dim": [32], "critic_l2_dim": [32], "pi_lr": [1e-3],
                        "qf_vf_lr": [1e-3], "sample_for_eval": ["False"], "use_true_q": ["False"],
                        "entropy_scale": [0.1, 0.01], "l_param": [6], "N_param": [16], "optim_type": ["intg"],
                        "q_update_type": ["non_sac"], "batch_size": [16], "buffer_size": [1000]}}
    ej, aj = tmp_path / "Pendulum-v0.json", tmp_path / "forward_kl.json"
    ej.write_text(json.dumps(env)); aj.write_text(json.dumps(agent))
    drv.main(["--env_json", str(ej), "--agent_json", str(aj), "--indices", "0", "1", "4", "--save_dir", str(tmp_path),
              "--device_rollout", "--quiet"])
    with open(tmp_path / "Pendulum-v0_forward_klresults" / "data_0_1_4.pkl", "rb") as f:
        data = pickle.load(f)
    assert sorted(data["experiment_data"]) == [0, 1]
    assert [r["random_seed"] for r in data["experiment_data"][0]["runs"]] == [0, 1]
    assert data["experiment_data"][1]["agent_params"]["entropy_scale"] == 0.01
    run = data["experiment_data"][1]["runs"][1]
    assert run["eval_episode_rewards"].shape == (3, 2) and run["timesteps_at_eval"].tolist() == [0, 50, 100]
    assert run["train_episode_steps"].tolist() == [50, 50] and np.isfinite(run["eval_episode_rewards"]).all()
