"""N > 1 path on CPU: world_size 2, gloo.  Ranks own disjoint INDEX shards, exchange nothing while
"training" and meet in exactly one all_gather of packed episodic returns (rlcontrol_amd/sweep.py)."""
import os
import sys

import numpy as np
import torch  # noqa: F401  (initialises torch.distributed before the spawned workers import it)
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_run(index):
    rng = np.random.RandomState(index)
    return {"random_seed": index // 49, "eval_episode_rewards": rng.randn(5, 3) - index,
            "train_episode_rewards": rng.randn(4 + index % 3)}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from rlcontrol_amd import sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sweep.rank_indices(0, 49, 49 * 5, rank, world)           # 5 seeds of setting 0
    vecs = [sweep.pack_run(i, _fake_run(i), (5, 3), 8) for i in mine]
    allv = sweep.all_gather_runs(vecs, runs_per_rank=3, vec_len=4 + 15 + 8)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, allv))


def test_two_rank_sweep_gathers_every_run_once():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from rlcontrol_amd import sweep
    by_rank = {r: (mine, allv) for r, mine, allv in got}
    assert by_rank[0][0] == [0, 98, 196] and by_rank[1][0] == [49, 147]       # disjoint round-robin shards
    assert np.array_equal(np.nan_to_num(by_rank[0][1]), np.nan_to_num(by_rank[1][1]))   # same on every rank
    rows = [r for r in by_rank[0][1] if not np.isnan(r[0])]
    assert sorted(int(r[0]) for r in rows) == [0, 49, 98, 147, 196]          # every run exactly once
    for r in rows:
        run = sweep.unpack_run(r, (5, 3), 8)
        want = _fake_run(run["index"])
        assert run["random_seed"] == want["random_seed"]
        assert np.array_equal(run["eval_episode_rewards"], want["eval_episode_rewards"])
        assert np.array_equal(run["train_episode_rewards"], want["train_episode_rewards"])


def test_rank_indices_cover_the_range_without_overlap():
    sys.path.insert(0, ROOT)
    from rlcontrol_amd import sweep
    for world in (1, 2, 8):
        parts = [sweep.rank_indices(3, 7, 200, r, world) for r in range(world)]
        flat = sorted(i for p in parts for i in p)
        assert flat == list(range(3, 200, 7))


# ---- the driver itself under two ranks (gloo): shards, one all-gather, rank 0 writes the complete pickle --------
ENV2 = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00025, "EpisodeSteps": 100,
        "EvalIntervalMilSteps": 0.0001, "EvalEpisodes": 2}


class _FakeAgent(object):
    def __init__(self, config):
        self.bias = config.actor_lr * 100 + config.random_seed

    def start(self, s, is_train):
        return np.array([min(self.bias, 2.0)])

    step = start

    def update(self, *a):
        pass

    def reset(self):
        pass


def _main_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    import main as drv
    drv.create_agent = lambda name, cfg: _FakeAgent(cfg)
    envf = os.path.join(tmp, "Pendulum-v0.json")
    if rank == 0:
        with open(envf, "w") as f:
            json.dump(ENV2, f)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    drv.main(["--env_json", envf, "--agent_json", os.path.join(ROOT, "jsonfiles/agent/ddpg.json"),
              "--indices", "0", "7", "70", "--save_dir", os.path.join(tmp, "res"), "--quiet"])
    dist.destroy_process_group()


def test_main_under_two_ranks_writes_one_complete_pickle(tmp_path):
    import pickle
    ctx = mp.get_context("spawn")
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    with open(tmp_path / "res" / "Pendulum-v0_ddpgresults" / "data_0_7_70.pkl", "rb") as f:
        multi = pickle.load(f)
    # the per-rank shard pickles (re-written after every index) are gone once the merged pickle exists
    assert sorted(os.listdir(tmp_path / "res" / "Pendulum-v0_ddpgresults")) == ["data_0_7_70.pkl"]
    # the same sweep in one process
    sys.path.insert(0, ROOT)
    import json
    import main as drv
    old = drv.create_agent
    drv.create_agent = lambda name, cfg: _FakeAgent(cfg)
    try:
        envf = tmp_path / "Pendulum-v0.json"
        single = drv.main(["--env_json", str(envf), "--agent_json", os.path.join(ROOT, "jsonfiles/agent/ddpg.json"),
                           "--indices", "0", "7", "70", "--save_dir", str(tmp_path / "one"), "--quiet"])
    finally:
        drv.create_agent = old
    assert sorted(multi["experiment_data"]) == sorted(single["experiment_data"])
    for sid in single["experiment_data"]:
        a, b = multi["experiment_data"][sid], single["experiment_data"][sid]
        assert a["agent_params"]["actor_lr"] == b["agent_params"]["actor_lr"]
        assert [r["random_seed"] for r in a["runs"]] == [r["random_seed"] for r in b["runs"]]
        for ra, rb in zip(a["runs"], b["runs"]):
            for k in ("eval_episode_rewards", "eval_episode_steps", "timesteps_at_eval", "train_episode_steps",
                      "train_episode_rewards"):
                assert np.array_equal(np.asarray(ra[k]), np.asarray(rb[k])), k
            assert ra["total_train_episodes"] == rb["total_train_episodes"]


# ---------------------------------------------------------------------------------------------------------
# bench.py's N > 1 branch: barrier-bracketed timed region, MAX over ranks, the one all-gather, whole-job value
# ---------------------------------------------------------------------------------------------------------
class _StubPopulation(object):
    """what bench.measure() needs from a population: update / sync / timer; rank-dependent step time"""
    def __init__(self, seconds_per_update):
        self.dt, self.updates, self.t0 = seconds_per_update, 0, None

    def update(self, n):
        import time
        time.sleep(self.dt * n)
        self.updates += n

    def sync(self):
        pass

    def timer_begin(self):
        import time
        self.t0 = time.perf_counter()

    def timer_end(self):
        import time
        return (time.perf_counter() - self.t0) * 1e3


def _bench_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import json
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pop = _StubPopulation(0.002 * (1 + rank))              # rank 1 is twice as slow: it sets the MAX
    dt, ev_ms = bench.measure(pop, 5, steps=4, warmup=2, dist=dist, device_sync=None)
    dt_max, gathered = bench.reduce_over_ranks(dt, 10.0 + rank, dist, world, "cpu")
    with open(os.path.join(out, "r%d.json" % rank), "w") as f:
        json.dump({"dt": dt, "ev_ms": ev_ms, "dt_max": dt_max, "gathered": gathered, "updates": pop.updates,
                   "value": bench.aggregate_value(world, 3, 5, 4, dt_max)}, f)
    dist.destroy_process_group()


def test_bench_multi_rank_bookkeeping_under_gloo(tmp_path):
    import json
    ctx = mp.get_context("spawn")
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0, r1 = (json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2))
    assert r0["updates"] == r1["updates"] == (2 + 4) * 5            # warm-up steps run, exactly K steps timed
    assert r0["dt_max"] == r1["dt_max"] >= max(r0["dt"], r1["dt"]) - 1e-9       # MAX over ranks, same on every rank
    assert r1["dt"] >= 4 * 5 * 0.004 and r0["dt"] >= r1["dt"] - 0.02           # the barrier holds rank 0 for the slow rank
    assert r0["ev_ms"] < r1["ev_ms"]                                 # per-rank kernel time is NOT what value uses
    assert r0["gathered"] == r1["gathered"] == [10.0, 11.0]           # the one collective: per-rank results, rank order
    assert abs(r0["value"] - 2 * 3 * 5 * 4 / r0["dt_max"]) < 1e-6     # whole-job units / max time


def test_bench_single_rank_reduce_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_over_ranks(0.5, 7.0) == (0.5, [7.0])
    assert bench.aggregate_value(1, 256, 32, 20, 2.0) == 256 * 32 * 20 / 2.0
    roof, hbm = bench.roofline_record(73.2e6, 2634064.0, 256 * 32, 10.6e-3)
    assert abs(roof["frac"] - 73.2e6 * 8192 / 10.6e-3 / 157.3e12) < 1e-12 and roof["bound"] == "mfma"
    assert abs(hbm["achieved"] - 2634064.0 * 8192 / 10.6e-3 / 1e9) < 1e-6


# ---- a rank that fails must not leave the others waiting in the collective -------------------------------------
class _FailingAgent(_FakeAgent):
    def update(self, *a):
        raise RuntimeError("boom on this rank")


def _main_worker_one_rank_fails(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    import main as drv
    drv.create_agent = (lambda name, cfg: _FailingAgent(cfg)) if rank == 1 else (lambda name, cfg: _FakeAgent(cfg))
    envf = os.path.join(tmp, "Pendulum-v0.json")
    if rank == 0:
        with open(envf, "w") as f:
            json.dump(ENV2, f)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    try:
        drv.main(["--env_json", envf, "--agent_json", os.path.join(ROOT, "jsonfiles/agent/ddpg.json"),
                  "--indices", "0", "7", "28", "--save_dir", os.path.join(tmp, "res"), "--quiet"])
    except RuntimeError as e:
        with open(os.path.join(tmp, "err%d.txt" % rank), "w") as f:
            f.write(str(e))
        sys.exit(3)
    sys.exit(0)


def test_failing_rank_ends_the_sweep_on_every_rank_promptly(tmp_path):
    ctx = mp.get_context("spawn")
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_main_worker_one_rank_fails, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 3                      # both ranks raised -- nobody hangs in the all-gather
    assert "this rank failed" in (tmp_path / "err1.txt").read_text()
    assert "another rank failed" in (tmp_path / "err0.txt").read_text()
    resdir = tmp_path / "res" / "Pendulum-v0_ddpgresults"
    assert not (resdir / "data_0_7_28.pkl").exists()
    # ... but what the healthy rank finished is on disk in the reference's schema (main.py:205-209 re-pickles after every
    # index), ready for the reference's offline merge (main_concurrent.py:107-154)
    import pickle
    with open(resdir / "data_0_7_28.rank0of2.pkl", "rb") as f:
        shard = pickle.load(f)
    seeds = sorted((sid, r["random_seed"]) for sid, sd in shard["experiment_data"].items() for r in sd["runs"])
    assert seeds == [(0, 0), (14, 0)]               # indices 0 and 14 of range(0, 28, 7) are rank 0's
    assert shard["experiment"]["agent"]["agent_name"] == "DDPG"
    assert not (resdir / "data_0_7_28.rank1of2.pkl").exists()      # rank 1 failed on its first index


def _main_worker_forced(port, tmp):
    sys.path.insert(0, ROOT)
    import json
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      RLC_FORCE_DIST="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    import main as drv
    drv.create_agent = lambda name, cfg: _FakeAgent(cfg)
    envf = os.path.join(tmp, "Pendulum-v0.json")
    with open(envf, "w") as f:
        json.dump(ENV2, f)
    calls = []
    orig_gather, orig_reduce = dist.all_gather, dist.all_reduce
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), orig_gather(*a, **k))[1]
    dist.all_reduce = lambda *a, **k: (calls.append("all_reduce"), orig_reduce(*a, **k))[1]
    drv.main(["--env_json", envf, "--agent_json", os.path.join(ROOT, "jsonfiles/agent/ddpg.json"),
              "--indices", "0", "7", "21", "--save_dir", os.path.join(tmp, "res"), "--quiet"])
    with open(os.path.join(tmp, "calls.json"), "w") as f:
        json.dump(calls, f)


def test_forced_dist_at_world_size_one_runs_the_collectives(tmp_path):
    """RLC_FORCE_DIST=1 (the one-GPU RCCL rehearsal, tests/test_gpu_rccl.py) under gloo: one rank, real collectives"""
    import json
    import pickle
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_main_worker_forced, args=(37500 + (os.getpid() % 2000), str(tmp_path)))
    p.start()
    p.join(120)
    assert p.exitcode == 0
    calls = json.load(open(tmp_path / "calls.json"))
    assert calls.count("all_gather") == 1 and calls.count("all_reduce") >= 2
    with open(tmp_path / "res" / "Pendulum-v0_ddpgresults" / "data_0_7_21.pkl", "rb") as f:
        data = pickle.load(f)
    assert sorted(data["experiment_data"]) == [0, 7, 14]
