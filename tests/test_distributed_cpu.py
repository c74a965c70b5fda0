"""N > 1 path on CPU: world_size 2, gloo.  Ranks own disjoint INDEX shards, exchange nothing while
"training" and meet in exactly one all_gather of packed episodic returns (rlcontrol_amd/sweep.py)."""
import os
import sys

import numpy as np
import torch
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
