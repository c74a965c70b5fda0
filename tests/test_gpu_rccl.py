"""The N > 1 command lines through RCCL on the one GPU of a box (BASELINE configs[4] rehearsal).

`bench.py --gpus N` and `main.py` under `python -m torch.distributed.run` are what the driver starts on an 8-GPU node:
one rank per GPU, `init_process_group("nccl")` (= RCCL), a barrier on both sides of the timed region, a MAX all-reduce of
the wall time and ONE all-gather of the per-rank results (the episodic-return all-gather of the INDEX sweep,
/root/reference main.py:111-141 + main_concurrent.py:107-154).  A one-GPU box cannot hold two RCCL ranks, so the same
command lines run here with ONE rank and the process group forced on (`--force-dist` / `RLC_FORCE_DIST=1`): every
collective call of the N > 1 path then goes through RCCL on a device tensor.  Each run is a FRESH process tree started
by the GPU-free launcher (tests/clean_launcher.py): nothing in it touches the GPU before `torch.cuda.set_device`.
"""
import json
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORCHRUN = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
            "--master-addr", "127.0.0.1"]
ENV = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "NCCL_DEBUG": "VERSION"}


@pytest.mark.gpu
def test_bench_one_rank_through_rccl(clean_launcher, tmp_path):
    port = 41000 + (os.getpid() % 2000)
    rep = clean_launcher(TORCHRUN + ["--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                                     "--backend", "nccl", "--force-dist", "--steps", "2", "--warmup", "1", "--agents", "64",
                                     "--updates-per-step", "16", "--no-cpu-baseline", "--no-side-records"],
                         env=ENV, timeout=900)
    assert rep["rc"] == 0, rep["stderr"][-3000:]
    line = [l for l in rep["stdout"].splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["collectives"].startswith("nccl:")            # the process group was RCCL, not skipped
    assert len(out["config"]["per_rank_result"]) == 1 and np.isfinite(out["config"]["per_rank_result"][0])
    assert out["config"]["kernel"] == "mfma"
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "rccl_bench_one_rank.json"), "w") as f:
            f.write(line + "\n")
        with open(os.path.join(keep, "rccl_bench_one_rank.stderr.txt"), "w") as f:
            f.write(rep["stderr"][-4000:])


@pytest.mark.gpu
def test_main_device_rollout_one_rank_through_rccl(clean_launcher, tmp_path):
    """main.py --device_rollout on 2 indices under torch.distributed.run: shard, MAX all-reduces, all-gather of the run
    records and rank 0's pickle, all on the RCCL process group."""
    env = {"environment": "Pendulum-v0", "TotalMilSteps": 0.00025, "EpisodeSteps": 100,
           "EvalIntervalMilSteps": 0.0001, "EvalEpisodes": 2}
    agent = {"agent": "DDPG", "sweeps": {"shared_l1_dim": [32], "actor_l2_dim": [32], "critic_l2_dim": [32],
                                         "actor_lr": [1e-3, 1e-4], "critic_lr": [1e-2], "norm_type": ["input_norm"],
                                         "exploration_policy": ["ou_noise"], "batch_size": [16],
                                         "buffer_size": [1000]}}
    ej, aj = tmp_path / "Pendulum-v0.json", tmp_path / "ddpg.json"
    ej.write_text(json.dumps(env)); aj.write_text(json.dumps(agent))
    port = 43000 + (os.getpid() % 2000)
    rep = clean_launcher(TORCHRUN + ["--master-port", str(port), os.path.join(ROOT, "main.py"), "--env_json", str(ej),
                                     "--agent_json", str(aj), "--indices", "0", "1", "2", "--save_dir", str(tmp_path),
                                     "--device_rollout", "--quiet"],
                         env=dict(ENV, RLC_FORCE_DIST="1", RLC_LOG_COLLECTIVES="1"), timeout=900)
    assert rep["rc"] == 0, rep["stderr"][-3000:]
    assert "backend=nccl" in rep["stdout"] + rep["stderr"]
    resdir = tmp_path / "Pendulum-v0_ddpgresults"
    assert sorted(os.listdir(resdir)) == ["data_0_1_2.pkl"]            # merged pickle only, shard file removed
    with open(resdir / "data_0_1_2.pkl", "rb") as f:
        data = pickle.load(f)
    assert sorted(data["experiment_data"]) == [0, 1]
    run = data["experiment_data"][1]["runs"][0]
    assert run["eval_episode_rewards"].shape == (3, 2) and run["timesteps_at_eval"].tolist() == [0, 100, 200]
    assert np.all(np.isfinite(run["eval_episode_rewards"]))
