"""Multi-GPU INDEX sweep: one process per GPU, independent shards, one all-gather at the end.

The reference "distributes" a sweep as unrelated OS processes whose result pickles are merged offline
(main.py:111-209, main_concurrent.py:107-154).  Here rank r of W runs INDEX values
START + (r + k*W)*STEP < STOP on GPU LOCAL_RANK; nothing is exchanged while training (no gradients are
shared between seeds).  When every rank is done, the per-run episodic-return arrays are exchanged with ONE
``all_gather`` (RCCL over xGMI on GPUs, gloo in the CPU test; ~10 KB per run: 201x10 eval returns +
~500 train-episode returns), and rank 0 writes the reference-schema pickle.
"""
import numpy as np


def rank_indices(start, step, stop, rank, world):
    """INDEX values of this rank: round-robin over range(start, stop, step)."""
    all_idx = list(range(start, stop, step))
    return all_idx[rank::world]


def pack_run(index, run_data, eval_shape, max_train_eps):
    """run_data dict (main.py schema) -> fixed-length float64 vector."""
    ev = np.asarray(run_data["eval_episode_rewards"], np.float64).reshape(-1)
    tr = np.asarray(run_data["train_episode_rewards"], np.float64).reshape(-1)
    n_ev = eval_shape[0] * eval_shape[1]
    vec = np.full(4 + n_ev + max_train_eps, np.nan)
    vec[0] = index
    vec[1] = run_data["random_seed"]
    vec[2] = min(ev.size, n_ev)
    vec[3] = min(tr.size, max_train_eps)
    vec[4:4 + int(vec[2])] = ev[:int(vec[2])]
    vec[4 + n_ev:4 + n_ev + int(vec[3])] = tr[:int(vec[3])]
    return vec


def unpack_run(vec, eval_shape, max_train_eps):
    n_ev = eval_shape[0] * eval_shape[1]
    k_ev, k_tr = int(vec[2]), int(vec[3])
    ev = vec[4:4 + k_ev]
    if k_ev == n_ev:
        ev = ev.reshape(eval_shape)
    return {"index": int(vec[0]), "random_seed": int(vec[1]), "eval_episode_rewards": ev.copy(),
            "train_episode_rewards": vec[4 + n_ev:4 + n_ev + k_tr].copy()}


def all_gather_runs(local_vectors, runs_per_rank, vec_len, device=None):
    """The one collective of the sweep.  local_vectors: list of packed runs of this rank (<= runs_per_rank).
    Returns the [world * runs_per_rank, vec_len] array of every rank's runs (NaN rows = unused slots)."""
    import torch
    import torch.distributed as dist
    buf = torch.full((runs_per_rank, vec_len), float("nan"), dtype=torch.float64)
    for i, v in enumerate(local_vectors):
        buf[i] = torch.from_numpy(np.asarray(v, np.float64))
    if device is not None:
        buf = buf.to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return buf.cpu().numpy()
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(out, buf)
    return torch.cat(out, 0).cpu().numpy()
