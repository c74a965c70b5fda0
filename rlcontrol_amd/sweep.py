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


# ---- complete run records (every array field of main.py's run_data), for the multi-process driver -------------
_SCALARS = ("random_seed", "total_train_episodes", "eval_time", "train_time")


def full_vec_len(eval_shape, max_train_eps):
    return 3 + len(_SCALARS) + 2 * eval_shape[0] * eval_shape[1] + eval_shape[0] + 2 * max_train_eps


def pack_full_run(index, run_data, eval_shape, max_train_eps):
    """run_data (main.py:188-209 schema) -> fixed-length float64 vector; NaN = unused slot."""
    n_ev = eval_shape[0] * eval_shape[1]
    ev = np.asarray(run_data["eval_episode_rewards"], np.float64).reshape(-1)
    evs = np.asarray(run_data["eval_episode_steps"], np.float64).reshape(-1)
    tr = np.asarray(run_data["train_episode_rewards"], np.float64).reshape(-1)
    trs = np.asarray(run_data["train_episode_steps"], np.float64).reshape(-1)
    tae = np.asarray(run_data["timesteps_at_eval"], np.float64).reshape(-1)       # the values the run RECORDED
    if ev.size > n_ev or tr.size > max_train_eps or tae.size > eval_shape[0]:
        raise ValueError("run %d does not fit the exchange record (%d evals, %d train episodes)" % (index, ev.size, tr.size))
    vec = np.full(full_vec_len(eval_shape, max_train_eps), np.nan)
    vec[0], vec[1], vec[2] = index, ev.size, tr.size
    o = 3
    for k in _SCALARS:
        vec[o] = float(run_data[k]); o += 1
    vec[o:o + ev.size] = ev; o += n_ev
    vec[o:o + evs.size] = evs; o += n_ev
    vec[o:o + tae.size] = tae; o += eval_shape[0]
    vec[o:o + tr.size] = tr; o += max_train_eps
    vec[o:o + trs.size] = trs
    return vec


def unpack_full_run(vec, eval_shape, max_train_eps):
    """-> (index, dict with the array / scalar fields pack_full_run carried)"""
    n_ev = eval_shape[0] * eval_shape[1]
    k_ev, k_tr = int(vec[1]), int(vec[2])
    out, o = {}, 3
    for k in _SCALARS:
        out[k] = float(vec[o]) if k.endswith("_time") else int(vec[o]); o += 1
    rows = k_ev // eval_shape[1] if eval_shape[1] else 0
    out["eval_episode_rewards"] = vec[o:o + k_ev].reshape(rows, eval_shape[1]).copy(); o += n_ev
    out["eval_episode_steps"] = vec[o:o + k_ev].reshape(rows, eval_shape[1]).astype(np.int64); o += n_ev
    tae = vec[o:o + eval_shape[0]]; o += eval_shape[0]
    out["timesteps_at_eval"] = tae[~np.isnan(tae)].astype(np.int64)
    out["train_episode_rewards"] = vec[o:o + k_tr].copy(); o += max_train_eps
    out["train_episode_steps"] = vec[o:o + k_tr].astype(np.int64)
    return int(vec[0]), out


def all_reduce_max(value, device=None):
    """max of an integer over the ranks (sizes the exchange record)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):       # a group of ONE rank still runs the collective
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


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
    if not (dist.is_available() and dist.is_initialized()):
        return buf.cpu().numpy()
    out = [torch.empty_like(buf) for _ in range(dist.get_world_size())]
    dist.all_gather(out, buf)
    return torch.cat(out, 0).cpu().numpy()
