#!/usr/bin/env python3
"""RLControl-style driver for the MI355X build (mirrors main.py:31-251 of the reference).

    python3 main.py --env_json jsonfiles/environment/Pendulum-v0.json \
                    --agent_json jsonfiles/agent/ddpg.json --indices START STEP STOP [--save_dir DIR]

Same CLI, same json surface, same sweep semantics (INDEX -> setting = INDEX % n_settings, run = seed =
INDEX // n_settings, utils/main_utils.py:92-99 + main.py:133-141) and the same result pickle
``<save_dir>/<env>_<agent>results/data_<START>_<STEP>_<STOP>.pkl`` re-written after every index
(main.py:80-95,188-209; the reference's swapped eval_time/train_time keys are reproduced so the
offline tooling reads identical fields).  Differences: no TensorFlow summary writer (writer=None),
and ``--gpu`` picks the HIP device.

``--device_rollout`` runs ALL indices of the range concurrently as one population on the GPU with the
environment simulated on the device (rlcontrol_amd/device_experiment.py; DDPG, SoftActorCritic or NAF on Pendulum-v0):
same schedule, same pickle, Philox random streams instead of numpy's.

Under ``python -m torch.distributed.run --nproc-per-node N main.py ...`` the INDEX range is dealt round-robin to
the N ranks (one GPU each, nothing exchanged while training); one all-gather of the run records at the end
(RCCL on GPUs, rlcontrol_amd/sweep.py) and rank 0 writes the single pickle; until then every rank keeps a
reference-schema pickle of its own shard up to date (``data_<..>.rank<r>of<N>.pkl``, re-written after every index).
"""
import argparse
import json
import os
import pickle
from collections import OrderedDict

import numpy as np

import rlcontrol_amd.environments.environments as envs
from rlcontrol_amd.experiment import Experiment
from rlcontrol_amd.utils.config import Config
from rlcontrol_amd.utils.main_utils import create_agent, get_sweep_parameters


def run_index(index, agent_json, env_json, train_env, test_env, env_params, arg_params, data, verbose=True):
    agent_params, total_num_sweeps = get_sweep_parameters(agent_json['sweeps'], index)
    sweep = index % total_num_sweeps
    if sweep not in data["experiment_data"]:
        data["experiment_data"][sweep] = {"agent_params": dict(agent_params), "runs": []}

    run_num = int(index / total_num_sweeps)
    random_seed = run_num
    arg_params = dict(arg_params, random_seed=random_seed)
    if verbose:
        print("Total HP settings: %d" % total_num_sweeps)
        print("SETTING_NUM: %d\nRUN_NUM: %d\nRANDOM_SEED: %d" % (sweep, run_num, random_seed))
        print('Agent setting: ', agent_params)
    agent_params["writer"] = None

    config = Config()
    config.merge_config(env_params)
    config.merge_config(agent_params)
    config.merge_config(arg_params)

    agent = create_agent(agent_json['agent'], config)
    experiment = Experiment(agent=agent, train_environment=train_env, test_environment=test_env, seed=random_seed,
                            writer=None, write_log=arg_params["write_log"], write_plot=arg_params["write_plot"],
                            verbose=verbose)
    run_data = _run_data(random_seed, env_json, experiment.run())
    data["experiment_data"][sweep]["runs"].append(run_data)
    return run_data


def _run_data(random_seed, env_json, result):
    (episode_rewards, eval_episode_rewards, train_episode_steps, eval_episode_steps, timesteps_at_eval,
     train_time, eval_time, train_ep, _) = result
    run_data = {"random_seed": random_seed}
    run_data["total_timesteps"] = env_json["TotalMilSteps"] * 1000000
    run_data["eval_interval_timesteps"] = env_json["EvalIntervalMilSteps"] * 1000000
    run_data["episodes_per_eval"] = env_json["EvalEpisodes"]
    run_data["eval_episode_rewards"] = np.array(eval_episode_rewards)
    run_data["eval_episode_steps"] = np.array(eval_episode_steps)
    run_data["timesteps_at_eval"] = np.array(timesteps_at_eval)
    run_data["train_episode_steps"] = np.array(train_episode_steps)
    run_data["train_episode_rewards"] = np.array(episode_rewards)
    run_data["total_train_episodes"] = train_ep
    run_data["eval_time"] = train_time      # sic: swapped in the reference (main.py:200-201)
    run_data["train_time"] = eval_time
    return run_data


# Config attributes that must be equal across the agents of one device population (everything except the
# per-agent learning rates / entropy scale and the seed)
_SHARED_KEYS = {
    "DDPG": ("shared_l1_dim", "actor_l2_dim", "critic_l2_dim", "batch_size", "buffer_size", "tau", "gamma",
             "warmup_steps", "norm_type", "network", "exploration_policy", "ou_theta", "ou_mu", "ou_sigma"),
    "SoftActorCritic": ("actor_l1_dim", "actor_l2_dim", "critic_l1_dim", "critic_l2_dim", "batch_size", "buffer_size",
                        "tau", "gamma", "warmup_steps", "norm_type", "exploration_policy", "sample_for_eval"),
    "NAF": ("l1_dim", "l2_dim", "batch_size", "buffer_size", "tau", "gamma", "warmup_steps", "norm_type",
            "exploration_policy"),
    "ReverseKL": ("actor_l1_dim", "actor_l2_dim", "critic_l1_dim", "critic_l2_dim", "batch_size", "buffer_size", "tau",
                  "gamma", "warmup_steps", "exploration_policy", "sample_for_eval", "N_param", "l_param", "optim_type",
                  "q_update_type", "use_true_q"),
}
_SHARED_KEYS["ForwardKL"] = _SHARED_KEYS["ReverseKL"]


def _make_population(agent_name, members, arg_params):
    """One population handle for a group of (index, sweep, agent_params, config) that share the network shape."""
    c0 = members[0][3]
    seeds = [np.uint64(m[3].random_seed) for m in members]
    device = int(arg_params.get("device", 0))
    # the on-device loop implements the shipped jsons' input handling only; a norm_type it does not apply must never be
    # reported in the pickle's agent_params as if it had run (the reference applies layer / batch norm for those values,
    # agents/network/base_network.py:53-65)
    from rlcontrol_amd.agents.network.base_network_manager import check_norm_type
    if agent_name in ("ReverseKL", "ForwardKL"):
        # the torch networks of these agents never apply the normaliser they are handed (reversekl_network.py:43)
        from rlcontrol_amd.hip_kl import KLPopulation, init_params
        if c0.exploration_policy != 'none' or c0.sample_for_eval == "True" or c0.use_true_q == "True":
            raise RuntimeError("the device loop implements the KL agents with exploration_policy 'none', "
                               "sample_for_eval 'False' and use_true_q 'False' (the shipped jsons)")
        pop = KLPopulation(
            "reverse" if agent_name == "ReverseKL" else "forward", n_agents=len(members), state_dim=c0.state_dim,
            action_dim=c0.action_dim, actor_l1_dim=c0.actor_l1_dim, actor_l2_dim=c0.actor_l2_dim,
            critic_l1_dim=c0.critic_l1_dim, critic_l2_dim=c0.critic_l2_dim, batch_size=c0.batch_size,
            buffer_size=int(c0.buffer_size), tau=c0.tau, action_max0=float(np.asarray(c0.action_max).reshape(-1)[0]),
            pi_lr=[m[3].pi_lr for m in members], qf_vf_lr=[m[3].qf_vf_lr for m in members],
            entropy_scale=[m[3].entropy_scale for m in members], seeds=seeds, n_param=c0.N_param,
            optim_type=c0.optim_type, q_update_type=c0.q_update_type, device=device,
            l_param=getattr(c0, "l_param", None), action_max=c0.action_max)
        for i, m in enumerate(members):
            pop.set_params(i, init_params(c0.state_dim, c0.action_dim, c0.actor_l1_dim, c0.actor_l2_dim, c0.critic_l1_dim,
                                          c0.critic_l2_dim, m[3].random_seed), init_target=True)
        return pop
    check_norm_type(c0, agent_name + " --device_rollout",
                    ('input_norm', 'layer') if agent_name == "SoftActorCritic" else ('none', 'input_norm', 'layer'))
    if agent_name == "DDPG":
        from rlcontrol_amd.hip_ddpg import DDPGPopulation, init_params
        if c0.exploration_policy != 'ou_noise':
            raise RuntimeError("the device loop implements DDPG's 'ou_noise' exploration policy only")
        separate = getattr(c0, "network", "hydra") == "separate"
        pop = DDPGPopulation(
            n_agents=len(members), state_dim=c0.state_dim, action_dim=c0.action_dim, shared_l1_dim=c0.shared_l1_dim,
            actor_l2_dim=c0.actor_l2_dim, critic_l2_dim=c0.critic_l2_dim, batch_size=c0.batch_size,
            buffer_size=int(c0.buffer_size), tau=c0.tau, state_min=c0.state_min, state_max=c0.state_max,
            action_min=c0.action_min, action_max=c0.action_max, actor_lr=[m[3].actor_lr for m in members],
            critic_lr=[m[3].critic_lr for m in members], seeds=seeds, clip_state=(c0.norm_type != 'none'),
            ou_theta=c0.ou_theta, ou_mu=c0.ou_mu, ou_sigma=c0.ou_sigma, device=device, norm_type=c0.norm_type,
            separate_networks=separate)
        for i, m in enumerate(members):
            pop.set_params(i, init_params(c0.state_dim, c0.action_dim, c0.shared_l1_dim, c0.actor_l2_dim,
                                          c0.critic_l2_dim, m[3].random_seed, c0.norm_type, separate), init_target=True)
        return pop
    if agent_name == "NAF":
        from rlcontrol_amd.hip_naf import NAFPopulation, init_params
        if c0.exploration_policy != 'none':
            raise RuntimeError("the device loop implements NAF's own covariance exploration (exploration_policy 'none')")
        pop = NAFPopulation(
            n_agents=len(members), state_dim=c0.state_dim, action_dim=c0.action_dim, l1_dim=c0.l1_dim, l2_dim=c0.l2_dim,
            batch_size=c0.batch_size, buffer_size=int(c0.buffer_size), tau=c0.tau, state_min=c0.state_min,
            state_max=c0.state_max, action_max=c0.action_max, learning_rate=[m[3].learning_rate for m in members],
            seeds=seeds, clip_state=(c0.norm_type != 'none'), device=device, norm_type=c0.norm_type,
            action_min=c0.action_min)       # the draw is clipped to [action_min, action_max] (naf_network.py:176)
        for i, m in enumerate(members):
            pop.set_params(i, init_params(c0.state_dim, c0.action_dim, c0.l1_dim, c0.l2_dim, m[3].random_seed,
                                          c0.norm_type), init_target=True)
        return pop
    from rlcontrol_amd.hip_sac import SACPopulation, init_params
    if c0.exploration_policy != 'none' or c0.sample_for_eval == "True" or c0.norm_type == 'none':
        raise RuntimeError("the device loop implements SoftActorCritic with exploration_policy 'none', "
                           "sample_for_eval 'False' and an input norm (the shipped sac.json)")
    pop = SACPopulation(
        n_agents=len(members), state_dim=c0.state_dim, action_dim=c0.action_dim, actor_l1_dim=c0.actor_l1_dim,
        actor_l2_dim=c0.actor_l2_dim, critic_l1_dim=c0.critic_l1_dim, critic_l2_dim=c0.critic_l2_dim,
        batch_size=c0.batch_size, buffer_size=int(c0.buffer_size), tau=c0.tau,
        state_min0=float(np.asarray(c0.state_min).reshape(-1)[0]), state_max0=float(np.asarray(c0.state_max).reshape(-1)[0]),
        action_max0=float(np.asarray(c0.action_max).reshape(-1)[0]), pi_lr=[m[3].pi_lr for m in members],
        qf_vf_lr=[m[3].qf_vf_lr for m in members], entropy_scale=[m[3].entropy_scale for m in members], seeds=seeds,
        clip_state=True, device=device, norm_type=c0.norm_type)
    for i, m in enumerate(members):
        pop.set_params(i, init_params(c0.state_dim, c0.action_dim, c0.actor_l1_dim, c0.actor_l2_dim, c0.critic_l1_dim,
                                      c0.critic_l2_dim, m[3].random_seed, c0.norm_type), init_target=True)
    return pop


def run_indices_on_device(indices, agent_json, env_json, env_params, arg_params, data, verbose=True, progress=None):
    """All `indices` at once: one population per group of indices that share the network / replay shape."""
    from rlcontrol_amd.device_experiment import DeviceExperiment
    agent_name = agent_json['agent']
    if agent_name not in _SHARED_KEYS:
        raise RuntimeError("--device_rollout is built for the DDPG, SoftActorCritic, NAF, ReverseKL and ForwardKL agents "
                           "(got %r)" % agent_name)
    groups = OrderedDict()
    for index in indices:
        agent_params, total_num_sweeps = get_sweep_parameters(agent_json['sweeps'], index)
        config = Config()
        config.merge_config(env_params)
        config.merge_config(agent_params)
        config.merge_config(dict(arg_params, random_seed=int(index / total_num_sweeps)))
        key = tuple(str(getattr(config, k, None)) for k in _SHARED_KEYS[agent_name])
        groups.setdefault(key, []).append((index, index % total_num_sweeps, dict(agent_params), config))
    out = {}
    for members in groups.values():
        c0 = members[0][3]
        pop = _make_population(agent_name, members, arg_params)
        exp = DeviceExperiment(pop, env_json, gamma=c0.gamma, warmup_steps=c0.warmup_steps,
                               noise_scale=[m[3].noise_scale for m in members] if agent_name == "NAF" else None)
        if verbose:
            print("device rollout: %d agents, %d steps each" % (len(members), exp.total_steps_limit))
        results = exp.run(progress=progress)
        for m, res in zip(members, results):
            out[m[0]] = (m[1], m[2], _run_data(m[3].random_seed, env_json, res))
        pop.close()
    for index in indices:            # pickle layout: runs appended in index order, as the sequential driver does
        sweep, agent_params, run_data = out[index]
        if sweep not in data["experiment_data"]:
            data["experiment_data"][sweep] = {"agent_params": agent_params, "runs": []}
        data["experiment_data"][sweep]["runs"].append(run_data)
    return data


def new_data_dict(agent_json, env_json):
    data = {"experiment": {"environment": {}, "agent": {}}, "experiment_data": {}}
    data["experiment"]["agent"]["agent_name"] = agent_json["agent"]
    data["experiment"]["agent"]["parameters"] = dict(agent_json["sweeps"])
    data["experiment"]["environment"]["env_name"] = env_json["environment"]
    data["experiment"]["environment"]["total_timesteps"] = env_json["TotalMilSteps"] * 1000000
    data["experiment"]["environment"]["steps_per_episode"] = env_json["EpisodeSteps"]
    data["experiment"]["environment"]["eval_interval_timesteps"] = env_json["EvalIntervalMilSteps"] * 1000000
    data["experiment"]["environment"]["eval_episodes"] = env_json["EvalEpisodes"]
    return data


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--env_json', type=str)
    parser.add_argument('--agent_json', type=str)
    parser.add_argument('--indices', type=int, nargs=3)
    parser.add_argument('--monitor', default=False, action='store_true')
    parser.add_argument('--render', default=False, action='store_true')
    parser.add_argument('--write_log', default=False, action='store_true')
    parser.add_argument('--write_plot', default=False, action='store_true')
    parser.add_argument('--save_dir', default="./results")
    parser.add_argument('--gpu', type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    parser.add_argument('--quiet', default=False, action='store_true')
    parser.add_argument('--device_rollout', default=False, action='store_true',
                        help="run all indices concurrently on the GPU with the environment on the device")
    args = parser.parse_args(argv)

    arg_params = {"write_log": args.write_log, "write_plot": args.write_plot, "device": args.gpu}

    # str.rstrip(".json") as in the reference (main.py:53-54); fine for the shipped names
    env_name = os.path.basename(args.env_json).rstrip(".json")
    agent_name = os.path.basename(args.agent_json).rstrip(".json")

    with open(args.env_json, 'r') as f:
        env_json = json.load(f, object_pairs_hook=OrderedDict)
    with open(args.agent_json, 'r') as f:
        agent_json = json.load(f, object_pairs_hook=OrderedDict)

    train_env = envs.create_environment(env_json)
    test_env = envs.create_environment(env_json)
    env_params = {
        "env_name": train_env.name,
        "state_dim": train_env.state_dim, "state_min": train_env.state_min, "state_max": train_env.state_max,
        "action_dim": train_env.action_dim, "action_min": train_env.action_min, "action_max": train_env.action_max,
    }
    data = new_data_dict(agent_json, env_json)
    save_dir = args.save_dir + "/" + env_name + "_" + agent_name + 'results/'

    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RLC_FORCE_DIST=1 takes the multi-process path at world size 1 too (one rank under torch.distributed.run): the
    # barrier, the MAX all-reduce and the all-gather then really go through RCCL -- the one-GPU rehearsal of the
    # 8-GPU INDEX sweep (tests/test_gpu_rccl.py)
    if world > 1 or (os.environ.get("RLC_FORCE_DIST", "0") == "1" and "RANK" in os.environ):
        return main_distributed(args, agent_json, env_json, train_env, test_env, env_params, arg_params, save_dir)

    if args.device_rollout:
        run_indices_on_device(list(range(args.indices[0], args.indices[2], args.indices[1])), agent_json, env_json,
                              env_params, arg_params, data, verbose=not args.quiet)
        os.makedirs(save_dir, exist_ok=True)
        with open(save_dir + "data_%d_%d_%d.pkl" % tuple(args.indices), "wb") as out_file:
            pickle.dump(data, out_file)
        return data

    for index in range(args.indices[0], args.indices[2], args.indices[1]):
        run_index(index, agent_json, env_json, train_env, test_env, env_params, arg_params, data,
                  verbose=not args.quiet)
        os.makedirs(save_dir, exist_ok=True)
        save_file = save_dir + "data_%d_%d_%d.pkl" % tuple(args.indices)
        with open(save_file, "wb") as out_file:
            pickle.dump(data, out_file)
    return data


def main_distributed(args, agent_json, env_json, train_env, test_env, env_params, arg_params, save_dir):
    """One process per GPU: this rank's shard of the INDEX range, then ONE all-gather of the run records."""
    import torch
    import torch.distributed as dist
    from rlcontrol_amd import sweep
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    on_gpu = torch.cuda.is_available()
    device = None
    if on_gpu:
        torch.cuda.set_device(args.gpu)
        device = torch.device("cuda", args.gpu)
    if not dist.is_initialized():
        dist.init_process_group("nccl" if on_gpu else "gloo", **({"device_id": device} if on_gpu else {}))
    if os.environ.get("RLC_LOG_COLLECTIVES", "0") == "1" and rank == 0:
        print("main_distributed: backend=%s world=%d" % (dist.get_backend(), world), flush=True)
    mine = sweep.rank_indices(args.indices[0], args.indices[1], args.indices[2], rank, world)
    local = new_data_dict(agent_json, env_json)
    verbose = (not args.quiet) and rank == 0
    n_settings = get_sweep_parameters(agent_json['sweeps'], 0)[1]
    total = int(env_json["TotalMilSteps"] * 1000000)
    interval = int(env_json["EvalIntervalMilSteps"] * 1000000)
    eval_shape = (total // interval + 1, int(env_json["EvalEpisodes"]))
    runs, failure = {}, None
    # Every rank re-writes a reference-schema pickle of ITS shard after each index it completes, as the reference
    # re-pickles after every index (main.py:205-209): a rank that dies late loses nothing that finished, and the shard
    # files merge offline exactly as the reference merges its per-process pickles (main_concurrent.py:107-154
    # combine_data_dictionaries).  Rank 0 removes them once the one merged pickle is on disk.
    shard_file = save_dir + "data_%d_%d_%d.rank%dof%d.pkl" % (tuple(args.indices) + (rank, world))

    def write_shard():
        os.makedirs(save_dir, exist_ok=True)
        with open(shard_file + ".tmp", "wb") as out_file:
            pickle.dump(local, out_file)
        os.replace(shard_file + ".tmp", shard_file)

    try:
        if args.device_rollout:
            if mine:
                run_indices_on_device(mine, agent_json, env_json, env_params, arg_params, local, verbose=verbose)
                write_shard()          # the device loop finishes all of the rank's indices together
        else:
            for index in mine:
                run_index(index, agent_json, env_json, train_env, test_env, env_params, arg_params, local, verbose=verbose)
                write_shard()
        # local runs in index order (each setting's list is in increasing index = increasing seed)
        for sweep_id, sd in local["experiment_data"].items():
            for rd in sd["runs"]:
                runs[rd["random_seed"] * n_settings + sweep_id] = rd
                if np.asarray(rd["eval_episode_rewards"]).size > eval_shape[0] * eval_shape[1]:
                    raise ValueError("run %d holds more evaluations than the exchange record" % rd["random_seed"])
    except Exception as e:           # a failing rank still reaches the collectives below, carrying an error flag:
        failure = e                  # the other ranks must not sit in the all-gather until torchrun tears the job down
        import traceback
        traceback.print_exc()
    if sweep.all_reduce_max(1 if failure is not None else 0, device):
        dist.barrier()
        dist.destroy_process_group()
        raise RuntimeError("rank %d: %s" % (rank, "this rank failed: %r" % (failure,) if failure is not None
                                            else "another rank failed; no pickle written"))
    max_tr = sweep.all_reduce_max(max([len(r["train_episode_rewards"]) for r in runs.values()] + [1]), device)
    vlen = sweep.full_vec_len(eval_shape, max_tr)
    vecs = [sweep.pack_full_run(i, runs[i], eval_shape, max_tr) for i in sorted(runs)]
    n_all = len(range(args.indices[0], args.indices[2], args.indices[1]))
    gathered = sweep.all_gather_runs(vecs, (n_all + world - 1) // world, vlen, device)
    data = new_data_dict(agent_json, env_json)
    if rank == 0:
        got = {}
        for row in gathered:
            if not np.isnan(row[0]):
                idx, fields = sweep.unpack_full_run(row, eval_shape, max_tr)
                got[idx] = fields
        for index in range(args.indices[0], args.indices[2], args.indices[1]):
            agent_params, _ = get_sweep_parameters(agent_json['sweeps'], index)
            sweep_id = index % n_settings
            if sweep_id not in data["experiment_data"]:
                data["experiment_data"][sweep_id] = {"agent_params": dict(agent_params), "runs": []}
            f = got[index]
            rd = {"random_seed": f["random_seed"], "total_timesteps": env_json["TotalMilSteps"] * 1000000,
                  "eval_interval_timesteps": env_json["EvalIntervalMilSteps"] * 1000000,
                  "episodes_per_eval": env_json["EvalEpisodes"],
                  "eval_episode_rewards": f["eval_episode_rewards"], "eval_episode_steps": f["eval_episode_steps"],
                  "timesteps_at_eval": f["timesteps_at_eval"],
                  "train_episode_steps": f["train_episode_steps"], "train_episode_rewards": f["train_episode_rewards"],
                  "total_train_episodes": f["total_train_episodes"], "eval_time": f["eval_time"],
                  "train_time": f["train_time"]}
            data["experiment_data"][sweep_id]["runs"].append(rd)
        os.makedirs(save_dir, exist_ok=True)
        with open(save_dir + "data_%d_%d_%d.pkl" % tuple(args.indices), "wb") as out_file:
            pickle.dump(data, out_file)
    dist.barrier()
    if os.path.exists(shard_file):       # the merged pickle holds everything the shard files held
        os.remove(shard_file)
    return data


if __name__ == '__main__':
    main()
