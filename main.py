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
    (episode_rewards, eval_episode_rewards, train_episode_steps, eval_episode_steps, timesteps_at_eval,
     train_time, eval_time, train_ep, _) = experiment.run()

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
    data["experiment_data"][sweep]["runs"].append(run_data)
    return run_data


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

    for index in range(args.indices[0], args.indices[2], args.indices[1]):
        run_index(index, agent_json, env_json, train_env, test_env, env_params, arg_params, data,
                  verbose=not args.quiet)
        os.makedirs(save_dir, exist_ok=True)
        save_file = save_dir + "data_%d_%d_%d.pkl" % tuple(args.indices)
        with open(save_file, "wb") as out_file:
            pickle.dump(data, out_file)
    return data


if __name__ == '__main__':
    main()
