"""Train / evaluate loop (mirrors experiment.py:14-217 of the reference).

Semantics kept:
  * one evaluation before any training (step 0), then one every ``eval_interval`` TRAINING steps,
    run IN THE MIDDLE of the training episode -- which calls ``agent.reset()`` and therefore resets
    the OU noise mid-episode (quirk Q8);
  * a step that ends the episode exactly at EPISODE_STEPS_LIMIT is "truncated": the agent's
    ``update`` is still called (so ``learn()`` runs) but the transition is not stored (Q7);
    environments whose name starts with 'Bimodal1DEnv' are exempt;
  * an episode cut by TOTAL_STEPS_LIMIT is not recorded in the per-episode lists;
  * ``run()`` returns the same 9-tuple as the reference (experiment.py:96-98).
TensorBoard summaries (write_log) are not produced: TensorFlow is not a dependency here.
"""
import time
from datetime import datetime

import numpy as np  # noqa: F401


class Experiment(object):
    def __init__(self, agent, train_environment, test_environment, seed, writer=None, write_log=False,
                 write_plot=False, verbose=True):
        self.agent = agent
        self.train_environment = train_environment
        self.train_environment.set_random_seed(seed)
        self.test_environment = test_environment
        self.test_environment.set_random_seed(seed)

        self.train_rewards_per_episode = []
        self.train_cum_steps = []
        self.train_episodes = 0
        self.timesteps_at_eval = []
        self.train_steps_per_episode = []
        self.eval_steps_per_episode = []
        self.eval_rewards_per_episode = []

        self.total_step_count = 0
        self.writer = writer
        self.write_log = write_log
        self.write_plot = write_plot
        self.verbose = verbose

        self.cum_train_time = 0.0
        self.cum_eval_time = 0.0

    def _say(self, msg):
        if self.verbose:
            print(msg)

    def run(self):
        episode_count = 0
        start_run = datetime.now()
        self._say("Start run at: " + str(start_run) + '\n')

        self.cum_eval_time += self.eval()
        self.timesteps_at_eval.append(self.total_step_count)

        while self.total_step_count < self.train_environment.TOTAL_STEPS_LIMIT:
            t0 = time.time()
            episode_reward, num_steps, force_terminated, eval_session_time = self.run_episode_train(is_train=True)
            train_ep_time = time.time() - t0 - eval_session_time
            self.cum_train_time += train_ep_time
            self._say("Train:: ep: " + str(episode_count) + ", r: " + str(episode_reward) + ", n_steps: "
                      + str(num_steps) + ", elapsed: " + time.strftime("%H:%M:%S", time.gmtime(train_ep_time)))
            if not force_terminated:
                self.train_rewards_per_episode.append(episode_reward)
                self.train_cum_steps.append(self.total_step_count)
                self.train_steps_per_episode.append(num_steps)
            episode_count += 1

        self.train_environment.close()
        end_run = datetime.now()
        self._say("End run at: " + str(end_run) + '\n')
        self._say("Total Time taken: " + str(end_run - start_run) + '\n')
        self._say("Training Time: " + time.strftime("%H:%M:%S", time.gmtime(self.cum_train_time)))
        self._say("Evaluation Time: " + time.strftime("%H:%M:%S", time.gmtime(self.cum_eval_time)))

        return (self.train_rewards_per_episode, self.eval_rewards_per_episode,
                self.train_steps_per_episode, self.eval_steps_per_episode,
                self.timesteps_at_eval, self.cum_train_time, self.cum_eval_time, self.train_episodes,
                self.train_cum_steps)

    def run_episode_train(self, is_train):
        env = self.train_environment
        self.train_episodes += 1
        eval_session_time = 0.0

        obs = env.reset()
        self.agent.reset()

        episode_reward = 0.
        done = False
        action = self.agent.start(obs, is_train)
        episode_step_count = 0

        while not (done or episode_step_count == env.EPISODE_STEPS_LIMIT
                   or self.total_step_count == env.TOTAL_STEPS_LIMIT):
            episode_step_count += 1
            self.total_step_count += 1

            obs_n, reward, done, info = env.step(action)
            episode_reward += reward

            if env.name.startswith('Bimodal1DEnv'):
                is_truncated = False
            else:
                is_truncated = bool(done and episode_step_count == env.EPISODE_STEPS_LIMIT)

            self.agent.update(obs, obs_n, float(reward), action, done, is_truncated)

            if not done:
                action = self.agent.step(obs_n, is_train)
            obs = obs_n

            if self.total_step_count % env.eval_interval == 0:
                self.timesteps_at_eval.append(self.total_step_count)
                eval_session_time += self.eval()

        force_terminated = not (done or episode_step_count == env.EPISODE_STEPS_LIMIT)
        return episode_reward, episode_step_count, force_terminated, eval_session_time

    def eval(self):
        rewards, steps = [], []
        eval_session_time = 0.0
        for i in range(self.test_environment.eval_episodes):
            t0 = time.time()
            episode_reward, num_steps = self.run_episode_eval(self.test_environment, is_train=False)
            elapsed = time.time() - t0
            steps.append(num_steps)
            rewards.append(episode_reward)
            eval_session_time += elapsed
            self._say("=== EVAL :: ep: " + str(i) + ", r: " + str(episode_reward) + ", n_steps: " + str(num_steps)
                      + ", elapsed: " + time.strftime("%H:%M:%S", time.gmtime(elapsed)))
        self.eval_rewards_per_episode.append(rewards)
        self.eval_steps_per_episode.append(steps)
        self.cum_eval_time += eval_session_time      # (the reference also adds it again in run(): kept)
        return eval_session_time

    def run_episode_eval(self, test_env, is_train):
        obs = test_env.reset()
        self.agent.reset()
        episode_reward = 0.
        done = False
        action = self.agent.start(obs, is_train)
        episode_step_count = 0
        while not (done or episode_step_count == test_env.EPISODE_STEPS_LIMIT):
            obs_n, reward, done, info = test_env.step(action)
            episode_reward += reward
            if not done:
                action = self.agent.step(obs_n, is_train)
            obs = obs_n
            episode_step_count += 1
        return episode_reward, episode_step_count
