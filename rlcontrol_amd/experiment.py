"""Host-driven train / evaluate schedule of one run (the role of experiment.py in the reference).

``Experiment(agent, train_environment, test_environment, seed, ...).run()`` returns the 9-tuple the reference's
driver unpacks: (train returns per episode, eval returns per evaluation, train steps per episode, eval steps per
evaluation, training-step counts at which evaluations ran, training wall time, evaluation wall time, number of
training episodes started, cumulative step count at the end of every recorded training episode).

Schedule (each point is asserted by tests/test_experiment_host.py):
  * evaluation number 0 runs before the first training step; afterwards one evaluation every ``eval_interval``
    TRAINING steps, wherever in a training episode that falls -- and since every evaluation episode calls
    ``agent.reset()``, the exploration noise of the interrupted training episode restarts (quirk Q8);
  * a step that ends an episode exactly at EPISODE_STEPS_LIMIT is "truncated": ``agent.update`` is still called
    (so ``learn()`` runs) with ``is_truncated=True`` and the agent drops the transition (quirk Q7); environments
    whose name starts with 'Bimodal1DEnv' are exempt from the truncation rule;
  * the episode that TOTAL_STEPS_LIMIT cuts short is counted as started but not recorded;
  * the next action is requested only if the episode goes on.
TensorBoard summaries (write_log) are not produced: TensorFlow is not a dependency here.
"""
import time
from datetime import datetime


def _hms(seconds):
    return time.strftime("%H:%M:%S", time.gmtime(seconds))


class Experiment(object):
    def __init__(self, agent, train_environment, test_environment, seed, writer=None, write_log=False,
                 write_plot=False, verbose=True):
        self.agent = agent
        self.train_environment, self.test_environment = train_environment, test_environment
        for env in (train_environment, test_environment):
            env.set_random_seed(seed)
        self.writer, self.write_log, self.write_plot, self.verbose = writer, write_log, write_plot, verbose

        self.total_step_count = 0
        self.train_episodes = 0
        self.train_rewards_per_episode, self.train_steps_per_episode, self.train_cum_steps = [], [], []
        self.eval_rewards_per_episode, self.eval_steps_per_episode, self.timesteps_at_eval = [], [], []
        self.cum_train_time = self.cum_eval_time = 0.0

    # ---------------------------------------------------------------------------------------------
    def _say(self, text):
        if self.verbose:
            print(text)

    def _budget_left(self):
        return self.total_step_count < self.train_environment.TOTAL_STEPS_LIMIT

    def run(self):
        began = datetime.now()
        self._say("Start run at: %s\n" % began)
        self._evaluation_point()
        episode = 0
        while self._budget_left():
            tick = time.time()
            ret, length, cut_short, eval_seconds = self.run_episode_train(is_train=True)
            spent = time.time() - tick - eval_seconds
            self.cum_train_time += spent
            self._say("Train:: ep: %d, r: %s, n_steps: %d, elapsed: %s" % (episode, ret, length, _hms(spent)))
            if not cut_short:
                self.train_rewards_per_episode.append(ret)
                self.train_steps_per_episode.append(length)
                self.train_cum_steps.append(self.total_step_count)
            episode += 1
        self.train_environment.close()
        ended = datetime.now()
        self._say("End run at: %s\n\nTotal Time taken: %s\n" % (ended, ended - began))
        self._say("Training Time: %s\nEvaluation Time: %s" % (_hms(self.cum_train_time), _hms(self.cum_eval_time)))
        return (self.train_rewards_per_episode, self.eval_rewards_per_episode, self.train_steps_per_episode,
                self.eval_steps_per_episode, self.timesteps_at_eval, self.cum_train_time, self.cum_eval_time,
                self.train_episodes, self.train_cum_steps)

    def _evaluation_point(self):
        """record where the evaluation happens, run it, return its wall time"""
        self.timesteps_at_eval.append(self.total_step_count)
        seconds = self.eval()
        self.cum_eval_time += seconds          # eval() adds it as well: the reference counts it twice, kept
        return seconds

    # ---------------------------------------------------------------------------------------------
    def run_episode_train(self, is_train):
        env, agent = self.train_environment, self.agent
        limit, budget = env.EPISODE_STEPS_LIMIT, env.TOTAL_STEPS_LIMIT
        exempt = env.name.startswith('Bimodal1DEnv')
        self.train_episodes += 1
        state = env.reset()
        agent.reset()
        action = agent.start(state, is_train)
        ret, length, done, eval_seconds = 0., 0, False, 0.0
        while not done and length != limit and self.total_step_count != budget:
            length += 1
            self.total_step_count += 1
            following, reward, done, _ = env.step(action)
            ret += reward
            hit_limit = (not exempt) and bool(done and length == limit)
            agent.update(state, following, float(reward), action, done, hit_limit)
            if not done:
                action = agent.step(following, is_train)
            state = following
            if self.total_step_count % env.eval_interval == 0:
                # timesteps_at_eval is appended before the evaluation, the double-counted time is not used here
                self.timesteps_at_eval.append(self.total_step_count)
                eval_seconds += self.eval()
        cut_short = not (done or length == limit)
        return ret, length, cut_short, eval_seconds

    # ---------------------------------------------------------------------------------------------
    def eval(self):
        env = self.test_environment
        returns, lengths, total = [], [], 0.0
        for number in range(env.eval_episodes):
            tick = time.time()
            ret, length = self.run_episode_eval(env, is_train=False)
            took = time.time() - tick
            total += took
            returns.append(ret)
            lengths.append(length)
            self._say("=== EVAL :: ep: %d, r: %s, n_steps: %d, elapsed: %s" % (number, ret, length, _hms(took)))
        self.eval_rewards_per_episode.append(returns)
        self.eval_steps_per_episode.append(lengths)
        self.cum_eval_time += total
        return total

    def run_episode_eval(self, test_env, is_train):
        state = test_env.reset()
        self.agent.reset()
        action = self.agent.start(state, is_train)
        ret, length, done = 0., 0, False
        while not done and length != test_env.EPISODE_STEPS_LIMIT:
            state, reward, done, _ = test_env.step(action)
            ret += reward
            length += 1
            if not done:
                action = self.agent.step(state, is_train)
        return ret, length
