"""CartPole-v0 / v1, restated: the closed-form environment of BASELINE config C3.

``gymnasium`` is absent from this image, so its classic-control CartPole is restated from memory
[unverifiable here]: cart mass 1.0, pole mass 0.1, half-length 0.5, force 10 N, tau 0.02 s, explicit
Euler integration, failure at |x| > 2.4 or |theta| > 12 degrees, reward 1 per step, reset state uniform in
[-0.05, 0.05]^4, plus the ``TimeLimit`` truncation (200 steps for v0, 500 for v1).  The functional pin is the
reference's own test (tests/agents/tree_search/test_mcts.py:5-19): UCT with budget 400 keeps the pole up for
all 200 steps -- checked for the reference planner on this env (tests/test_oracle_golden.py) and for the
device planner (tests/test_gpu_cartpole.py).

The object is an ordinary deep-copyable gym-style env (5-tuple ``step``), so the reference's planners run on
it unmodified; the device planners read its ``cartpole_params()`` / ``state`` / ``steps`` instead of copying it.
"""
import math

import numpy as np

from .finite_mdp import Discrete


class CartPoleEnv(object):
    metadata = {}

    def __init__(self, max_episode_steps=200):
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5                                   # half the pole's length
        self.polemass_length = self.masspole * self.length
        self.force_mag = 10.0
        self.tau = 0.02
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        self.max_episode_steps = int(max_episode_steps)
        self.action_space = Discrete(2)
        self.observation_space = None
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        self.state = None
        self.steps = 0
        self.steps_beyond_terminated = None

    @property
    def unwrapped(self):
        return self

    def cartpole_params(self):
        """The constants a closed-form clone of this env needs (mp_cartpole_params)."""
        return dict(gravity=self.gravity, masscart=self.masscart, masspole=self.masspole, length=self.length,
                    force_mag=self.force_mag, tau=self.tau, theta_threshold=self.theta_threshold_radians,
                    x_threshold=self.x_threshold, max_steps=self.max_episode_steps, euler=1)

    def seed(self, seed=None):
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        self.state = tuple(float(v) for v in self.np_random.uniform(low=-0.05, high=0.05, size=(4,)))
        self.steps = 0
        self.steps_beyond_terminated = None
        return np.array(self.state, dtype=np.float32), {}

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot ** 2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        terminated = bool(x < -self.x_threshold or x > self.x_threshold
                          or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)
        if not terminated:
            reward = 1.0
        elif self.steps_beyond_terminated is None:
            self.steps_beyond_terminated = 0
            reward = 1.0
        else:
            self.steps_beyond_terminated += 1
            reward = 0.0
        self.steps += 1
        truncated = self.max_episode_steps > 0 and self.steps >= self.max_episode_steps
        return np.array(self.state, dtype=np.float32), reward, terminated, truncated, {}

    def render(self, *a, **k):
        return None

    def close(self):
        pass
