"""Value iteration on the MI355X planning core (reference
``rl_agents/agents/dynamic_programming/value_iteration.py``); the sweeps run in ``mp_vi_solve``
(rl_agents_amd/csrc/vi.hip)."""
import logging

import numpy as np

from rl_agents_amd import device_model
from rl_agents_amd.agents.common.abstract import StatelessPlannerAgent

logger = logging.getLogger(__name__)


class ValueIterationAgent(StatelessPlannerAgent):
    """Drop-in for ``rl_agents.agents.dynamic_programming.value_iteration.ValueIterationAgent``."""

    def __init__(self, env, config=None):
        super(ValueIterationAgent, self).__init__(config)
        self.finite_mdp = self.is_finite_mdp(env)
        self.mdp = device_model.finite_mdp_of(env)          # raises the reference's TypeError otherwise
        self.env = env
        self.models = device_model.ModelCache()
        self.sweeps = 0
        self.state_action_value = self.get_state_action_value()

    @classmethod
    def default_config(cls):
        return dict(gamma=1.0, iterations=100)

    def act(self, state):
        # an environment that is not itself a finite MDP is converted again on every call and the
        # state recovered from the conversion (value_iteration.py:29-35); the device model is re-uploaded
        # and re-solved only if the converted tables changed
        if not self.finite_mdp:
            self.mdp = self.env.unwrapped.to_finite_mdp()
            state = self.mdp.state
            self.state_action_value = self.get_state_action_value()
        return np.argmax(self.state_action_value[state, :])

    def _model(self):
        return self.models.get(device_model.spec_from_mdp(self.mdp))

    def get_state_action_value(self):
        model = self._model()
        cached = getattr(model, "_vi_cache", None)
        key = (self.config["gamma"], self.config["iterations"])
        if cached is not None and cached[0] == key:
            self.sweeps = cached[2]
            return cached[1]
        q, self.sweeps = self.models.ctx.vi_solve(model, self.config["gamma"], self.config["iterations"])
        model._vi_cache = (key, q, self.sweeps)
        return q

    def policy_table(self):
        """Boltzmann distribution over the solved Q table, one row per state: softmax(Q[s, :] / temperature) with
        ``config["temperature"]`` (default 1).  Not in the reference, whose ValueIterationAgent cannot serve as the
        prior agent its own vi_prior.json asks for (no ``action_distribution``); see mcts_with_prior.py."""
        if not self.finite_mdp:     # converted environments: same refresh as act()
            self.mdp = self.env.unwrapped.to_finite_mdp()
            self.state_action_value = self.get_state_action_value()
        q = self.state_action_value
        z = np.exp((q - q.max(axis=1, keepdims=True)) / self.config.get("temperature", 1.0))
        return z / z.sum(axis=1, keepdims=True)

    def action_distribution(self, state):
        """{action: probability} in ``state`` (AbstractStochasticAgent.action_distribution, common/abstract.py:105)."""
        row = self.policy_table()[state]
        return {a: row[a] for a in range(len(row))}

    def get_state_value(self):
        return self.models.ctx.vi_solve_v(self._model(), self.config["gamma"], self.config["iterations"])

    @staticmethod
    def best_action_value(action_values):
        return action_values.max(axis=-1)

    @staticmethod
    def is_finite_mdp(env):
        """True when ``env`` itself is a finite-MDP environment, whose ``mdp`` is then used as is -- this package's
        FiniteMDPEnv or, when installed, the ``finite_mdp`` package's (value_iteration.py:75-82)."""
        base = getattr(env, "unwrapped", env)
        from rl_agents_amd.envs.finite_mdp import FiniteMDPEnv
        candidates = [FiniteMDPEnv]
        try:
            import importlib
            candidates.append(importlib.import_module("finite_mdp.envs.finite_mdp_env").FiniteMDPEnv)
        except (ImportError, AttributeError):
            pass
        return isinstance(base, tuple(candidates))

    def plan_trajectory(self, state, horizon=10):
        """Greedy roll-out of the solved Q table through the model (value_iteration.py:84-96): the visited states
        and greedy actions, ending with ``(terminal_state, None)`` when a terminal state is entered."""
        q = self.get_state_action_value()
        visited, chosen = [], []
        for _ in range(horizon):
            greedy = np.argmax(q[state])
            visited.append(state)
            chosen.append(greedy)
            state = self.mdp.next_state(state, greedy)
            if self.mdp.terminal[state]:
                visited.append(state)
                chosen.append(None)
                break
        return visited, chosen
