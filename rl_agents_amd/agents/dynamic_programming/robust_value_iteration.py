"""Robust (min over models) value iteration on the MI355X planning core (reference
``rl_agents/agents/dynamic_programming/robust_value_iteration.py``)."""
import numpy as np

from rl_agents_amd import device_model
from rl_agents_amd.agents.common.abstract import AbstractAgent
from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent


class RobustValueIterationAgent(ValueIterationAgent):
    """Drop-in for ``...robust_value_iteration.RobustValueIterationAgent``: models come from the config
    (``models: [{mode, transition, reward}, ...]``), the fixed point is recomputed on every ``act``
    (robust_value_iteration.py:29-30) -- served from the per-model cache when nothing changed."""

    def __init__(self, env, config=None):
        AbstractAgent.__init__(self, config)
        self.env = env
        self.mode = None
        self.transitions = np.array([])   # M x S x A (x S)
        self.rewards = np.array([])       # M x S x A
        self.models = device_model.ModelCache()
        self.sweeps = 0
        self.models_from_config()

    @classmethod
    def default_config(cls):
        config = super(RobustValueIterationAgent, cls).default_config()
        config.update(dict(models=[]))
        return config

    def models_from_config(self):
        if not self.config.get("models", None):
            raise ValueError("No finite MDP model provided in agent configuration")
        self.mode = self.config["models"][0]["mode"]     # all models share one mode
        self.transitions = np.array([mdp["transition"] for mdp in self.config["models"]])
        self.rewards = np.array([mdp["reward"] for mdp in self.config["models"]])
        if self.mode not in ("deterministic", "stochastic"):
            raise ValueError("Unknown mode")

    def act(self, state):
        return np.argmax(self.get_state_action_value()[state, :])

    def _model(self):
        return self.models.get(device_model.TableSpec(self.mode, self.transitions, self.rewards))

    def get_state_action_value(self):
        model = self._model()
        cached = getattr(model, "_vi_cache", None)
        key = (self.config["gamma"], self.config["iterations"])
        if cached is not None and cached[0] == key:
            self.sweeps = cached[2]
            return cached[1]
        q, self.sweeps = self.models.ctx.vi_solve(model, self.config["gamma"], self.config["iterations"], robust=True)
        model._vi_cache = (key, q, self.sweeps)
        return q

    def get_state_value(self):
        raise NotImplementedError("the V-form robust iteration is not on the device path; use get_state_action_value")

    @staticmethod
    def worst_case(model_action_values):
        return np.min(model_action_values, axis=0)
