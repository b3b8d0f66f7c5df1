"""Robust (min over models) value iteration on the MI355X planning core (reference
``rl_agents/agents/dynamic_programming/robust_value_iteration.py``): ``Q <- min_m (R_m + gamma * next_v_m(max_a Q))``,
no terminal masking, solved by ``mp_vi_solve(robust=1)`` (rl_agents_amd/csrc/vi.hip)."""
import numpy as np

from rl_agents_amd import device_model
from rl_agents_amd.agents.common.abstract import StatelessPlannerAgent
from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent

_SUPPORTED_MODES = ("deterministic", "stochastic")    # robust_value_iteration.py:50-58 has no sparse branch


class RobustValueIterationAgent(ValueIterationAgent):
    """Drop-in for ``...robust_value_iteration.RobustValueIterationAgent``.

    The candidate models come from the agent config (``models: [{mode, transition, reward}, ...]``, e.g.
    ``scripts/configs/FiniteMDPEnv/large/agents/robust_value_iteration.json``), not from the environment.  The
    reference recomputes the fixed point on every ``act`` (robust_value_iteration.py:29-30); here the solve is
    served from the per-model cache when neither the models nor gamma / iterations changed."""

    def __init__(self, env, config=None):
        StatelessPlannerAgent.__init__(self, config)       # skip ValueIterationAgent.__init__: no env MDP is read
        self.env = env
        self.models = device_model.ModelCache()
        self.sweeps = 0
        self.mode, self.transitions, self.rewards = None, np.array([]), np.array([])   # M x S x A (x S), M x S x A
        self.models_from_config()

    @classmethod
    def default_config(cls):
        return dict(ValueIterationAgent.default_config(), models=[])

    def models_from_config(self):
        models = self.config.get("models") or []
        if not models:
            raise ValueError("No finite MDP model provided in agent configuration")
        self.mode = models[0]["mode"]                      # the reference assumes one mode for all models
        if self.mode not in _SUPPORTED_MODES:
            raise ValueError("Unknown mode")
        self.transitions = np.array([m["transition"] for m in models])
        self.rewards = np.array([m["reward"] for m in models])

    def _model(self):
        return self.models.get(device_model.TableSpec(self.mode, self.transitions, self.rewards))

    def get_state_action_value(self):
        model = self._model()
        key = (self.config["gamma"], self.config["iterations"])
        cached = getattr(model, "_vi_cache", None)
        if cached is None or cached[0] != key:
            q, sweeps = self.models.ctx.vi_solve(model, key[0], key[1], robust=True)
            cached = model._vi_cache = (key, q, sweeps)
        self.sweeps = cached[2]
        return cached[1]

    def act(self, state):
        return np.argmax(self.get_state_action_value()[state, :])

    def get_state_value(self):
        """V <- max_a min_m (R_m + gamma next_v_m(V)) to its fixed point (robust_value_iteration.py:32-37), on the device."""
        return self.models.ctx.vi_solve_v(self._model(), self.config["gamma"], self.config["iterations"], robust=True)

    @staticmethod
    def worst_case(model_action_values):
        """min over the model axis (robust_value_iteration.py:46-48)."""
        return np.min(model_action_values, axis=0)
