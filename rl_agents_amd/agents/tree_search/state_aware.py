"""State-aware optimistic planning on the MI355X planning core (reference
``rl_agents/agents/tree_search/state_aware.py``); the iterations run in ``mp_saopd_plan`` (rl_agents_amd/csrc/saopd.hip).

What the reference's planner OBJECT keeps between ``plan()`` calls -- the ``state_values`` / ``state_nodes`` dictionaries
and, through them, the nodes of earlier trees (``reset()`` only installs a new root) -- lives on the device in a
``native.StateAwarePlanners`` batch that this planner holds for as long as the model and the number of roots stay the
same.  Consecutive ``plan()`` calls therefore reproduce a reference agent's episode plan by plan
(tests/test_gpu_agents.py); changing the model or the batch size starts new planners.
"""
import logging

import numpy as np

from rl_agents_amd import native
from rl_agents_amd.agents.tree_search.abstract import AbstractTreeSearchAgent, build_tree
from rl_agents_amd.agents.tree_search.deterministic import OptimisticDeterministicPlanner

logger = logging.getLogger(__name__)


class StateAwarePlanner(OptimisticDeterministicPlanner):
    """State-aware planner (state_aware.py:70-127) for one or many independent planners of one finite MDP."""
    carries_state = True    # per-slot state on the device: callers keep the batch composition fixed

    def __init__(self, env, config=None):
        super(StateAwarePlanner, self).__init__(env, config)
        self._device = None         # (model, n_planners, native.StateAwarePlanners)

    @classmethod
    def default_config(cls):
        cfg = super(StateAwarePlanner, cls).default_config()
        cfg.update({"backup_aggregated_nodes": True, "prune_suboptimal_leaves": True, "accuracy": 0})
        return cfg

    def device_planners(self, model, n):
        held = self._device
        if held is None or held[0] is not model or held[1] != n:
            if held is not None:
                logger.warning("state-aware planner: model or batch size changed (%d -> %d planners); the state values and "
                               "state-node lists kept from earlier plans are dropped", held[1], n)
                held[2].close()
            held = self._device = (model, n, native.StateAwarePlanners(self.models.ctx, model, n))
        return held[2]

    def forget(self):
        """Drop the planners' kept state (a new planner object in the reference's terms)."""
        if self._device is not None:
            self._device[2].close()
            self._device = None

    def plan_batch(self, state, root_states, root_steps=None, rng_states=None):
        model = self.model_for(state)
        n = len(root_states)
        if rng_states is None:
            rng_states = self.batch_rng_states(n)
        cfg = self.config
        if cfg["gamma"] == 1:
            raise ZeroDivisionError("division by zero")              # 1 / (1 - gamma), state_aware.py:83,120
        planners = self.device_planners(model, n)
        out = planners.plan(root_states, int(cfg["budget"]), cfg["gamma"], cfg.get("terminal_reward", 0), rng_states,
                            accuracy=cfg["accuracy"], backup_aggregated_nodes=cfg["backup_aggregated_nodes"],
                            prune_suboptimal_leaves=cfg["prune_suboptimal_leaves"],
                            max_plan_len=int(cfg["budget"]) // model.A + 1)
        if not getattr(self, "defer_errors", False):               # (a batched caller checks its live slots only)
            self.raise_for_status(out["status"])
        out["rng_states"] = rng_states
        self.relabel(out, model)
        self.last, self._root, self._last_actions, self._last_model = out, None, model.A, model
        self.env_steps += int(out["env_steps"].sum())
        return out

    # -- device-resident evaluation loop (BatchedEvaluation): the planners' arenas, state values and lists already live on
    # the device between plans; only the call form changes (mp_saopd_plan with MP_MEM_DEVICE: asynchronous, a planner whose
    # backup queue fills up reports MP_ERR_ALLOC and stays failed instead of being rolled back)
    def supports_device_loop(self):
        return True

    def plan_batch_device(self, state, model, n, d_state, d_steps, d_rng, d_plans, d_len, d_env_steps, d_status, d_value=None,
                          keep_actions=None):
        import torch
        cfg = self.config
        if cfg["gamma"] == 1:
            raise ZeroDivisionError("division by zero")
        planners = self.device_planners(model, n)
        if getattr(self, "_d_updates", None) is None or self._d_updates.shape[0] != n:
            self._d_updates = torch.zeros(n, dtype=torch.int64, device=d_state.device)
        planners.plan_device(d_state, int(cfg["budget"]), cfg["gamma"], cfg.get("terminal_reward", 0), d_rng,
                             int(d_plans.shape[1]), accuracy=cfg["accuracy"], backup_aggregated_nodes=cfg["backup_aggregated_nodes"],
                             prune_suboptimal_leaves=cfg["prune_suboptimal_leaves"], plans=d_plans, plan_len=d_len,
                             env_steps=d_env_steps, updates=self._d_updates, status=d_status)
        self.last, self._root = None, None

    def raise_for_device_status(self, d_status, live):
        bad = d_status[live]
        if bad.numel():
            self.raise_for_status(bad.cpu().numpy())

    @staticmethod
    def raise_for_status(status):
        """The reference's exceptions for the per-planner status codes."""
        status = np.asarray(status)
        if (status == native.MP_ERR_REWARD_RANGE).any():
            raise ValueError("This planner assumes that all rewards are normalized in [0, 1]")  # deterministic.py:46-47
        if (status == native.MP_ERR_ARG).any():
            raise ValueError("max() arg is an empty sequence")     # every leaf pruned (state_aware.py:95)
        if (status != 0).any():
            raise RuntimeError("state-aware planner: backup queue overflow on the device (MP_SAOPD_QUEUE_LIMIT_MB)")

    def export_tree(self, root=0):
        arrays, state_values = self._device[2].export(root)
        arrays = self.relabel_tree(arrays, getattr(self, "_last_model", None))
        arrays["value_lower"] = arrays["lower"]
        # get_value_upper_bound (state_aware.py:65-67): value_lower + gamma**depth * state_values[observation]
        gamma = self.config["gamma"]
        arrays["value_upper"] = np.array([lo + (gamma ** int(d)) * state_values[s] for lo, d, s in
                                          zip(arrays["lower"], arrays["depth"], arrays["state"])])
        arrays["observation"] = arrays["state"]
        tree = build_tree(arrays, "value_upper", extra=("value_lower", "value_upper", "reward", "done", "state",
                                                        "observation", "alive"), planner=self)
        tree.state_values = state_values
        return tree

    @property
    def state_values(self):
        """state_values of planner 0 as an array over states (the reference's dict, str(observation) -> bound)."""
        return None if self._device is None else self._device[2].export(0)[1]


class StateAwarePlannerAgent(AbstractTreeSearchAgent):
    """Drop-in for ``rl_agents.agents.tree_search.state_aware.StateAwarePlannerAgent``."""
    PLANNER_TYPE = StateAwarePlanner
