"""Optimistic planning for deterministic systems on the MI355X planning core
(reference ``rl_agents/agents/tree_search/deterministic.py``); the expansions run in ``mp_opd_plan``
(rl_agents_amd/csrc/opd.hip)."""
import logging

from rl_agents_amd import native
from rl_agents_amd.agents.tree_search.abstract import AbstractPlanner, AbstractTreeSearchAgent, build_tree

logger = logging.getLogger(__name__)


class OptimisticDeterministicPlanner(AbstractPlanner):
    """OPD planner (deterministic.py:91-122) for one or many roots of one finite MDP."""

    def __init__(self, env, config=None):
        super(OptimisticDeterministicPlanner, self).__init__(config)
        self.env = env

    def plan_batch(self, state, root_states, root_steps=None, rng_states=None):
        self.about_to_plan()
        model = self.model_for(state)
        n = len(root_states)
        if rng_states is None:
            rng_states = self.batch_rng_states(n)
        cfg = self.config
        budget = int(cfg["budget"])
        if cfg["gamma"] == 1 and budget >= model.A:
            raise ZeroDivisionError("float division by zero")       # gamma ** depth / (1 - gamma), deterministic.py:53
        out = self.models.ctx.opd_plan(model, root_states, budget, cfg["gamma"], cfg.get("terminal_reward", 0),
                                       rng_states, max_plan_len=budget // model.A + 1)
        if (out["status"] == native.ERR_REWARD_RANGE).any():
            raise ValueError("This planner assumes that all rewards are normalized in [0, 1]")  # deterministic.py:46-47
        out["rng_states"] = rng_states
        self.relabel(out, model)
        self.last, self._root, self._last_actions, self._last_model = out, None, model.A, model
        self.claim_device_tree()
        self.env_steps += int(out["env_steps"].sum())
        return out

    # -- device-resident evaluation loop (BatchedEvaluation) ----------------------------------------------------------------
    def supports_device_loop(self):
        return type(self).plan_batch is OptimisticDeterministicPlanner.plan_batch     # (not the state-aware / robust planners)

    def device_plan_len(self, model):
        return 1

    def plan_batch_device(self, state, model, n, d_state, d_steps, d_rng, d_plans, d_len, d_env_steps, d_status, d_value=None):
        """One asynchronous batched plan (mp_opd_plan, MP_MEM_DEVICE): only enqueues; reward-range errors land in d_status.
        ``d_value``: optional float64 [n] buffer for the roots' lower bounds."""
        cfg = self.config
        budget = int(cfg["budget"])
        if cfg["gamma"] == 1 and budget >= model.A:
            raise ZeroDivisionError("float division by zero")
        self.about_to_plan()
        self.models.ctx.opd_plan_device(model, n, d_state, budget, cfg["gamma"], cfg.get("terminal_reward", 0), d_rng,
                                        int(d_plans.shape[1]), plans=d_plans, plan_len=d_len, root_lower=d_value,
                                        env_steps=d_env_steps, status=d_status)
        self.claim_device_tree()
        self.last, self._root = None, None

    def raise_for_device_status(self, d_status, live):
        if bool(((d_status == native.ERR_REWARD_RANGE) & live).any().item()):
            raise ValueError("This planner assumes that all rewards are normalized in [0, 1]")  # deterministic.py:46-47

    def export_tree(self, root=0):
        self.require_device_tree()
        a = self._last_actions
        cap = 1 + (int(self.config["budget"]) // a) * a
        arrays = self.relabel_tree(self.models.ctx.opd_tree(root, cap), getattr(self, "_last_model", None))
        arrays["value_lower"], arrays["value_upper"] = arrays["lower"], arrays["upper"]
        tree = build_tree(arrays, "lower", extra=("value_lower", "value_upper", "reward", "done", "state"), planner=self)
        return with_observations(tree)

    def get_visits(self):
        """Observations stepped through (abstract.py:163-167): every node but the root was created by ONE env step, into
        its ``observation``.  Covers the last plan (the reference's log grows over the planner's lifetime)."""
        from collections import defaultdict
        visits = defaultdict(int)
        if self.root is not None:
            for node, _ in self.root.breadth_first_search(self.root):
                if node.parent is not None:
                    visits[str(node.observation)] += 1
        return visits


def with_observations(tree):
    """DeterministicNode.observation (deterministic.py:13,41-43): None at the root, the observation of the step that
    created the node elsewhere -- for a finite-MDP environment the state reached."""
    tree.observation = None
    stack = list(tree.children.values())
    while stack:
        node = stack.pop()
        node.observation = node.state
        stack.extend(node.children.values())
    return tree


class DeterministicPlannerAgent(AbstractTreeSearchAgent):
    """Drop-in for ``rl_agents.agents.tree_search.deterministic.DeterministicPlannerAgent``."""
    PLANNER_TYPE = OptimisticDeterministicPlanner
