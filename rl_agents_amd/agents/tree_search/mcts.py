"""MCTS / UCT agent on the MI355X planning core (reference ``rl_agents/agents/tree_search/mcts.py``).

Same class names, config keys and results as the reference; the episodes run in ``mp_uct_plan``
(rl_agents_amd/csrc/uct.hip).  Deviations, all documented in DESIGN.md:
* ``horizon`` given without ``episodes``: the reference raises KeyError (mcts.py:116-118,180); here
  ``episodes = budget // horizon``;
* ``closed_loop=True`` and environments exposing ``get_available_actions`` raise (open loop over
  ``range(action_space.n)`` only), instead of silently running something else.
"""
import logging

import numpy as np

from rl_agents_amd import device_model
from rl_agents_amd.agents.tree_search.abstract import AbstractPlanner, AbstractTreeSearchAgent, build_tree
from rl_agents_amd.agents.tree_search.olop import OLOP

logger = logging.getLogger(__name__)


def policy_probabilities(policy_config, n_actions):
    """Action distribution of a prior / rollout policy config over actions 0..n-1 (mcts.py:33-97).

    ``random`` and ``random_available`` are uniform; ``preference`` makes ``action`` ``ratio`` times
    likelier than each other action (uniform if that action does not exist)."""
    kind = policy_config["type"]
    if kind in ("random", "random_available"):
        return np.ones(n_actions) / n_actions
    if kind == "preference":
        action, ratio = policy_config["action"], policy_config.get("ratio", 2)
        if 0 <= action < n_actions:
            p = np.ones(n_actions) / (n_actions - 1 + ratio)
            p[action] *= ratio
            return p
        return np.ones(n_actions) / n_actions
    raise ValueError("Unknown policy type")


class MCTS(AbstractPlanner):
    """UCT planner (mcts.py:100-200) for one or many roots of one finite MDP (or closed-form CartPole)."""
    supports_cartpole = True

    def __init__(self, env, prior_policy, rollout_policy, config=None):
        super(MCTS, self).__init__(config)
        self.env = env
        self.prior_policy = prior_policy        # policy config dicts (resolved per model: needs |A|)
        self.rollout_policy = rollout_policy
        # per-state policies (mcts_with_prior.py): callable(state, model) -> (prior [S,A], rollout [S,A]), or None
        self.policy_source = None
        self._policies = {}
        if not self.config["horizon"]:
            self.config["episodes"], self.config["horizon"] = OLOP.allocation(self.config["budget"],
                                                                              self.config["gamma"])
        elif not self.config.get("episodes"):
            self.config["episodes"] = max(self.config["budget"] // self.config["horizon"], 1)
        if self.config["closed_loop"]:
            raise NotImplementedError("closed_loop MCTS is not available on the device planner")

    @classmethod
    def default_config(cls):
        cfg = super(MCTS, cls).default_config()
        # the reference derives the default temperature from the DEFAULT gamma (0.8), whatever gamma
        # the user configures (mcts.py:121-124); kept
        cfg.update({"temperature": 2 / (1 - cfg["gamma"]), "closed_loop": False})
        return cfg

    def reset(self):
        super(MCTS, self).reset()
        self._keep_action = None

    def step_by_subtree(self, action):
        """Tree reuse: the device re-roots the kept trees at the start of the next plan, provided the trees on the
        context are still this planner's (nothing else planned in between); otherwise the tree is reset."""
        ctx = self.models.ctx
        if self.last is None or getattr(ctx, "_uct_tree_owner", None) is not self:
            self.step_by_reset()
            return
        self._keep_action = int(action)
        self.last, self._root = None, None

    def plan_batch(self, state, root_states, root_steps=None, rng_states=None, keep_actions=None):
        model = self.model_for(state)
        n = len(root_states)
        if rng_states is None:
            rng_states = self.batch_rng_states(n)
        cfg = self.config
        ctx = self.models.ctx
        if keep_actions is None and getattr(self, "_keep_action", None) is not None and n == 1:
            keep_actions = [self._keep_action]
        self._keep_action = None
        if keep_actions is not None and getattr(ctx, "_uct_tree_owner", None) is self:
            ctx.uct_step_tree(keep_actions)
        else:
            ctx.uct_reset_tree()
        if self.policy_source is not None:
            prior, rollout = self.policy_source(state, model)
            out = ctx.uct_plan(model, root_states, cfg["episodes"], cfg["horizon"], cfg["gamma"], cfg["temperature"],
                               None, None, rng_states, root_steps=root_steps, max_plan_len=max(cfg["horizon"], 1),
                               policy=self.device_policy(model, prior, rollout))
            self._last_tables = (np.asarray(device_model.finite_mdp_of(state).transition), prior,
                                 np.asarray(root_states, dtype=np.int64))
        else:
            out = ctx.uct_plan(model, root_states, cfg["episodes"], cfg["horizon"], cfg["gamma"], cfg["temperature"],
                               policy_probabilities(self.prior_policy, model.A),
                               policy_probabilities(self.rollout_policy, model.A), rng_states,
                               root_steps=root_steps, max_plan_len=max(cfg["horizon"], 1))
            self._last_tables = None
        out["rng_states"] = rng_states
        self.last, self._root, self._last_actions = out, None, model.A
        ctx._uct_tree_owner = self
        self.env_steps += int(out["env_steps"].sum())
        return out

    def device_policy(self, model, prior, rollout):
        """Upload (once per model and table contents) the per-state policy tables."""
        key = (id(model), id(prior), id(rollout))
        hit = self._policies.get(key)
        if hit is None or hit[0] is not model or hit[1] is not prior or hit[2] is not rollout:
            if len(self._policies) >= 4:
                self._policies.clear()
            hit = (model, prior, rollout, self.models.ctx.load_policy(model, prior, rollout))
            self._policies[key] = hit
        return hit[3]

    def export_tree(self, root=0):
        arrays = self.models.ctx.uct_tree(root)
        if self._last_tables is None:
            return build_tree(arrays, "value", prior=policy_probabilities(self.prior_policy, self._last_actions))
        # per-state priors: a child's prior is the prior agent's probability of its action in the state of its
        # parent (mcts.py:237-246); states follow from the root state and the deterministic transitions
        transition, prior, roots = self._last_tables
        tree = build_tree(arrays, "value")
        tree.prior, tree.state = 1.0, int(roots[root])
        stack = [tree]
        while stack:
            node = stack.pop()
            for action, child in node.children.items():
                child.state = int(transition[node.state, action])
                child.prior = float(prior[node.state, action])
                stack.append(child)
        return tree


class MCTSAgent(AbstractTreeSearchAgent):
    """Drop-in for ``rl_agents.agents.tree_search.mcts.MCTSAgent``."""

    def make_planner(self):
        for key in ("prior_policy", "rollout_policy"):
            policy_probabilities(self.config[key], 2)           # validates the policy type early
        return MCTS(self.env, self.config["prior_policy"], self.config["rollout_policy"], self.config)

    @classmethod
    def default_config(cls):
        config = super(MCTSAgent, cls).default_config()
        config.update({"budget": 100, "horizon": None, "prior_policy": {"type": "random_available"},
                       "rollout_policy": {"type": "random_available"}, "env_preprocessors": []})
        return config
