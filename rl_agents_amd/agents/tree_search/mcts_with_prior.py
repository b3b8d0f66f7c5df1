"""MCTS guided by another agent's action distribution (reference ``rl_agents/agents/tree_search/mcts_with_prior.py``).

The reference replaces the planner's prior and rollout policies by ``agent_policy_available`` (:47-62): for the state
a policy is asked about, ``prior_agent.act(observation)`` then ``prior_agent.action_distribution(observation)``.  On a
finite MDP the observation is the state index, so both policies are ``[S, A]`` tables; they are tabulated once per
model on the host, uploaded with ``mp_policy_load`` and looked up per state by ``mp_uct_plan_policy``
(rl_agents_amd/csrc/uct.hip).  Results equal the reference's at equal seeds (tests/test_gpu_agents.py).

Prior agents: anything with ``action_distribution(state) -> {action: probability}`` over actions ``0..A-1`` that
depends on the observed state only; agents exposing ``policy_table() -> [S, A]`` (this package's
ValueIterationAgent: Boltzmann over its Q table) skip the per-state calls.  The reference's default prior agent is its
torch DQN, which is outside this package: the default here is the ValueIterationAgent that the reference's own
``vi_prior.json`` names.
"""
import hashlib

import numpy as np

from rl_agents_amd.agents.common.factory import agent_factory
from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
from rl_agents_amd.configuration import Configurable


def tabulate_prior_agent(prior_agent, n_states, n_actions):
    """``[S, A]`` table of ``prior_agent.action_distribution(s)``, queried the way mcts_with_prior.py:47-54 does."""
    table = getattr(prior_agent, "policy_table", None)
    if callable(table):
        out = np.array(table(), dtype=np.float64)
        if out.shape != (n_states, n_actions):
            raise ValueError("prior agent policy_table() has shape {}, expected {}".format(out.shape, (n_states, n_actions)))
        return out
    out = np.zeros((n_states, n_actions), dtype=np.float64)
    for s in range(n_states):
        prior_agent.act(s)                              # "trigger the computation of action distribution" (:51)
        distribution = prior_agent.action_distribution(s)
        actions, probabilities = list(distribution.keys()), list(distribution.values())
        if sorted(actions) != list(range(n_actions)):
            raise ValueError("the prior agent must give a probability for each of the {} actions".format(n_actions))
        out[s, actions] = probabilities
    return out


class MCTSWithPriorPolicyAgent(MCTSAgent):
    """Drop-in for ``rl_agents.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent``."""

    def __init__(self, env, config=None):
        Configurable.__init__(self, config)
        self.prior_agent = agent_factory(env, self.config["prior_agent"])
        if "model_save" in self.config["prior_agent"]:
            self.prior_agent.load(self.config["prior_agent"]["model_save"])
        super(MCTSWithPriorPolicyAgent, self).__init__(env, self.config)
        self._tables = {}
        self.planner.policy_source = self.policy_tables

    @classmethod
    def default_config(cls):
        config = super(MCTSWithPriorPolicyAgent, cls).default_config()
        config.update({"prior_agent": {
            "__class__": "<class 'rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"}})
        return config

    def policy_tables(self, state, model):
        """(prior, rollout) tables for the model behind ``state``: one distribution serves both (:31-32)."""
        self.prior_agent.env = state                    # "reset prior agent environment" (:49)
        table = tabulate_prior_agent(self.prior_agent, model.S, model.A)
        order = getattr(model, "action_order", None)
        if order is not None:           # the model's columns are in the environment's listing order (device_model)
            table = np.ascontiguousarray(table[:, order])
        available = getattr(model, "available", None)
        if available is not None:
            # agent_policy_available (:56-62): the distribution over the available actions, renormalised by numpy's sum
            restricted = np.zeros_like(table)
            for s in range(model.S):
                av = np.flatnonzero(available[s])
                p = np.array([table[s, a] for a in av])
                p /= np.sum(p)
                restricted[s, av] = p
            table = restricted
        key = hashlib.sha1(table.tobytes()).hexdigest()
        return self._tables.setdefault(key, table), self._tables[key]

    def record(self, state, action, reward, next_state, done, info):
        raise NotImplementedError()

    def save(self, filename):
        return self.prior_agent.save(filename)

    def load(self, filename):
        return self.prior_agent.load(filename)
