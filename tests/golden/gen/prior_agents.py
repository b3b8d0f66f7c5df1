"""Prior agents plugged into the UNMODIFIED reference MCTSWithPriorPolicyAgent by the golden generator.

The reference's own config for this agent (scripts/configs/HighwayEnv/agents/MCTSWithPriorPolicyAgent/vi_prior.json)
names ValueIterationAgent as the prior agent, which has no ``action_distribution`` and therefore cannot run; its
default prior is a torch DQN.  What mcts_with_prior.py:47-62 needs from a prior agent is ``env`` (assigned),
``act(observation)`` and ``action_distribution(observation) -> {action: probability}``: this module supplies the
smallest such agent -- a Boltzmann distribution over the reference ValueIterationAgent's own Q table -- so that the
reference's planner-side code path is the thing exercised.
"""
import numpy as np

from rl_agents.agents.dynamic_programming.value_iteration import ValueIterationAgent


def boltzmann_table(q, temperature):
    """softmax(Q[s, :] / temperature) row by row; the formula tests/ and the package's VI agent repeat."""
    z = np.exp((q - q.max(axis=1, keepdims=True)) / temperature)
    return z / z.sum(axis=1, keepdims=True)


class BoltzmannQAgent(object):
    def __init__(self, env, config=None):
        self.config = dict(config or {})
        self.env = env
        vi = ValueIterationAgent(env, dict(gamma=self.config.get("gamma", 0.9), iterations=self.config.get("iterations", 100)))
        self.q = np.array(vi.state_action_value)
        self.table = boltzmann_table(self.q, self.config.get("temperature", 1.0))
        if self.config.get("mask"):  # zero out some actions in some states (zero-probability children)
            rng = np.random.Generator(np.random.PCG64(self.config["mask"]))
            drop = rng.random(self.table.shape) < 0.25
            drop[np.arange(len(drop)), self.table.argmax(axis=1)] = False
            self.table = np.where(drop, 0.0, self.table)
            self.table = self.table / self.table.sum(axis=1, keepdims=True)

    def act(self, observation):
        return int(np.argmax(self.q[observation, :]))

    def action_distribution(self, observation):
        return {a: self.table[observation, a] for a in range(self.table.shape[1])}

    def seed(self, seed=None):
        return [seed]

    def reset(self):
        pass


class BoltzmannVIAgent(ValueIterationAgent):
    """The reference's OWN ValueIterationAgent -- which converts its environment again and re-solves at EVERY act()
    (value_iteration.py:29-35) -- plus the ``action_distribution`` that mcts_with_prior.py:47-54 asks a prior agent for and that
    the reference's vi_prior.json assumes it has: a Boltzmann distribution over the Q table the last act() solved.  With it the
    unmodified MCTSWithPriorPolicyAgent re-solves value iteration on the table of whatever environment copy a policy is asked
    about: the per-episode, per-step-changing case (tests/golden/per_episode_prior.npz)."""

    def action_distribution(self, observation):
        table = boltzmann_table(np.array(self.state_action_value), self.config.get("temperature", 1.0))
        return {a: table[observation, a] for a in range(table.shape[1])}
