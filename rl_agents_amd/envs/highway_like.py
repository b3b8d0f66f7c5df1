"""An environment with highway-env's SURFACE for the planners (north_star: "highway-v0-as-finite-mdp"), over the seeded
highway-shaped tables of :mod:`generators`.  Real ``highway_env`` is absent from this image; what matters at the
boundary is where the pieces live, and that is restated here [from memory of highway-env's ``AbstractEnv``]:

* the action restriction lives ON THE ENV: ``get_available_actions()`` looks at the ego vehicle -- IDLE always, LANE_LEFT /
  LANE_RIGHT only where a side lane exists, FASTER / SLOWER only inside the speed range -- and the reference's planners
  ask it node by node (tree_search/mcts.py:59-73, deterministic.py:32-35);
* ``to_finite_mdp()`` (the call ValueIterationAgent makes, dynamic_programming/value_iteration.py:12-21) returns a
  deterministic MDP over the (speed, lane, time) grid WITHOUT any availability table, carrying ``original_shape``
  (read by dynamic_programming/graphics.py:44).

So a device planner that wants the restriction as an ``[S, A]`` table has to derive it:
:func:`rl_agents_amd.device_model.available_actions_of` does, from ``original_shape`` (or an ``available_table`` hook),
and cross-checks the row of the current state against what the env itself answers.
"""
import copy

import numpy as np

from . import generators
from .finite_mdp import DeterministicMDP, Discrete

LANE_LEFT, IDLE, LANE_RIGHT, FASTER, SLOWER = range(5)


class HighwayLikeEnv(object):
    """(speed, lane, time) ego state on a highway-shaped table; 5 actions; restriction on the env, not in the MDP."""

    metadata = {}

    def __init__(self, n_speeds=3, n_lanes=4, n_times=10, seed=3, state=0, table=None):
        self.table = table if table is not None else generators.highway_shaped(n_speeds, n_lanes, n_times, seed=seed)
        self.shape = tuple(int(x) for x in self.table["original_shape"])
        self.action_space = Discrete(5)
        self.observation_space = Discrete(int(np.prod(self.shape)))
        self.steps = 0
        self.config = {"max_steps": 0}
        self.start = int(state)
        self.speed_index, self.lane_index, self.time_index = (int(x) for x in np.unravel_index(self.start, self.shape))

    @property
    def unwrapped(self):
        return self

    @property
    def state_index(self):
        return int(np.ravel_multi_index((self.speed_index, self.lane_index, self.time_index), self.shape))

    def reset(self, *, seed=None, options=None):
        self.speed_index, self.lane_index, self.time_index = (int(x) for x in np.unravel_index(self.start, self.shape))
        self.steps = 0
        return self.state_index, {}

    def get_available_actions(self):
        """highway-env style: a list of action indexes, IDLE first."""
        actions = [IDLE]
        if self.lane_index > 0:
            actions.append(LANE_LEFT)
        if self.lane_index < self.shape[1] - 1:
            actions.append(LANE_RIGHT)
        if self.speed_index < self.shape[0] - 1:
            actions.append(FASTER)
        if self.speed_index > 0:
            actions.append(SLOWER)
        return actions

    def step(self, action):
        s = self.state_index
        reward = float(self.table["reward"][s, action])
        terminated = bool(self.table["terminal"][s])           # flag of the state acted FROM (finite_mdp.py semantics)
        nxt = int(self.table["transition"][s, action])
        self.speed_index, self.lane_index, self.time_index = (int(x) for x in np.unravel_index(nxt, self.shape))
        self.steps += 1
        return nxt, reward, terminated, False, {}

    def to_finite_mdp(self):
        mdp = DeterministicMDP(self.table["transition"], self.table["reward"], terminal=self.table["terminal"],
                               state=self.state_index)
        mdp.original_shape = self.shape        # no `available`: highway-env's conversion has none
        # every call returns a NEW object over the SAME table: give it the table's version (MDP.tables_version protocol), so
        # that an agent re-converting at every step (value_iteration.py:29-35) neither re-hashes nor re-uploads it
        token = self.table.get("_version_token")
        if token is None:
            from .finite_mdp import _TABLE_TOKENS
            token = self.table["_version_token"] = next(_TABLE_TOKENS)
        mdp.__dict__["_tables_token"], mdp.__dict__["_tables_counter"] = token, 0
        mdp.__dict__["_dirty_log"], mdp.__dict__["_dirty_base"] = {}, 0
        mdp.__dict__["_token_shared"] = True   # (copy-on-write identity: MDP._own_identity)
        mdp.__dict__.pop("_private", None)
        return mdp

    def render(self, *a, **k):
        return None

    def close(self):
        pass

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, v if k == "table" else copy.deepcopy(v, memo))
        return new
