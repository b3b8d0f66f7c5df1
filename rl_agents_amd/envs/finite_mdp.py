"""Finite-MDP environment: the host-side data format on the input edge of the planning path.

The reference reads a finite MDP through the third-party ``finite_mdp`` package, which is NOT
under /root/reference and not installed here.  What the reference's planners
touch is small and fully determined by their call sites, so it is restated here:

* ``env.mdp`` / ``env.unwrapped.to_finite_mdp()`` with ``mode``, ``transition``, ``reward``,
  ``terminal``, ``state``, ``next_state(s, a)`` (+ ``next`` in sparse mode)
  -- rl_agents/agents/dynamic_programming/value_iteration.py:12-21,31-34,52-62,91-92
* ``env.action_space.n``, ``env.step(a) -> (obs, reward, terminated, truncated, info)``
  -- rl_agents/agents/tree_search/abstract.py:158-161, mcts.py:145,173, deterministic.py:41
* the config-dict wire format ``{"mode", "transition", "reward", "terminal", "max_steps"}``
  -- scripts/configs/FiniteMDPEnv/**/env_*.json

Step semantics (documented choice, ``finite_mdp`` being absent): acting in state ``s`` with
action ``a`` yields ``reward[s, a]``, moves to ``transition[s, a]`` and reports
``terminated = terminal[s]`` -- the flag of the state the action was taken FROM.  This is the
only reading consistent with the reference's value iteration, which zeroes the continuation
of *source* rows (value_iteration.py:62) and with its ``trap``/``doors`` configs, whose
terminal states carry the +1/-1 reward that is collected by acting in them.
``done_rule="next"`` selects the other convention (``terminated = terminal[s']``).
``truncated`` is raised once ``max_steps`` steps were taken since ``reset`` (0/None = never).
"""
import copy

import numpy as np


class Discrete(object):
    """Minimal stand-in for ``gymnasium.spaces.Discrete`` (only ``n`` is read by planners)."""

    def __init__(self, n):
        self.n = int(n)

    def __repr__(self):
        return "Discrete({})".format(self.n)


_TABLE_TOKENS = iter(range(1, 1 << 62))       # process-wide identities of table sets (MDP.tables_version)
_TABLE_FIELDS = ("transition", "reward", "terminal", "next", "available")


class MDP(object):
    """``tables_version`` protocol (read by rl_agents_amd.device_model: SURVEY.md 8 f-2 -- an agent that re-converts its
    environment on every ``act``, value_iteration.py:29-35, should neither re-hash nor re-upload tables that did not change):
    ``mdp.tables_version`` is a hashable ``(token, counter)``; EQUAL VALUES PROMISE IDENTICAL TABLES.  The counter advances
    whenever a table attribute is assigned (``mdp.reward = new_array``) or rows are edited through :meth:`edit_rows`, which also
    remembers WHICH rows changed so that the device model is patched (mp_model_update_rows) instead of re-uploaded.  The arrays
    an MDP holds are read-only views: an element assignment through the MDP (``mdp.reward[s, a] = x``) raises instead of silently
    leaving a stale device model -- use ``edit_rows`` or assign a new array.  An environment that edits, behind the MDP's back,
    an array it kept its own reference to must call :meth:`touch`.  Objects without ``tables_version`` (any third-party MDP) are
    keyed by a content hash of their tables, as before."""
    mode = None

    def __setattr__(self, name, value):
        if name in _TABLE_FIELDS:
            d = self.__dict__
            self._own_identity()
            d["_tables_counter"] = d.get("_tables_counter", 0) + 1
            d.setdefault("_tables_token", next(_TABLE_TOKENS))
            d["_dirty_log"] = {}                       # a table was replaced: nothing is known about single rows
            d["_dirty_base"] = d["_tables_counter"]
            d.setdefault("_private", {}).pop(name, None)      # (the MDP does not own the caller's array: see edit_rows)
            if isinstance(value, np.ndarray):
                value = value.view()
                value.setflags(write=False)
        object.__setattr__(self, name, value)

    def _own_identity(self):
        """Copy-on-write identity: an MDP that SHARES its version token with other objects over the same tables (every
        ``to_finite_mdp()`` of one highway-like table) leaves that family the first time ITS tables change -- otherwise two
        objects edited differently would both reach (token, 1) and break 'equal versions = identical tables'."""
        d = self.__dict__
        if d.pop("_token_shared", False):
            d["_tables_token"], d["_tables_counter"] = next(_TABLE_TOKENS), 0
            d["_dirty_log"], d["_dirty_base"] = {}, 0

    @property
    def tables_version(self):
        return (self.__dict__.get("_tables_token"), self.__dict__.get("_tables_counter", 0))

    def touch(self):
        """The tables were changed behind the MDP's back (through another reference to an array): a new version, every row
        suspect."""
        d = self.__dict__
        self._own_identity()
        d["_tables_counter"] = d.get("_tables_counter", 0) + 1
        d["_dirty_log"], d["_dirty_base"] = {}, d["_tables_counter"]

    def edit_rows(self, rows, transition=None, reward=None, terminal=None, next_states=None):
        """Replace the rows ``rows`` (state indices) of the given tables and remember them as the delta of this version (the
        device model is then patched row by row: mp_model_update_rows).  The edit never writes into an array the MDP was
        handed (it may be a slice of a larger array, or shared with other MDP objects): the first edit of a table takes a
        PRIVATE copy, later edits go into that copy in place."""
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        d = self.__dict__
        own = d.setdefault("_private", {})
        shared = d.get("_token_shared", False)
        for name, new in (("transition", transition), ("reward", reward), ("terminal", terminal), ("next", next_states)):
            if new is None:
                continue
            priv = own.get(name)
            if priv is None:
                priv = own[name] = np.array(d[name])   # a copy this object owns (never `arr.base`: numpy collapses a view's
            priv[rows] = new                            # base to the OWNER of the memory, e.g. a whole stack of tables)
            view = priv.view()
            view.setflags(write=False)
            d[name] = view
        if shared:
            # the family's device model is not this object's any more: a fresh identity whose first version is whole
            self._own_identity()
            d["_tables_counter"] = 1
            d["_dirty_log"], d["_dirty_base"] = {}, 1
            return
        d["_tables_counter"] = d.get("_tables_counter", 0) + 1
        d.setdefault("_dirty_log", {})[d["_tables_counter"]] = rows.copy()

    def dirty_rows_since(self, counter):
        """Rows edited since version ``counter`` of THIS object's tables (sorted, unique), or None when that is not known
        (a whole table was assigned since, or the version is older than the log)."""
        d = self.__dict__
        if counter < d.get("_dirty_base", 0) or counter > d.get("_tables_counter", 0):
            return None
        log = d.get("_dirty_log", {})
        parts = []
        for c in range(counter + 1, d.get("_tables_counter", 0) + 1):
            if c not in log:
                return None
            parts.append(log[c])
        if len(log) > 64:                               # keep the log short: forget what nobody can ask for cheaply any more
            for c in sorted(log)[:-32]:
                del log[c]
            d["_dirty_base"] = min(log) - 1
        return np.unique(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64)

    def __deepcopy__(self, memo):
        # a copy's tables may be edited independently of the original's: it gets its own identity (and writable arrays of
        # its own behind the read-only views)
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in _TABLE_FIELDS and isinstance(v, np.ndarray):
                v = np.array(v)
                v.setflags(write=False)
                new.__dict__[k] = v
            elif k not in ("_tables_token", "_tables_counter", "_dirty_log", "_dirty_base", "_private", "_token_shared"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        new.__dict__["_tables_token"], new.__dict__["_tables_counter"] = next(_TABLE_TOKENS), 0
        new.__dict__["_dirty_log"], new.__dict__["_dirty_base"] = {}, 0
        return new

    def __init__(self, transition, reward, terminal=None, state=0, done_rule="source"):
        self.transition = transition
        self.reward = np.ascontiguousarray(reward, dtype=np.float64)
        n_states = self.reward.shape[0]
        if terminal is None:
            terminal = np.zeros(n_states, dtype=bool)
        # configs sometimes give terminal as [[0],[1],...] (anti_vi/env_1.json)
        self.terminal = np.asarray(terminal).reshape(n_states).astype(bool)
        self.state = int(state)
        if done_rule not in ("source", "next"):
            raise ValueError("done_rule must be 'source' or 'next'")
        self.done_rule = done_rule

    @property
    def n_states(self):
        return self.reward.shape[0]

    @property
    def n_actions(self):
        return self.reward.shape[1]

    def next_state(self, state, action, np_random=None):
        raise NotImplementedError

    def step(self, action, np_random=None):
        s = self.state
        reward = float(self.reward[s, action])
        s_next = self.next_state(s, action, np_random)
        done = bool(self.terminal[s] if self.done_rule == "source" else self.terminal[s_next])
        self.state = int(s_next)
        return self.state, reward, done

    def to_config(self):
        cfg = dict(mode=self.mode,
                   transition=np.asarray(self.transition).tolist(),
                   reward=self.reward.tolist(),
                   terminal=self.terminal.astype(int).tolist())
        if self.mode == "sparse":
            cfg["next"] = np.asarray(self.next).tolist()
        return cfg

    @staticmethod
    def from_config(config):
        mode = config["mode"]
        kw = dict(terminal=config.get("terminal"), state=config.get("state", 0),
                  done_rule=config.get("done_rule", "source"))
        if mode == "deterministic":
            return DeterministicMDP(config["transition"], config["reward"], **kw)
        if mode == "stochastic":
            return StochasticMDP(config["transition"], config["reward"], **kw)
        if mode == "sparse":
            return SparseMDP(config["transition"], config["next"], config["reward"], **kw)
        raise ValueError("Unknown mode")


class DeterministicMDP(MDP):
    mode = "deterministic"

    def __init__(self, transition, reward, **kw):
        super().__init__(np.ascontiguousarray(transition, dtype=np.int64), reward, **kw)
        if self.transition.shape != self.reward.shape:
            raise ValueError("transition and reward must both be [S, A]")

    def next_state(self, state, action, np_random=None):
        return int(self.transition[state, action])


class StochasticMDP(MDP):
    mode = "stochastic"

    def __init__(self, transition, reward, **kw):
        super().__init__(np.ascontiguousarray(transition, dtype=np.float64), reward, **kw)
        s, a = self.reward.shape
        if self.transition.shape != (s, a, s):
            raise ValueError("transition must be [S, A, S]")

    def next_state(self, state, action, np_random=None):
        rng = np_random if np_random is not None else np.random
        return int(rng.choice(self.n_states, p=self.transition[state, action]))


class SparseMDP(MDP):
    """``transition[s, a, b]`` = probability of moving to ``next[s, a, b]`` (value_iteration.py:56-59)."""
    mode = "sparse"

    def __init__(self, transition, next_states, reward, **kw):
        super().__init__(np.ascontiguousarray(transition, dtype=np.float64), reward, **kw)
        self.next = np.ascontiguousarray(next_states, dtype=np.int64)
        if self.next.shape != self.transition.shape:
            raise ValueError("next and transition must both be [S, A, B]")

    def next_state(self, state, action, np_random=None):
        rng = np_random if np_random is not None else np.random
        b = int(rng.choice(self.transition.shape[2], p=self.transition[state, action]))
        return int(self.next[state, action, b])


class FiniteMDPEnv(object):
    """Gym-style environment over an :class:`MDP` (5-tuple ``step``, ``unwrapped``, ``to_finite_mdp``)."""

    metadata = {}

    def __init__(self, config=None):
        self.config = {"mode": "deterministic", "transition": [[0]], "reward": [[0]], "max_steps": 0}
        self.mdp = None
        self.steps = 0
        self.np_random = np.random.default_rng()
        self.action_space = None
        self.observation_space = None
        if config is not None:
            self.configure(config)

    # -- gym surface ---------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    def configure(self, config):
        self.config.update(config)
        self.mdp = MDP.from_config(self.config)
        self.action_space = Discrete(self.mdp.n_actions)
        self.observation_space = Discrete(self.mdp.n_states)
        self.steps = 0

    def seed(self, seed=None):
        self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        self.mdp.state = int(self.config.get("state", 0))
        self.steps = 0
        return self.mdp.state, {}

    def step(self, action):
        state, reward, done = self.mdp.step(int(action), np_random=self.np_random)
        self.steps += 1
        max_steps = self.config.get("max_steps", 0)
        truncated = bool(max_steps) and self.steps >= max_steps
        return state, reward, done, truncated, {}

    def to_finite_mdp(self):
        return self.mdp

    def copy_with_config(self, config):
        """A copy of this environment -- same current state and step count -- with ``config`` applied: the
        preprocessor the reference's model lists use to derive candidate models of a finite MDP
        (scripts/configs/FiniteMDPEnv/large/agents/discrete_robust_planner.json: ``{"method": "copy_with_config",
        "args": {mode, transition, reward, terminal}}``).  [restated: ``finite_mdp`` is absent from this image]"""
        new = copy.deepcopy(self)
        state, steps = self.mdp.state, self.steps
        new.configure(dict(config))
        new.mdp.state, new.steps = state, steps
        return new

    def render(self, *a, **k):
        return None

    def close(self):
        pass

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))
        return new


class MaskedFiniteMDPEnv(FiniteMDPEnv):
    """A finite-MDP environment that restricts the actions available in each state, the way highway-env's
    ``get_available_actions`` does (no lane change off the road, no speed change beyond the limits).  The reference's
    planners query it through ``state.get_available_actions()`` (mcts.py:59-73,88-91, deterministic.py:32-35); here the
    restriction is a table ``config["available"][S][A]`` of 0/1 flags (at least one action per state), also exposed as
    ``env.mdp.available`` -- which is what the device planners upload as a per-state action bitmask.  Stepping an
    unavailable action is still defined by the tables (the reference's ``random`` policy does it)."""

    def configure(self, config):
        super().configure(config)
        avail = self.config.get("available")
        if avail is None:
            avail = np.ones(self.mdp.reward.shape, dtype=bool)
        avail = np.asarray(avail).astype(bool).reshape(self.mdp.reward.shape)
        if not avail.any(axis=1).all():
            raise ValueError("every state needs at least one available action")
        self.mdp.available = avail

    def get_available_actions(self):
        """Sorted list of the actions available in the current state."""
        return [int(a) for a in np.flatnonzero(self.mdp.available[self.mdp.state])]


class OrderedMaskedFiniteMDPEnv(MaskedFiniteMDPEnv):
    """A :class:`MaskedFiniteMDPEnv` that LISTS its available actions in a fixed non-ascending order
    (``config["listing_order"]``: a permutation of the action ids; highway-env lists IDLE first) and keeps the restriction
    on the env object, not in the MDP: the planners see it through ``get_available_actions()`` (reference) or through the
    ``available_table(mdp)`` hook (device: :func:`rl_agents_amd.device_model.availability_of`).  Any MDP mode."""

    def configure(self, config):
        super().configure(config)
        n = self.mdp.reward.shape[-1]
        order = [int(a) for a in self.config.get("listing_order", range(n))]
        if sorted(order) != list(range(n)):
            raise ValueError("listing_order must be a permutation of the action ids")
        self._order = order
        self._available = self.mdp.available
        del self.mdp.available                      # the restriction lives on the env object

    def get_available_actions(self):
        return [a for a in self._order if self._available[self.mdp.state, a]]

    def available_table(self, mdp):
        return np.asarray(self._available, dtype=bool), list(self._order)
