"""Model extraction: environment object -> device-resident transition model.

The reference's planners treat the environment as an opaque Python object that is deep-copied
(``safe_deepcopy_env``, common/factory.py:119-134) and stepped (tree_search/abstract.py:158-161).
A GPU kernel cannot call that, so the boundary extracts what the object *is* for the planners:

* a finite MDP -- ``env.unwrapped.mdp`` of a FiniteMDPEnv, or ``env.unwrapped.to_finite_mdp()``
  (highway-env and friends; the same call ValueIterationAgent makes, value_iteration.py:12-21) --
  becomes tables ``transition[S,A]``, ``reward[S,A]``, ``terminal[S]`` on the device;
* anything else raises ``TypeError`` (there is deliberately no CPU fallback).

Uploads are cached per context and keyed by a content hash of the tables, so agents that rebuild the
MDP on every ``act`` (value_iteration.py:29-35) only pay for an upload when the tables changed.
"""
import functools
import hashlib
import os

try:                                    # optional: only the speed of the model-cache key depends on it
    from xxhash import xxh3_128 as _xxh3_128
except ImportError:                     # pragma: no cover
    _xxh3_128 = None

import numpy as np

from . import runtime


class TableSpec(object):
    """Host-side view of a finite MDP in the reference's wire format (mode / transition / reward / terminal)."""

    def __init__(self, mode, transition, reward, terminal=None, next_states=None, done_rule="source", max_steps=0,
                 available=None, action_order=None, version=None, dirty_rows_since=None):
        # version: hashable identity of the table CONTENTS promised by the environment (MDP.tables_version: equal values =
        # identical tables) -- the cache then keys the model without hashing 16 B per (s, a) on every act();
        # dirty_rows_since(counter): the rows changed since an earlier version of the same tables, for the delta upload
        self.version, self.dirty_rows_since = version, dirty_rows_since
        self.mode = mode
        self.reward = np.ascontiguousarray(reward, dtype=np.float64)
        if mode == "deterministic":
            self.transition = np.ascontiguousarray(transition, dtype=np.int64)
        elif mode in ("stochastic", "sparse"):
            self.transition = np.ascontiguousarray(transition, dtype=np.float64)
        else:
            raise ValueError("Unknown mode")                      # value_iteration.py:60-61
        self.next = None if next_states is None else np.ascontiguousarray(next_states, dtype=np.int64)
        n_states = self.reward.shape[-2]
        self.terminal = (np.zeros(n_states, dtype=np.uint8) if terminal is None
                         else np.ascontiguousarray(np.asarray(terminal).reshape(n_states).astype(np.uint8)))
        self.done_rule = done_rule
        self.max_steps = int(max_steps or 0)
        # actions state.get_available_actions() lists per state (bool [S, A]); None = no restriction
        self.available = (None if available is None else
                          np.ascontiguousarray(np.asarray(available).reshape(self.reward.shape[-2:]).astype(np.uint8)))
        # the order in which the environment LISTS its actions (get_available_actions()), when it is not ascending: a
        # permutation `order` with order[j] = the j-th listed action id.  The reference creates a node's children in
        # listing order (deterministic.py:32-43, mcts.py:237-246) and its tie-breaks index that order, so the device
        # plans in the PERMUTED action space (column j of every table = action order[j]) and the planners map labels
        # back at the boundary (AbstractPlanner.relabel).
        self.action_order = None
        if action_order is not None:
            order = np.asarray(action_order, dtype=np.int64).reshape(-1)
            if sorted(order.tolist()) != list(range(self.reward.shape[-1])):
                raise ValueError("action_order must be a permutation of the action ids")
            if not np.array_equal(order, np.arange(len(order))):
                self.action_order = order
                if mode == "deterministic":
                    self.transition = np.ascontiguousarray(self.transition[..., order])                 # [.., S, A]
                else:                                                                                   # [S, A, S] / [S, A, B]
                    self.transition = np.ascontiguousarray(self.transition[:, order])
                    if self.next is not None:
                        self.next = np.ascontiguousarray(self.next[:, order])
                self.reward = np.ascontiguousarray(self.reward[..., order])
                if self.available is not None:
                    self.available = np.ascontiguousarray(self.available[:, order])

    @property
    def n_states(self):
        return self.reward.shape[-2]

    @property
    def n_actions(self):
        return self.reward.shape[-1]

    def static_key(self):
        """What a version does not cover: shapes, the episode rules and what the ENVIRONMENT adds to the MDP's tables (the
        availability table -- a few KB, hashed -- and the listing order)."""
        av = None if self.available is None else hashlib.blake2b(self.available.view(np.uint8).reshape(-1), digest_size=8).hexdigest()
        return (self.mode, self.transition.shape, self.done_rule, self.max_steps, av,
                None if self.action_order is None else tuple(int(a) for a in self.action_order))

    def key(self):
        # a 128-bit content digest: a collision would silently serve a stale model.  The key is recomputed on every
        # act() (the tables may have been edited in place), so its speed is most of a single-root act(): XXH3-128
        # (10+ GB/s) where the xxhash package is present -- BLAKE2b (1.5 GB/s: 0.55 of the 0.9 ms of an MCTSAgent.act()
        # at S = 10 000) otherwise.  Neither is asked to resist an adversary here, only accidents.
        h = _xxh3_128() if _xxh3_128 is not None else hashlib.blake2b(digest_size=16)
        for arr in (self.transition, self.reward, self.terminal, self.next, self.available):
            h.update(b"-" if arr is None else arr.view(np.uint8).reshape(-1))
        return (self.mode, self.transition.shape, self.done_rule, self.max_steps, h.hexdigest(),
                None if self.action_order is None else tuple(int(a) for a in self.action_order))


def finite_mdp_of(env):
    """The finite MDP behind ``env``: ``.mdp`` of a finite-MDP env, else ``to_finite_mdp()``; TypeError otherwise."""
    base = getattr(env, "unwrapped", env)
    mdp = getattr(base, "mdp", None)
    if mdp is not None and hasattr(mdp, "transition") and hasattr(mdp, "reward"):
        return mdp
    if hasattr(base, "to_finite_mdp"):
        return base.to_finite_mdp()
    raise TypeError("Environment must be of type finite_mdp.envs.finite_mdp.FiniteMDPEnv or handle a "
                    "conversion method called 'to_finite_mdp' to such a type.")


def spec_from_mdp(mdp, max_steps=0, available=None, action_order=None):
    version = getattr(mdp, "tables_version", None)
    if version is not None and (not isinstance(version, tuple) or version[0] is None):
        version = None
    return TableSpec(mdp.mode, mdp.transition, mdp.reward, getattr(mdp, "terminal", None),
                     next_states=getattr(mdp, "next", None) if mdp.mode == "sparse" else None,
                     done_rule=getattr(mdp, "done_rule", "source"), max_steps=max_steps, available=available,
                     action_order=action_order, version=version,
                     dirty_rows_since=getattr(mdp, "dirty_rows_since", None) if version is not None else None)


def grid_available(original_shape):
    """Action availability of a (speed V, lane L, time T) time-to-collision grid in highway-env's style [from memory,
    package absent]: LANE_LEFT (0) needs a lane to the left, LANE_RIGHT (2) one to the right, FASTER (3) a higher and
    SLOWER (4) a lower target speed; IDLE (1) is always available.  -> bool [V * L * T, 5]."""
    return _grid_available(tuple(int(x) for x in original_shape))


@functools.lru_cache(maxsize=64)
def _grid_available(shape):
    # (a function of the shape alone; a per-episode evaluation asks it for every environment at every step: cached, and
    # handed out read-only so that no caller can edit the cached table)
    n_speeds, n_lanes, n_times = shape
    v, l, _ = np.meshgrid(np.arange(n_speeds), np.arange(n_lanes), np.arange(n_times), indexing="ij")
    v, l = v.ravel(), l.ravel()
    table = np.stack([l > 0, np.ones_like(l, dtype=bool), l < n_lanes - 1, v < n_speeds - 1, v > 0], axis=1)
    table.setflags(write=False)
    return table


GRID_LISTING_ORDER = (1, 0, 2, 3, 4)   # highway-env lists IDLE first, then LANE_LEFT, LANE_RIGHT, FASTER, SLOWER [from memory]


def availability_of(env, mdp):
    """(table bool [S, A], listing order or None) of an environment exposing ``get_available_actions`` -- (None, None)
    when it has no such method.  The reference asks the env object node by node (mcts.py:59-97, deterministic.py:32-35)
    and creates a node's children IN THE ORDER THE ENV LISTS THEM; a device planner needs the whole table and that
    order, taken from, in this order:

    1. ``mdp.available`` -- table environments (rl_agents_amd.envs.MaskedFiniteMDPEnv), listed ascending;
    2. ``env.unwrapped.available_table(mdp)`` -- the documented hook for environments that keep the restriction on the
       env object: return bool [S, A] in the MDP's state numbering, or ``(table, order)`` with ``order`` the
       permutation of action ids in which ``get_available_actions()`` lists them;
    3. for MDPs carrying ``original_shape = (V, L, T)`` with 5 actions -- what highway-env's ``to_finite_mdp()`` returns
       (value_iteration.py:12-21; the restriction stays on the env there) -- the lane / speed edge rule of
       :func:`grid_available`, listed IDLE first (:data:`GRID_LISTING_ORDER`).

    A derived table (2, 3) is cross-checked against the env itself for the state it is in, on every call -- the listed
    actions AND their order: if ``get_available_actions()`` disagrees the planner refuses (``ValueError``) rather than
    plan on a guessed restriction.  Anything else raises ``TypeError``."""
    base = getattr(env, "unwrapped", env)
    if not hasattr(base, "get_available_actions"):
        return None, None
    reward = np.asarray(mdp.reward)
    n_states, n_actions = reward.shape[-2:]
    available, order, source = getattr(mdp, "available", None), None, "mdp.available"
    if available is None and callable(getattr(base, "available_table", None)):
        available, source = base.available_table(mdp), "env.available_table(mdp)"
        if isinstance(available, tuple):
            available, order = available
    shape = getattr(mdp, "original_shape", None)
    if available is None and shape is not None and len(shape) == 3 and n_actions == 5 and int(np.prod(shape)) == n_states:
        available, order, source = grid_available(shape), GRID_LISTING_ORDER, "the (V, L, T) grid rule on mdp.original_shape"
    if available is None:
        raise TypeError("the environment restricts its available actions but neither its finite MDP has an `available` "
                        "[S, A] table, nor the env an `available_table(mdp)` hook, nor the MDP a (V, L, T) "
                        "`original_shape`: the device planners cannot query get_available_actions() node by node")
    available = np.asarray(available).astype(bool).reshape(n_states, n_actions)
    if order is not None:
        order = np.asarray(order, dtype=np.int64).reshape(-1)
        if sorted(order.tolist()) != list(range(n_actions)):
            raise ValueError("the listing order from {} is not a permutation of the {} actions".format(source, n_actions))
        if np.array_equal(order, np.arange(n_actions)):
            order = None
    if source != "mdp.available":
        listed = [int(a) for a in base.get_available_actions()]
        seq = range(n_actions) if order is None else order
        row = [int(a) for a in seq if available[int(mdp.state), int(a)]]
        if listed != row:
            raise ValueError("availability from {} lists actions {} in state {} but the environment's "
                             "get_available_actions() returns {}".format(source, row, int(mdp.state), listed))
    return available, order


def available_actions_of(env, mdp):
    """The table half of :func:`availability_of` (bool [S, A], or None for an unrestricted environment)."""
    return availability_of(env, mdp)[0]


def is_cartpole(env):
    """Closed-form CartPole (rl_agents_amd.envs.CartPoleEnv or anything exposing the same surface)."""
    base = getattr(env, "unwrapped", env)
    return hasattr(base, "cartpole_params") and hasattr(base, "state")


def env_root_state(env):
    """(state, steps taken) of an environment: what a clone of it consists of on the device -- a state index
    for a finite MDP, the (x, x_dot, theta, theta_dot) tuple for CartPole."""
    base = getattr(env, "unwrapped", env)
    if is_cartpole(env):
        return tuple(float(v) for v in base.state), int(getattr(base, "steps", 0) or 0)
    mdp = finite_mdp_of(env)
    return int(mdp.state), int(getattr(base, "steps", 0) or 0)


def env_max_steps(env):
    base = getattr(env, "unwrapped", env)
    cfg = getattr(base, "config", None)
    if isinstance(cfg, dict):
        return int(cfg.get("max_steps", 0) or 0)
    return 0


class ModelCache(object):
    """Device models of one context, keyed by table content; least-recently-used eviction."""

    def __init__(self, ctx=None, capacity=8):
        self._ctx = ctx
        self.capacity = capacity
        self._models = {}
        self._order = []
        self._by_token = {}         # (tables token, static key) -> (version counter, cache key) of the model that holds them
        self._version_alias = {}    # version key -> cache key (content hash, or the version key itself for patched models)
        self.uploads = 0
        self.row_updates = 0        # rows patched by delta uploads
        self._version_hits = {}     # version key -> lookups served without hashing (the sampled content guard below)
        self._untrusted = set()     # tokens caught changing content under an unchanged version: hashed on every lookup
        self.version_violations = 0

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = runtime.get_context()
        return self._ctx

    def get(self, spec):
        if spec.version is not None and not os.environ.get("MP_NO_TABLE_VERSIONS"):
            return self._get_versioned(spec)
        return self._get_keyed(spec.key(), spec)

    def _get_versioned(self, spec):
        """The environment vouches for its tables' identity (MDP.tables_version): no hashing.  A newer version of tables this
        cache already holds, with a known set of changed rows, PATCHES the device model (mp_model_update_rows: the delta
        upload of SURVEY.md 8 f-2) instead of uploading a new one."""
        token, counter = spec.version
        if token in self._untrusted:
            return self._get_keyed(spec.key(), spec)
        static = spec.static_key()
        key = ("version", token, counter, static)
        alias = self._version_alias.get(key)
        if alias is not None and alias in self._models:
            # SAMPLED GUARD of the promise "equal versions = identical tables": an MDP holds read-only VIEWS of the arrays it
            # was given, so an in-place edit through the caller's own reference (a config array, say) changes the tables under
            # an unchanged version.  A model keyed by content (alias != key; patched models own private copies and cannot be
            # aliased) is re-hashed on its 4th hit and every 64th after (every lookup under MP_VERIFY_TABLE_VERSIONS=1); a
            # mismatch warns, stops trusting that token for good and serves the model of the tables as they are NOW.
            hits = self._version_hits[key] = self._version_hits.get(key, 0) + 1
            if alias != key and (hits == 4 or hits % 64 == 0 or os.environ.get("MP_VERIFY_TABLE_VERSIONS")):
                now = spec.key()
                if now != alias:
                    import warnings
                    warnings.warn("the tables of a finite MDP changed while its tables_version {} did not (an array edited in "
                                  "place behind MDP.touch()/edit_rows?): its versions are no longer trusted and every lookup "
                                  "hashes the tables again".format((token, counter)), RuntimeWarning, stacklevel=3)
                    self._untrusted.add(token)
                    self.version_violations += 1
                    self._version_alias.pop(key, None)
                    return self._get_keyed(now, spec)
            if len(self._version_hits) > 256:
                self._version_hits = {k: v for k, v in self._version_hits.items() if k in self._version_alias}
            return self._get_keyed(alias, spec)
        held = self._by_token.get((token, static))
        if held is not None and held[1] in self._models and spec.mode == "deterministic" and spec.dirty_rows_since is not None \
                and spec.transition.ndim == 2 and not os.environ.get("MP_NO_DELTA_UPLOAD"):
            rows = spec.dirty_rows_since(held[0])
            if rows is not None and len(rows) * 4 <= spec.n_states:
                model = self._models.pop(held[1])
                self._order.remove(held[1])
                if len(rows):
                    # (terminal flags ride along only when they changed: a flag change re-packs the whole MDP's records)
                    old = getattr(model.spec, "terminal", None)
                    same_flags = old is not None and old.shape == spec.terminal.shape and np.array_equal(old[rows], spec.terminal[rows])
                    model.update_rows(rows, spec.transition[rows], spec.reward[rows], None if same_flags else spec.terminal[rows])
                    self.row_updates += int(len(rows))
                    model._vi_cache = None              # (solutions / policies derived from the old tables)
                    model.epoch = getattr(model, "epoch", 0) + 1
                model.spec = spec
                self._models[key] = model                # (a patched model is known by its version only: its content hash
                self._order.append(key)                  # would have to be recomputed, which is what the version spares)
                self._version_alias[key] = key
                self._by_token[(token, static)] = (counter, key)
                return model
        # tables this cache has not seen under that identity (a NEW MDP object per conversion -- highway-env's to_finite_mdp --
        # or the first call): their CONTENT decides, once; the version then names the model that holds them
        alias = self._version_alias.get(key)
        if alias is None or alias not in self._models:
            alias = spec.key()
            if len(self._version_alias) > 64:
                self._version_alias = {k: v for k, v in self._version_alias.items() if v in self._models}
            self._version_alias[key] = alias
        model = self._get_keyed(alias, spec)
        self._by_token[(token, static)] = (counter, alias)
        if len(self._by_token) > 4 * self.capacity:
            self._by_token = {k: v for k, v in self._by_token.items() if v[1] in self._models}
        return model

    def _get_keyed(self, key, spec):
        model = self._models.get(key)
        if model is None:
            model = self._upload(spec)
            model.spec = spec                       # (what it was uploaded from: MCTS.get_visits replays plans on a private context)
            self.uploads += 1
            self._models[key] = model
            if len(self._order) >= self.capacity:
                old = self._order.pop(0)
                self._models.pop(old).close()
        else:
            self._order.remove(key)
        self._order.append(key)
        return model

    def get_cartpole(self, params):
        key = ("cartpole",) + tuple(sorted(params.items()))
        model = self._models.get(key)
        if model is None:
            model = self.ctx.load_cartpole(params)
            self.uploads += 1
            self._models[key] = model
            self._order.append(key)
        return model

    def _upload(self, spec):
        if spec.mode == "deterministic":
            model = self.ctx.load_table(spec.transition, spec.reward, spec.terminal, done_rule=spec.done_rule,
                                        max_steps=spec.max_steps, available=spec.available)
            model.action_order = spec.action_order      # None, or: column j of the device tables = action order[j]
            return model
        if spec.mode == "stochastic":
            model = self.ctx.load_dense(spec.transition, spec.reward, spec.terminal)
        else:
            model = self.ctx.load_sparse(spec.transition, spec.next, spec.reward, spec.terminal)
        model.action_order = spec.action_order          # (as above: the planners map labels back at the boundary)
        return model
