"""CPU tests of the host side: config merge, factory, policies, receding horizon, model extraction / cache,
C-ABI symbol table, budget allocation.  No GPU compute is called."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

from rl_agents_amd import device_model, native
from rl_agents_amd.agents.common.factory import agent_factory, load_agent_config, preprocess_env
from rl_agents_amd.agents.tree_search import mcts as mcts_mod
from rl_agents_amd.agents.tree_search.abstract import AbstractTreeSearchAgent, build_tree, np_random
from rl_agents_amd.configuration import Configurable
from rl_agents_amd.envs import FiniteMDPEnv, generators

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads and exports exactly the entry points include/mi355plan.h declares."""
    header = open(os.path.join(REPO, "include", "mi355plan.h")).read()
    declared = set(re.findall(r"\b(mp_[a-z0-9_]+)\s*\(", header))
    declared -= {"mp_ctx", "mp_model"}
    lib = ctypes.CDLL(native.lib_path())
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(native.SIGNATURES), declared ^ set(native.SIGNATURES)
    assert native.load().mp_abi_version() == 7


def test_context_fails_loudly_without_gpu_or_library(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(native.NativeError):
        native.Context(0)
    monkeypatch.setattr(native, "_LIB", None)
    monkeypatch.setattr(native._build, "LIB_PATH", "/nonexistent/libmi355plan.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load()


def test_olop_allocation_matches_golden(golden):
    z = golden["misc"]
    for (b, g), (e, h) in zip(z["alloc/in"], z["alloc/out"]):
        assert native.olop_allocation(int(b), float(g)) == (int(e), int(h))
    with pytest.raises(ValueError):
        native.olop_allocation(1, 0.8)


def test_configurable_merges_both_ways():
    class C(Configurable):
        @classmethod
        def default_config(cls):
            return {"a": 1, "nested": {"x": 1, "y": 2}}
    user = {"nested": {"y": 5}, "extra": True}
    c = C(user)
    assert c.config == {"a": 1, "nested": {"x": 1, "y": 5}, "extra": True}
    assert user == c.config and user is not c.config          # the caller's dict is completed in place


def test_agent_factory_and_json_inheritance(tmp_path):
    base = tmp_path / "base.json"
    child = tmp_path / "child.json"
    base.write_text(json.dumps({"__class__": "<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>",
                                "budget": 300, "gamma": 0.9}))
    child.write_text(json.dumps({"base_config": str(base), "budget": 1000}))
    cfg = load_agent_config(str(child))
    assert cfg["budget"] == 1000 and cfg["gamma"] == 0.9 and "base_config" not in cfg
    env = FiniteMDPEnv(generators.gridworld())
    agent = agent_factory(env, cfg)
    assert isinstance(agent, mcts_mod.MCTSAgent)
    assert (agent.planner.config["episodes"], agent.planner.config["horizon"]) == native.olop_allocation(1000, 0.9)
    with pytest.raises(ValueError):
        agent_factory(env, {"budget": 3})


def test_mcts_defaults_and_policies():
    env = FiniteMDPEnv(generators.gridworld())
    agent = mcts_mod.MCTSAgent(env, {"budget": 400, "gamma": 0.95})
    pc = agent.planner.config
    assert pc["temperature"] == 2 / (1 - 0.8)                  # from the DEFAULT gamma, as in the reference
    assert pc["step_strategy"] == "reset" and pc["closed_loop"] is False
    agent2 = mcts_mod.MCTSAgent(env, {"budget": 1000, "horizon": 30})
    assert agent2.planner.config["episodes"] == 33            # documented deviation (reference: KeyError)
    np.testing.assert_array_equal(mcts_mod.policy_probabilities({"type": "random"}, 4), np.ones(4) / 4)
    p = mcts_mod.policy_probabilities({"type": "preference", "action": 1, "ratio": 3}, 4)
    np.testing.assert_array_equal(p, np.ones(4) / (4 - 1 + 3) * np.array([1, 3, 1, 1]))
    np.testing.assert_array_equal(mcts_mod.policy_probabilities({"type": "preference", "action": 9, "ratio": 3}, 4),
                                  np.ones(4) / 4)
    with pytest.raises(ValueError):
        mcts_mod.MCTSAgent(env, {"prior_policy": {"type": "nope"}})
    assert mcts_mod.MCTSAgent(env, {"closed_loop": True}).planner.config["closed_loop"] is True
    # policies over restricted action sets, state by state as the reference's policy functions return them (mcts.py:46-97)
    from tests.helpers import reference_policy_lists
    avail = generators.random_available(23, 5, seed=3, rate=0.5)
    for pol in ({"type": "random"}, {"type": "random_available"}, {"type": "preference", "action": 2, "ratio": 3},
                {"type": "preference", "action": 4, "ratio": 2.5}, {"type": "preference", "action": 7, "ratio": 2}):
        table, listed, slots = mcts_mod.policy_tables(pol, avail)
        assert slots is None                                   # ascending listing = the column order
        ref = reference_policy_lists(pol, avail)
        for s in range(23):
            assert list(np.flatnonzero(listed[s])) == ref["actions"][s]
            assert np.array_equal(table[s, ref["actions"][s]], ref["p"][s]) and table[s].sum() == pytest.approx(1.0)
            assert (table[s, ~listed[s]] == 0).all()
    # ... and on an environment that lists in a non-ascending order (IDLE first): columns in the PRIOR policy's order, the
    # other policy's listing as slots -- state by state the reference's lists again
    order = np.array([1, 0, 2, 3, 4])
    rank = np.empty(5, dtype=np.int64)
    rank[order] = np.arange(5)
    for col_ids in (order, np.arange(5)):                      # prior lists like the env / prior is `random`
        cols = avail[:, col_ids]
        for pol in ({"type": "random"}, {"type": "random_available"}, {"type": "preference", "action": 2, "ratio": 3}):
            dev = dict(pol, action=int(np.flatnonzero(col_ids == pol["action"])[0])) if pol["type"] == "preference" else pol
            table, listed, slots = mcts_mod.policy_tables(dev, cols, col_ids, rank)
            ref = reference_policy_lists(pol, avail, order)
            for s in range(23):
                seq = np.arange(5) if slots is None else slots[s]
                mine = [int(col_ids[c]) for c in seq if table[s, c] > 0]
                assert mine == ref["actions"][s], (pol, s)
                assert np.array_equal([table[s, c] for c in seq if table[s, c] > 0], ref["p"][s])


def test_receding_horizon_bookkeeping():
    calls = []

    class FakePlanner(object):
        def plan(self, state, observation):
            calls.append(observation)
            return [0, 1, 2, 3]

        def step_tree(self, actions):
            pass

        def step_by_reset(self):
            pass

    class Agent(AbstractTreeSearchAgent):
        def make_planner(self):
            return FakePlanner()

    agent = Agent(env=object(), config={"receding_horizon": 3})
    assert [agent.plan(i)[0] for i in range(7)] == [0, 1, 2, 0, 1, 2, 0]
    assert calls == [0, 3, 6]                                   # re-plans every 3rd step
    agent.reset()
    assert agent.remaining_horizon == 0 and agent.steps == 0


def test_seeding_matches_gymnasium_convention():
    g, entropy = np_random(5)
    ref = np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    assert entropy == 5 and g.random() == ref.random()
    st = native.rng_state_from_generator(ref)
    g2 = np.random.Generator(np.random.PCG64(0))
    native.generator_set_state(g2, st)
    assert g2.random() == ref.random()
    with pytest.raises(ValueError):
        np_random(-1)


def test_model_extraction_and_cache():
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    env = FiniteMDPEnv(dict(cfg, state=7, max_steps=12))
    env.reset()
    env.step(1)
    mdp = device_model.finite_mdp_of(env)
    spec = device_model.spec_from_mdp(mdp, max_steps=device_model.env_max_steps(env))
    assert spec.mode == "deterministic" and spec.n_states == 120 and spec.n_actions == 5 and spec.max_steps == 12
    assert device_model.env_root_state(env) == (int(cfg["transition"][7, 1]), 1)
    with pytest.raises(TypeError):
        device_model.finite_mdp_of(object())

    class FakeCtx(object):
        def load_table(self, *a, **k):
            class M(object):
                def close(self):
                    pass
            return M()

    cache = device_model.ModelCache(FakeCtx(), capacity=2)
    m1 = cache.get(spec)
    assert cache.get(device_model.spec_from_mdp(mdp, max_steps=12)) is m1 and cache.uploads == 1
    changed = np.array(cfg["reward"], copy=True)
    changed[0, 0] += 0.125
    spec2 = device_model.TableSpec("deterministic", cfg["transition"], changed, cfg["terminal"])
    assert cache.get(spec2) is not m1 and cache.uploads == 2
    with pytest.raises(ValueError):
        device_model.TableSpec("bogus", cfg["transition"], cfg["reward"])


def test_preprocess_env_and_tree_view():
    class Env(object):
        unwrapped = None

        def simplify(self):
            return "simplified"
    e = Env()
    e.unwrapped = e
    assert preprocess_env(e, [{"method": "simplify"}]) == "simplified"
    assert preprocess_env(e, [{"method": "missing"}]) is e
    arrays = dict(parent=np.array([-1, 0, 0, 1, 1]), action=np.array([-1, 0, 1, 0, 1]),
                  count=np.array([5, 3, 2, 2, 1]), value=np.array([.5, .6, .4, .7, .1]))
    root = build_tree(arrays, "value")
    assert sorted(root.children) == [0, 1] and root.children[0].children[1].path() == [0, 1]
    assert root.children[0].children[0].depth == 2 and root.children[1].is_leaf() and root.get_value() == .5
    assert root.selection_rule() == 0 and root.children[0].selection_rule() == 0 and root.children[1].selection_rule() is None
    with_prior = build_tree(arrays, "value", prior=[0.25, 0.75])
    child = with_prior.children[1]
    assert child.prior == 0.75 and child.selection_strategy(10.0) == .4 + 10.0 * 2 * 0.75 / (2 + 1)


def test_shard_bounds_cover_everything():
    from rl_agents_amd.distributed import shard_bounds
    for n in (0, 1, 7, 8, 4096, 8193):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(b[1] == c[0] for b, c in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_tabulate_prior_agent_queries_like_the_reference():
    """mcts_with_prior.py:47-54: act(observation) then action_distribution(observation), keys in any order."""
    from rl_agents_amd.agents.tree_search.mcts_with_prior import tabulate_prior_agent

    class Prior(object):
        def __init__(self):
            self.calls = []

        def act(self, s):
            self.calls.append(("act", s))

        def action_distribution(self, s):
            self.calls.append(("dist", s))
            return {2: 0.5, 0: 0.25 + 0.01 * s, 1: 0.25 - 0.01 * s}

    prior = Prior()
    table = tabulate_prior_agent(prior, 3, 3)
    assert prior.calls == [("act", 0), ("dist", 0), ("act", 1), ("dist", 1), ("act", 2), ("dist", 2)]
    np.testing.assert_array_equal(table[1], [0.26, 0.24, 0.5])

    class Fast(object):
        def policy_table(self):
            return np.full((3, 3), 1 / 3)

    assert tabulate_prior_agent(Fast(), 3, 3).shape == (3, 3)
    with pytest.raises(ValueError):
        tabulate_prior_agent(Fast(), 4, 3)

    class Partial(Prior):
        def action_distribution(self, s):
            return {0: 1.0}

    with pytest.raises(ValueError):
        tabulate_prior_agent(Partial(), 2, 3)


def test_seed_sequence_states_equal_numpy():
    """mp_seed_sequence_states restates numpy's SeedSequence (pool of 4, entropy words little-endian, no spawn key) and
    PCG64's seeding from generate_state(4, uint64): records identical to the generators numpy builds -- for
    SeedSequence(k), SeedSequence([entropy, k]) with entropies of one and two words, zero, and more words than the pool."""
    from rl_agents_amd import native

    def numpy_records(make, n):
        out = np.zeros((n, 6), dtype=np.uint64)
        for i in range(n):
            out[i] = native.rng_state_from_generator(np.random.Generator(np.random.PCG64(np.random.SeedSequence(make(i)))))
        return out
    assert np.array_equal(native.seed_sequence_states((), 0, 70), numpy_records(lambda i: i, 70))
    assert np.array_equal(native.seed_sequence_states((), 2 ** 32 - 3, 8), numpy_records(lambda i: 2 ** 32 - 3 + i, 8))
    for e in (0, 1234, 2 ** 40 + 17, (1 << 63) - 1):
        assert np.array_equal(native.seed_sequence_states([e], 5, 33), numpy_records(lambda i: [e, 5 + i], 33)), e
    assert np.array_equal(native.seed_sequence_states([5, 6, 2 ** 70, 9], 0, 9), numpy_records(lambda i: [5, 6, 2 ** 70, 9, i], 9))
    # the planners' batch streams: root i of a batch draws from SeedSequence([entropy, first_root + i])
    from rl_agents_amd.agents.tree_search.abstract import AbstractPlanner

    class P(object):
        _entropy = 340282366920938463463374607431768211455 % (1 << 63)
    got = AbstractPlanner.batch_rng_states(P(), 5, first_root=11)
    assert np.array_equal(got, numpy_records(lambda i: [P._entropy, 11 + i], 5))


def test_availability_of_env_side_restrictions():
    """device_model.available_actions_of: table from mdp.available, from the env's `available_table` hook, or from the
    (V, L, T) grid rule on `original_shape`; derived tables are cross-checked against the env's own answer for the state
    it is in; an env that restricts actions with none of the three raises TypeError (no guessing)."""
    import pytest
    from rl_agents_amd import device_model
    from rl_agents_amd.envs import HighwayLikeEnv, MaskedFiniteMDPEnv, generators
    env = HighwayLikeEnv(3, 4, 10, seed=3, state=41)
    mdp = env.to_finite_mdp()
    assert not hasattr(mdp, "available")
    table = device_model.available_actions_of(env, mdp)
    assert table.shape == (120, 5) and table.dtype == bool
    _, order = device_model.availability_of(env, mdp)
    assert list(order) == [1, 0, 2, 3, 4]          # IDLE first: the order the env lists its actions in
    for s in range(120):                # the rule reproduces the env's own LIST (actions and order) in every state
        e = HighwayLikeEnv(table=env.table, state=s)
        assert e.get_available_actions() == [a for a in order if table[s, a]]
    # the spec a planner uploads is in listing order: column j of every table = action order[j]
    spec = device_model.spec_from_mdp(mdp, available=table, action_order=order)
    assert np.array_equal(spec.transition, np.asarray(mdp.transition)[:, order])
    assert np.array_equal(spec.available.astype(bool), table[:, order]) and list(spec.action_order) == [1, 0, 2, 3, 4]
    assert spec.key() != device_model.spec_from_mdp(mdp, available=table).key()
    # hook: an env may state its restriction as a table itself

    class Hooked(HighwayLikeEnv):
        def to_finite_mdp(self):
            m = super().to_finite_mdp()
            del m.original_shape
            return m

        def available_table(self, mdp):
            return generators.highway_available(self.table), [1, 0, 2, 3, 4]
    h = Hooked(3, 4, 10, seed=3, state=7)
    assert np.array_equal(device_model.available_actions_of(h, h.to_finite_mdp()), table)
    assert list(device_model.availability_of(h, h.to_finite_mdp())[1]) == [1, 0, 2, 3, 4]

    class WrongOrder(Hooked):           # right actions, wrong listing order: refused as well
        def available_table(self, mdp):
            return generators.highway_available(self.table)
    with pytest.raises(ValueError):
        device_model.available_actions_of(WrongOrder(3, 4, 10, seed=3, state=10), WrongOrder(3, 4, 10, seed=3, state=10).to_finite_mdp())
    # a derived table that contradicts the env is refused

    class Liar(Hooked):
        def available_table(self, mdp):
            return np.ones((120, 5), dtype=bool)
    with pytest.raises(ValueError):
        device_model.available_actions_of(Liar(3, 4, 10, seed=3, state=0), Liar(3, 4, 10, seed=3, state=0).to_finite_mdp())
    # nothing to derive from

    class Opaque(Hooked):
        available_table = None

        def to_finite_mdp(self):
            m = super().to_finite_mdp()
            return m
    o = Opaque(3, 4, 10, seed=3)
    o.__class__ = type("Opaque2", (HighwayLikeEnv,), {"to_finite_mdp": Hooked.to_finite_mdp})
    with pytest.raises(TypeError):
        device_model.available_actions_of(o, o.to_finite_mdp())
    # table environments keep working, an unrestricted env gives None
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    m = MaskedFiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"],
                                terminal=cfg["terminal"], available=generators.highway_available(cfg)))
    assert np.array_equal(device_model.available_actions_of(m, m.mdp), table)
    from rl_agents_amd.envs import FiniteMDPEnv
    f = FiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"]))
    assert device_model.available_actions_of(f, f.mdp) is None


def test_vi_exact_plan_replays_numpy_add_reduce():
    """mp_vi_exact_plan (host only): the leaves and additions the bit-exact dense backup follows, replayed here the way
    the kernel does it -- eight strided accumulators per leaf, the three-level combine, the remainder one by one, the
    recursion's additions height by height -- equal numpy.add.reduce bit for bit on every row length shape."""
    from rl_agents_amd import native
    g = np.random.Generator(np.random.PCG64(7))
    for n in [1, 5, 7, 8, 9, 15, 16, 100, 127, 128, 129, 130, 255, 256, 257, 1000, 1029, 2500, 4099, 8191, 8192, 8193, 8200, 10000,
              16384, 16385, 24576, 31250, 50000]:
        leaves, nodes, hoff, nb_max = native.vi_exact_plan(n)
        assert tuple(leaves[0]) == (0, 0) and leaves[-1].sum() == n and np.all(leaves[1:, 0] == leaves[:-1].sum(axis=1))
        assert np.all(leaves[:, 1] <= 128) and nb_max == max(1, int(leaves[:, 1].max()) // 8) and len(nodes) == len(leaves) - 1
        assert np.all(leaves[:-1, 1] % 8 == 0)                  # only the last leaf can have a remainder
        assert len(hoff) - 1 >= -(-n // 8192)                   # one running-sum addition per 8192-element piece
        for trial in range(3):
            a = g.standard_normal(n) * np.exp(g.standard_normal(n) * 3)
            slots = np.zeros(len(leaves) + len(nodes))
            for l, (off, ln) in enumerate(leaves):
                nb = ln // 8
                if nb > 0:
                    r = a[off:off + 8].copy()
                    for i in range(1, nb):
                        r = r + a[off + 8 * i:off + 8 * i + 8]
                    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
                else:
                    res = 0.0
                for i in range(off + 8 * nb, off + ln):
                    res = res + a[i]
                slots[l] = res
            written = np.zeros(len(slots), dtype=bool)
            written[:len(leaves)] = True
            for h in range(len(hoff) - 1):
                ks = range(hoff[h], hoff[h + 1])
                assert all(written[nodes[k, 0]] and written[nodes[k, 1]] for k in ks)   # operands come from lower heights
                for k in ks:
                    slots[len(leaves) + k] = slots[nodes[k, 0]] + slots[nodes[k, 1]]
                for k in ks:
                    written[len(leaves) + k] = True
            assert hoff[-1] == len(nodes) and written.all()
            assert slots[-1].tobytes() == np.add.reduce(a).tobytes(), (n, trial)


def test_listing_order_permutes_every_model_kind():
    """A non-ascending listing order (round 4: also on stochastic / sparse models): the spec's columns follow the order
    -- transition, next states, rewards, availability -- and availability_of reads table and order off the env's hook,
    cross-checked against get_available_actions()."""
    from rl_agents_amd.envs import OrderedMaskedFiniteMDPEnv
    order = [1, 0, 2, 3, 4]
    for cfg in (generators.random_sparse(20, 5, 2, seed=1), generators.random_stochastic(12, 5, seed=2),
                generators.random_deterministic(15, 5, seed=3)):
        avail = generators.random_available(cfg["reward"].shape[0], 5, seed=7, rate=0.4)
        env = OrderedMaskedFiniteMDPEnv(dict({k: v for k, v in cfg.items() if k != "original_shape"}, state=3,
                                             available=avail, listing_order=order))
        env.reset()
        listed = env.get_available_actions()
        assert listed == [a for a in order if avail[3, a]] and not hasattr(env.mdp, "available")
        table, got = device_model.availability_of(env, env.mdp)
        assert np.array_equal(table, avail) and list(got) == order
        spec = device_model.spec_from_mdp(env.mdp, available=table, action_order=got)
        assert list(spec.action_order) == order
        assert np.array_equal(spec.reward, np.asarray(cfg["reward"])[:, order])
        assert np.array_equal(spec.available.astype(bool), avail[:, order])
        if cfg["mode"] == "deterministic":
            assert np.array_equal(spec.transition, np.asarray(cfg["transition"])[:, order])
        else:
            assert np.array_equal(spec.transition, np.asarray(cfg["transition"])[:, order, :])
        if cfg["mode"] == "sparse":
            assert np.array_equal(spec.next, np.asarray(cfg["next"])[:, order, :])
        assert device_model.spec_from_mdp(env.mdp, available=table, action_order=got).key() == spec.key()
        assert device_model.spec_from_mdp(env.mdp, available=table).key() != spec.key()
    with pytest.raises(ValueError):
        OrderedMaskedFiniteMDPEnv(dict(generators.random_sparse(5, 3, 2, seed=1), listing_order=[0, 0, 1]))


def test_mdp_tables_version_protocol():
    """MDP.tables_version (rl_agents_amd/envs/finite_mdp.py): equal versions promise identical tables -- assigning a table
    or editing rows advances it, element assignment through the MDP raises, edit_rows remembers the rows, copies get their
    own identity, and a HighwayLikeEnv's re-conversion returns the same version while its table stays what it was."""
    import copy
    from rl_agents_amd.envs import ChangingHighwayEnv, FiniteMDPEnv, HighwayLikeEnv, generators
    cfg = {k: v for k, v in generators.highway_shaped(3, 4, 10, seed=1).items() if k != "original_shape"}
    env = FiniteMDPEnv(cfg)
    m = env.mdp
    v0 = m.tables_version
    assert m.tables_version == v0 and m.dirty_rows_since(v0[1]).size == 0
    with pytest.raises(ValueError):
        m.reward[0, 0] = 0.25                         # read-only view: no silent stale device model
    m.edit_rows([5, 3], reward=np.full((2, 5), 0.5), terminal=[True, False])
    v1 = m.tables_version
    assert v1[0] == v0[0] and v1[1] == v0[1] + 1
    np.testing.assert_array_equal(m.dirty_rows_since(v0[1]), [3, 5])
    assert m.reward[5, 0] == 0.5 and bool(m.terminal[5]) and not bool(m.terminal[3])
    m.edit_rows([7], transition=np.zeros((1, 5), np.int64))
    np.testing.assert_array_equal(m.dirty_rows_since(v0[1]), [3, 5, 7])
    np.testing.assert_array_equal(m.dirty_rows_since(v1[1]), [7])
    m.reward = np.array(m.reward)                     # a whole table assigned: rows unknown
    assert m.tables_version[1] == v1[1] + 2 and m.dirty_rows_since(v1[1]) is None
    twin = copy.deepcopy(env)
    assert twin.mdp.tables_version[0] != m.tables_version[0]
    np.testing.assert_array_equal(twin.mdp.reward, m.reward)
    twin.mdp.edit_rows([0], reward=np.ones((1, 5)))
    assert m.reward[0, 0] != 1.0                      # the copy's arrays are its own
    h = HighwayLikeEnv()
    assert h.to_finite_mdp().tables_version == h.to_finite_mdp().tables_version
    c = ChangingHighwayEnv(table_seed=3)
    before = c.to_finite_mdp().tables_version
    c.step(1)
    assert c.to_finite_mdp().tables_version != before


def test_model_cache_keys_by_version_without_hashing(monkeypatch):
    """device_model.ModelCache with a versioned spec: no content hash is computed (TableSpec.key is never called), a changed
    version with known dirty rows patches the held model (update_rows), an unknown change uploads."""
    from rl_agents_amd import device_model
    from rl_agents_amd.envs import FiniteMDPEnv, generators

    class FakeModel(object):
        def __init__(self):
            self.rows = []

        def update_rows(self, rows, t, r, term):
            self.rows.append(np.asarray(rows).copy())

        def close(self):
            pass

    cache = device_model.ModelCache(ctx=object())
    monkeypatch.setattr(cache, "_upload", lambda spec: FakeModel())
    hashed = []
    real_key = device_model.TableSpec.key
    monkeypatch.setattr(device_model.TableSpec, "key", lambda self: (hashed.append(1), real_key(self))[1])
    cfg = {k: v for k, v in generators.highway_shaped(3, 4, 10, seed=1).items() if k != "original_shape"}
    env = FiniteMDPEnv(cfg)
    m1 = cache.get(device_model.spec_from_mdp(env.mdp))
    assert len(hashed) == 1                           # tables never seen under this identity: their content decides, once
    assert cache.get(device_model.spec_from_mdp(env.mdp)) is m1 and cache.uploads == 1 and len(hashed) == 1
    # another MDP OBJECT with the same tables (a re-conversion that builds a new object each time): one more hash, no upload
    twin = FiniteMDPEnv(cfg)
    assert cache.get(device_model.spec_from_mdp(twin.mdp)) is m1 and cache.uploads == 1 and len(hashed) == 2
    assert cache.get(device_model.spec_from_mdp(twin.mdp)) is m1 and len(hashed) == 2
    env.mdp.edit_rows([4, 9], reward=np.zeros((2, 5)))
    m2 = cache.get(device_model.spec_from_mdp(env.mdp))
    assert m2 is m1 and cache.uploads == 1 and cache.row_updates == 2
    np.testing.assert_array_equal(m1.rows[0], [4, 9])
    assert cache.get(device_model.spec_from_mdp(env.mdp)) is m1 and len(m1.rows) == 1 and len(hashed) == 2
    env.mdp.reward = np.array(env.mdp.reward) * 0.5   # unknown change: hashed once, a new upload
    m3 = cache.get(device_model.spec_from_mdp(env.mdp))
    assert m3 is not m1 and cache.uploads == 2 and len(hashed) == 3


def test_mdp_edit_rows_never_writes_into_the_callers_arrays():
    """an MDP built on a SLICE of a larger array (`stack[t]`) must not write through `.base` -- numpy
    collapses a view's base to the owner of the memory, i.e. the whole stack.  edit_rows takes a private copy."""
    from rl_agents_amd.envs.finite_mdp import DeterministicMDP
    from rl_agents_amd.envs import generators
    cfgs = [generators.highway_shaped(2, 3, 5, seed=i) for i in range(3)]
    stack_t, stack_r = np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs])
    keep_t, keep_r = stack_t.copy(), stack_r.copy()
    m = DeterministicMDP(stack_t[1], stack_r[1], terminal=cfgs[1]["terminal"])
    m.edit_rows([2], reward=np.ones((1, stack_r.shape[2])))
    assert m.reward.shape == stack_r[1].shape and m.transition.shape == stack_t[1].shape
    np.testing.assert_array_equal(stack_r, keep_r)            # the caller's stack is untouched ...
    np.testing.assert_array_equal(stack_t, keep_t)
    assert (m.reward[2] == 1.0).all() and np.array_equal(m.reward[3], keep_r[1][3])     # ... and the MDP sees its edit
    m.edit_rows([3], reward=np.zeros((1, stack_r.shape[2])), transition=np.zeros((1, stack_r.shape[2]), np.int64))
    assert (m.reward[2] == 1.0).all() and (m.reward[3] == 0.0).all() and (m.transition[3] == 0).all()
    np.testing.assert_array_equal(stack_t, keep_t)
    np.testing.assert_array_equal(m.dirty_rows_since(m.tables_version[1] - 2), [2, 3])
    # a read-only array handed over: same thing
    ro = keep_r[0].copy()
    ro.setflags(write=False)
    m2 = DeterministicMDP(keep_t[0], ro)
    m2.edit_rows([0], reward=np.full((1, ro.shape[1]), 0.25))
    assert ro[0, 0] == keep_r[0][0, 0] and m2.reward[0, 0] == 0.25


def test_mdp_objects_sharing_a_version_token_part_ways_when_edited():
    """every to_finite_mdp() of one highway-like table shares a (token, 0) version AND the table arrays; two
    such objects edited differently must not both become (token, 1), and neither edit may reach the shared table."""
    from rl_agents_amd.envs import HighwayLikeEnv
    h = HighwayLikeEnv()
    a, b, c = h.to_finite_mdp(), h.to_finite_mdp(), h.to_finite_mdp()
    assert a.tables_version == b.tables_version == c.tables_version
    shared = np.array(h.table["reward"])
    a.edit_rows([1], reward=np.full((1, 5), 0.125))
    b.edit_rows([1], reward=np.full((1, 5), 0.875))
    assert a.tables_version != b.tables_version and a.tables_version[0] != c.tables_version[0] != b.tables_version[0]
    assert a.reward[1, 0] == 0.125 and b.reward[1, 0] == 0.875
    np.testing.assert_array_equal(h.table["reward"], shared)
    np.testing.assert_array_equal(c.reward, shared)
    assert h.to_finite_mdp().tables_version == c.tables_version      # the family's version still names the unedited table
    assert a.dirty_rows_since(0) is None                             # (a fresh identity: its first version is a whole table)
    a.edit_rows([2], reward=np.zeros((1, 5)))
    np.testing.assert_array_equal(a.dirty_rows_since(1), [2])        # ... deltas from then on
    d = h.to_finite_mdp()
    d.reward = shared * 0.5                                          # assigning a table leaves the family too
    assert d.tables_version[0] != c.tables_version[0]


def test_model_cache_catches_tables_edited_behind_an_unchanged_version(monkeypatch):
    """an MDP holds read-only VIEWS, so the caller's own reference can still change the tables under an
    unchanged tables_version.  The cache's sampled guard (4th hit, every 64th; every hit under MP_VERIFY_TABLE_VERSIONS)
    re-hashes, warns, serves the model of the CURRENT tables and stops trusting that token."""
    from rl_agents_amd import device_model
    from rl_agents_amd.envs.finite_mdp import DeterministicMDP
    from rl_agents_amd.envs import generators

    class FakeModel(object):
        def close(self):
            pass

    cfg = generators.highway_shaped(3, 4, 10, seed=1)
    t, r = np.array(cfg["transition"]), np.array(cfg["reward"])
    mdp = DeterministicMDP(t, r, terminal=cfg["terminal"])
    cache = device_model.ModelCache(ctx=object())
    monkeypatch.setattr(cache, "_upload", lambda spec: FakeModel())
    m1 = cache.get(device_model.spec_from_mdp(mdp))
    assert cache.get(device_model.spec_from_mdp(mdp)) is m1
    r[0, 0] += 1.0                                    # behind the MDP's back: same version, other tables
    assert mdp.reward[0, 0] == r[0, 0]
    monkeypatch.setenv("MP_VERIFY_TABLE_VERSIONS", "1")
    with pytest.warns(RuntimeWarning, match="tables_version"):
        m2 = cache.get(device_model.spec_from_mdp(mdp))
    assert m2 is not m1 and cache.uploads == 2 and cache.version_violations == 1
    monkeypatch.delenv("MP_VERIFY_TABLE_VERSIONS")
    r[0, 0] += 1.0                                    # the token is no longer trusted: every lookup hashes
    m3 = cache.get(device_model.spec_from_mdp(mdp))
    assert m3 is not m2 and cache.uploads == 3
    # without the env knob the guard still fires, on the 4th hit of a version
    mdp2 = DeterministicMDP(np.array(t), np.array(cfg["reward"]), terminal=cfg["terminal"])
    r2 = mdp2.reward.base
    cache2 = device_model.ModelCache(ctx=object())
    monkeypatch.setattr(cache2, "_upload", lambda spec: FakeModel())
    first = cache2.get(device_model.spec_from_mdp(mdp2))
    r2[1, 1] += 1.0
    got = [cache2.get(device_model.spec_from_mdp(mdp2)) for _ in range(3)]
    assert all(g is first for g in got)               # (the window the sampling leaves open)
    with pytest.warns(RuntimeWarning):
        assert cache2.get(device_model.spec_from_mdp(mdp2)) is not first


def test_restrict_and_renormalise_equals_the_per_state_reference_loop():
    """trainer.per_episode_evaluation.restrict_and_renormalise (every row of a batch at once) == agent_policy_available
    (mcts_with_prior.py:56-62) state by state -- np.sum over the listed probabilities, including numpy's 8-accumulator form when
    all of 8 actions are listed -- bit for bit."""
    from rl_agents_amd.trainer.per_episode_evaluation import restrict_and_renormalise
    from tests.helpers import restricted_agent_policy_lists
    g = np.random.Generator(np.random.PCG64(0))
    for a in (2, 3, 5, 7, 8):
        t = g.random((6, 40, a))
        t /= t.sum(axis=-1, keepdims=True)
        av = g.random((40, a)) < 0.6
        av[np.arange(40), g.integers(0, a, 40)] = True
        av[:5] = True                                   # (rows with everything listed)
        out = restrict_and_renormalise(t, av)
        for n in range(6):
            ref = restricted_agent_policy_lists(t[n], av)
            for st in range(40):
                want = np.zeros(a)
                want[ref["actions"][st]] = ref["p"][st]
                assert np.array_equal(out[n, st], want), (a, n, st)


def test_opd_leaf_load_wait_counts_match_the_generated_code():
    """the hand-counted `s_waitcnt vmcnt(N)` before opd.hip's scalar leaf-record loads, checked against the
    disassembly of the object the library was linked from (tools/check_isa.py; rl_agents_amd.build refuses a library that
    fails it)."""
    from rl_agents_amd import build
    obj = os.path.join(build.LIB_DIR, "opd.o")
    if not os.path.exists(obj):
        pytest.skip("opd.o not kept (library built elsewhere)")
    assert build.check_generated_code() >= 12


def test_restated_sincos_equals_host_libm():
    """csrc/libm_sincos.hpp on the HOST: the form mp_libm_sincos_variant picks reproduces this process's libm sin / cos (what
    math.sin / math.cos -- gymnasium's CartPole -- call) on two million angles of the restated range; a child process whose
    libm is steered to its other variant (GLIBC_TUNABLES: no FMA) must pick the other form and match it as well."""
    import math
    import subprocess
    import sys
    from rl_agents_amd import native
    variant = native.libm_sincos_variant()
    assert variant in (1, 2), "neither restated form reproduces this host's libm"
    g = np.random.Generator(np.random.PCG64(3))
    x = np.concatenate([g.uniform(-0.855468, 0.855468, 1_000_000), g.uniform(-0.25, 0.25, 1_000_000),
                        g.uniform(-1e-7, 1e-7, 1000), np.arange(-110, 110) / 128.0, [0.0, -0.0, 0.126, -0.126, 2.0 ** -26, 2.0 ** -27]])
    s, c = native.libm_sincos(x, variant)
    s0, c0 = native.libm_sincos(x, 0)
    assert np.array_equal(s, s0) and np.array_equal(c, c0)
    assert [math.sin(v) for v in x[:20000]] == s[:20000].tolist() and [math.cos(v) for v in x[:20000]] == c[:20000].tolist()
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from rl_agents_amd import native; v = native.libm_sincos_variant();"
            "x = np.random.Generator(np.random.PCG64(4)).uniform(-0.855468, 0.855468, 400000);"
            "s, c = native.libm_sincos(x, v); s0, c0 = native.libm_sincos(x, 0);"
            "print(v, int(np.array_equal(s, s0) and np.array_equal(c, c0)))") % REPO
    env = dict(os.environ, GLIBC_TUNABLES="glibc.cpu.hwcaps=-FMA,-AVX2_Usable,-FMA4", MI355PLAN_NO_TORCH="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-500:]
    other, ok = (int(v) for v in out.stdout.decode().split()[-2:])
    assert ok == 1 and other in (1, 2)
    if variant == 1:
        assert other == 2, "with FMA masked off glibc must select (and the probe find) the uncontracted form"


def test_product_never_imports_the_oracle():
    """The oracle is the checker, never the thing shipped: no module of the package (nor its C / HIP sources) imports,
    includes or loads anything under oracle/ -- only tests/, tools/, __graft_entry__.smoke() and the benchmark may."""
    import re
    pkg = os.path.join(REPO, "rl_agents_amd")
    bad = []
    for root, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".cpp")):
                continue
            text = open(os.path.join(root, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle\b|#include\s+[\"<][^\">]*oracle|planning_oracle|liboracle", text, re.M):
                bad.append(os.path.relpath(os.path.join(root, f), REPO))
    assert not bad, bad


def test_scalar_generator_step_restated_equals_numpy():
    """csrc/pcg64.hpp Pcg64U (round 6: the wave-uniform generator of uct_lone_kernel in scalar registers) steps the 128-bit LCG with
    64-bit pieces only: low product, high product by 32-bit limbs (mulhi), the add's carry as the majority bit of the top bits, and
    XSL-RR as two 64-bit shifts.  Restated here with Python integers masked to 64 bits and checked against numpy's PCG64 stream
    (the device result itself is compared with numpy by every GPU parity test that leaves a generator record)."""
    m64 = (1 << 64) - 1

    def mulhi(a, b):
        a0, a1, b0, b1 = a & 0xffffffff, a >> 32, b & 0xffffffff, b >> 32
        p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
        mid = (p00 >> 32) + (p01 & 0xffffffff) + (p10 & 0xffffffff)
        return (p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32)) & m64

    def step(s_hi, s_lo, inc_hi, inc_lo):
        m_lo, m_hi = 0x4385DF649FCCF645, 0x2360ED051FC65DA4
        lo = (s_lo * m_lo) & m64
        hi = (mulhi(s_lo, m_lo) + s_lo * m_hi + s_hi * m_lo) & m64
        lo2 = (lo + inc_lo) & m64
        carry = ((lo & inc_lo) | ((lo | inc_lo) & (~lo2 & m64))) >> 63
        return (hi + inc_hi + carry) & m64, lo2

    def output(s_hi, s_lo):
        x, rot = s_hi ^ s_lo, s_hi >> 58
        return ((x >> rot) | (x << ((64 - rot) & 63))) & m64

    for seed in (0, 1, 12345, 2 ** 63 + 9):
        bg = np.random.PCG64(seed)
        st = bg.state["state"]
        s_hi, s_lo = st["state"] >> 64, st["state"] & m64
        inc_hi, inc_lo = st["inc"] >> 64, st["inc"] & m64
        want = bg.random_raw(300)
        for k in range(300):
            s_hi, s_lo = step(s_hi, s_lo, inc_hi, inc_lo)
            assert output(s_hi, s_lo) == int(want[k]), (seed, k)
        assert (s_hi << 64) | s_lo == bg.state["state"]["state"]
    # the carry formula on the corner cases of a 64-bit add
    for a, b in ((m64, 1), (m64, m64), (1 << 63, 1 << 63), (0, 0), ((1 << 63) - 1, 1), (m64 - 5, 5), (m64 - 5, 6)):
        lo2 = (a + b) & m64
        assert ((a & b) | ((a | b) & (~lo2 & m64))) >> 63 == (a + b) >> 64
