"""uct_lone_kernel (round 5): ONE ROOT PER WORKGROUP -- the plan of a single agent's act() and of every batch of at most one root
per CU.  Model, per-call tables and the tree in LDS; a level's children scored one per lane; a rollout's actions drawn by the
lanes in parallel with PCG64 jump-ahead, the walk one dependent LDS read per step, rewards one step behind.  Same plans,
statistics, env-step counts, generator states and exported trees as the oracle.  Reference: MCTS.run / evaluate
(mcts.py:132-184), Node.random_argmax (abstract.py:296-311)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def _cmp(ctx, cfg, n_roots, episodes, horizon, gamma, temperature, prior, rollout, seed=0, max_steps=0, steps0=None,
         done_rule="source", expect="uct_lone", trees=(0,)):
    from oracle import oracle
    from rl_agents_amd import native
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term, done_rule=done_rule, max_steps=max_steps)
    s0 = np.random.Generator(np.random.PCG64(seed)).integers(0, r.shape[0], size=n_roots).astype(np.int32)
    rng = native.seed_sequence_states((), 1000 * seed, n_roots)
    rng_ref = rng.copy()
    ctx.uct_reset_tree()
    out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, prior, rollout, rng, root_steps=steps0,
                       max_plan_len=max(horizon, 1))
    assert ctx.last_kernel_variant() == expect, ctx.last_kernel_variant()
    ref = oracle.uct_plan_batch(t, r, term, s0, episodes, horizon, gamma, temperature, prior, rollout, rng_ref, steps0=steps0,
                                max_steps=max_steps, done_rule=done_rule, max_plan_len=max(horizon, 1), n_threads=8)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["plan_len"], ref["plan_len"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"])
    assert np.array_equal(out["root_child_value"], ref["root_child_value"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    for root in trees:                      # the whole tree (written from LDS to the exported layout), node for node
        if root >= n_roots:
            continue
        tree = ctx.uct_tree(root, 1 + episodes * r.shape[1])
        one = oracle.uct_plan(t, r, term, int(s0[root]), episodes, horizon, gamma, temperature, prior, rollout,
                              native.seed_sequence_states((), 1000 * seed, n_roots)[root], max_steps=max_steps,
                              steps0=0 if steps0 is None else int(steps0[root]), done_rule=done_rule,
                              max_plan_len=max(horizon, 1))["tree"]
        for k in ("parent", "action", "count", "value", "first_child"):
            np.testing.assert_array_equal(tree[k], one[k], err_msg="tree[{}] of root {}".format(k, root))
    model.close()
    return out


@pytest.mark.parametrize("n_roots", [1, 3, 15, 64, 256])
def test_lone_headline_geometry(ctx, n_roots):
    """Headline table (S = 10 000, |A| = 5: 150 of the CU's 160 KB of LDS), budget 1000 as 33 x 30."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 2 / (1 - 0.8), p, p, seed=n_roots, trees=(0, n_roots - 1))


def test_lone_is_for_at_most_one_root_per_cu_and_by_request(ctx, monkeypatch):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 257, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_lone_mw")   # (round 6: two planning wavefronts per workgroup)
    _cmp(ctx, cfg, 2049, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_row_shared", trees=())   # (beyond 8 roots per CU: the row kernel)
    monkeypatch.setenv("MP_UCT_LONE", "0")
    _cmp(ctx, cfg, 8, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_global")
    monkeypatch.setenv("MP_UCT_LONE", "1")
    _cmp(ctx, cfg, 700, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_lone")     # (three rounds of workgroups)


@pytest.mark.parametrize("n_actions", [2, 3, 4, 6, 7, 8])
def test_lone_every_action_count(ctx, n_actions):
    g = np.random.Generator(np.random.PCG64(n_actions))
    s = 300
    cfg = dict(transition=g.integers(0, s, size=(s, n_actions)), reward=g.choice(np.linspace(0, 1, 17), size=(s, n_actions)),
               terminal=g.random(s) < 0.05)
    pr = g.random(n_actions) + 0.1
    pr /= pr.sum()
    ro = g.random(n_actions) + 0.1
    ro /= ro.sum()
    _cmp(ctx, cfg, 100, 40, 12, 0.9, 5.0, pr, ro, seed=n_actions, trees=(0, 99))


@pytest.mark.parametrize("horizon", [1, 2, 4, 5, 39, 62, 63])
def test_lone_horizons(ctx, horizon):
    """Rollouts of every length up to the 63 steps one wavefront's lanes draw at once."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 33, 25, horizon, 0.95, 10.0, p, p, seed=horizon)


def test_lone_longer_horizons_and_many_episodes_fall_back(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 33, 10, 64, 0.95, 10.0, p, p, seed=1, expect="uct_row_shared", trees=())
    _cmp(ctx, cfg, 5, 10, 64, 0.95, 10.0, p, p, seed=1, expect="uct_global", trees=())
    big = generators.highway_shaped(10, 10, 100, seed=0)         # a 600-episode tree does not fit LDS beside this model
    _cmp(ctx, big, 2, 600, 5, 0.8, 10.0, p, p, seed=3, expect="uct_global", trees=())


def test_lone_truncation_terminal_conventions_and_zero_probabilities(ctx):
    """TimeLimit truncation with per-root step counts (also roots already past the limit), both terminal conventions, a rollout
    policy with zero-probability actions at either end, a preference-like prior."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    n = 200
    steps0 = (np.arange(n) % 13).astype(np.int32)
    prior = np.array([0.1, 0.5, 0.1, 0.2, 0.1])
    for done_rule in ("source", "next"):
        for rollout in (np.array([0.0, 0.25, 0.5, 0.25, 0.0]), np.array([0.0, 0.0, 1.0, 0.0, 0.0]), np.ones(5) / 5):
            _cmp(ctx, cfg, n, 30, 8, 0.8, 10.0, prior, rollout, seed=4, max_steps=10, steps0=steps0, done_rule=done_rule,
                 trees=(0, 12))


def test_lone_hands_kept_subtrees_to_the_other_kernels(ctx):
    """step_strategy 'subtree': the first plan (fresh tree) runs on the lone kernel, its tree -- written to the exported layout --
    is re-rooted and CONTINUED by the one-lane kernel; three plans equal the oracle's."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    p = np.ones(5) / 5
    s, rng = 5, native.seed_sequence_states((), 11, 1)
    ref_rng = rng.copy()
    tree = None
    ctx.uct_reset_tree()
    for step in range(3):
        out = ctx.uct_plan(model, [s], 25, 12, 0.8, 10.0, p, p, rng, max_plan_len=12)
        assert ctx.last_kernel_variant() == ("uct_lone" if step == 0 else "uct_global")
        ref = oracle.uct_plan(t, r, term, s, 25, 12, 0.8, 10.0, p, p, ref_rng[0], max_plan_len=12, init_tree=tree)
        ref_rng[0] = ref["rng_after"]
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], ref["plan"])
        np.testing.assert_array_equal(rng, ref_rng)
        a = int(ref["plan"][0])
        ctx.uct_step_tree([a])
        tree = oracle.uct_reroot(ref["tree"], a, 5)
        s = int(t[s, a])
    ctx.uct_reset_tree()
    model.close()


@pytest.mark.parametrize("done_rule", ["source", "next"])
@pytest.mark.parametrize("length", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13])
def test_lone_walk_every_group_phase(ctx, length, done_rule):
    """The walk runs in groups of four speculative steps with one test per group, then step by step (round 6): a corridor whose
    terminal state is `length` steps away makes every rollout end at a chosen phase of a group (every action moves on; the
    states past the terminal one exist, so what a speculative step reads is a valid record) -- under both terminal rules, with
    horizons that leave 0 .. 3 steps for the step-by-step tail, and with the env's step limit cutting the rollout instead."""
    n_act, s = 3, 40
    t = np.minimum(np.arange(s)[:, None] + 1, s - 1) * np.ones((1, n_act), dtype=np.int64)
    g = np.random.Generator(np.random.PCG64(7 * length))
    cfg = dict(transition=t, reward=g.choice(np.linspace(-1, 1, 9), size=(s, n_act)), terminal=np.arange(s) == length + 2)
    p = np.ones(n_act) / n_act
    for horizon in (length + 4, length + 5, length + 6, 63):
        _cmp(ctx, cfg, 5, 9, horizon, 0.9, 3.0, p, p, seed=length, done_rule=done_rule)
    cfg["terminal"] = np.zeros(s, dtype=bool)                      # no terminal state: the step limit ends the rollouts
    for max_steps in (length, length + 1, length + 3):
        _cmp(ctx, cfg, 5, 9, 30, 0.9, 3.0, p, p, seed=length, max_steps=max_steps, done_rule=done_rule,
             steps0=np.array([0, 1, 2, 0, 1], dtype=np.int32))


def test_lone_serves_as_many_roots_as_its_workgroups_fit_a_cu(ctx):
    """Round 6: a model that leaves room in a CU's LDS plans as many roots on uct_lone_kernel as its workgroups fit the CU together
    (at most two planning wavefronts per SIMD: 8 x the CUs); one root more, or a model that fills the LDS, takes the row kernel."""
    from rl_agents_amd.envs import generators
    p = np.ones(5) / 5
    small = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)       # S = 1 000: ~20 KB of LDS per workgroup
    _cmp(ctx, small, 300, 20, 17, 0.95, 10.0, p, p, seed=17, trees=(0, 299))
    _cmp(ctx, small, 1500, 12, 9, 0.9, 10.0, p, p, seed=18, trees=(0, 777, 1499))
    tiny = generators.highway_shaped(3, 4, 10, seed=3)                               # S = 120
    _cmp(ctx, tiny, 2048, 9, 8, 0.8, 10.0, p, p, seed=19, trees=(0, 2047))
    _cmp(ctx, tiny, 2049, 9, 8, 0.8, 10.0, p, p, seed=19, expect="uct_row_shared", trees=())


@pytest.mark.parametrize("n_roots", [257, 700, 1024, 1025, 2048])
def test_lone_multi_wave_headline_geometry(ctx, n_roots):
    """uct_lone_kernel<.., MW> (round 6): 2 / 4 / 8 planning wavefronts per workgroup around ONE copy of the headline table's
    transitions (S = 10 000: 100 of the CU's 160 KB), each root's tree and jump table behind them, an episode's rewards from the
    16-byte records in L2 -- the default for 257 .. 2048 roots.  Plans, statistics, generator records and whole trees vs the oracle."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 2 / (1 - 0.8), p, p, seed=n_roots, expect="uct_lone_mw", trees=(0, 1, n_roots // 2, n_roots - 1))


@pytest.mark.parametrize("waves", [2, 4, 8])
def test_lone_multi_wave_forced_ragged_and_rules(ctx, monkeypatch, waves):
    """Forced (MP_UCT_LONE_WAVES) on batches that leave the last workgroup ragged or need several rounds of workgroups, both
    terminal rules, a step limit with per-root step counts, non-uniform policies with zero-probability actions, |A| = 3 and 8, and
    a reward table with more than 256 distinct values (no compact reward index exists: the records are the only source)."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_LONE_WAVES", str(waves))
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.02, seed=11)
    prior = np.array([0.1, 0.5, 0.1, 0.2, 0.1])
    for n_roots in (1, waves + 1, 3000):
        for done_rule in ("source", "next"):
            _cmp(ctx, cfg, n_roots, 12, 9, 0.9, 5.0, prior, np.array([0.0, 0.25, 0.5, 0.25, 0.0]), seed=waves, done_rule=done_rule,
                 expect="uct_lone_mw", trees=(0, n_roots - 1))
    steps0 = (np.arange(77) % 6).astype(np.int32)
    _cmp(ctx, cfg, 77, 20, 14, 0.8, 10.0, prior, np.ones(5) / 5, seed=5, max_steps=9, steps0=steps0, expect="uct_lone_mw", trees=(0, 76))
    g = np.random.Generator(np.random.PCG64(100 + waves))
    for n_act in (3, 8):
        s = 400
        many = dict(transition=g.integers(0, s, size=(s, n_act)), reward=g.random((s, n_act)), terminal=g.random(s) < 0.05)
        p = np.ones(n_act) / n_act
        _cmp(ctx, many, 41, 15, 11, 0.95, 4.0, p, p, seed=n_act, expect="uct_lone_mw", trees=(0, 40))


def test_lone_one_wave_for_models_without_a_compact_reward_index(ctx):
    """A table with more than 256 distinct rewards has no one-byte reward index, so the plain one-root-per-workgroup form (model =
    transitions + reward indices in LDS) does not apply; the MW form with ONE planning wavefront does (transitions in LDS, rewards
    from the records): a single agent's act() on such a model, and every small batch of it."""
    g = np.random.Generator(np.random.PCG64(21))
    s, n_act = 3000, 5
    cfg = dict(transition=g.integers(0, s, size=(s, n_act)), reward=g.random((s, n_act)), terminal=g.random(s) < 0.03)
    p = np.ones(n_act) / n_act
    for n_roots in (1, 7, 256):
        _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 10.0, p, p, seed=n_roots, expect="uct_lone_mw", trees=(0, n_roots - 1))
    _cmp(ctx, cfg, 300, 12, 20, 0.9, 10.0, p, p, seed=3, done_rule="next", expect="uct_lone_mw", trees=(0, 299))


def test_lone_one_wave_for_a_model_whose_transitions_alone_fit_the_lds(ctx):
    """S = 15 000, |A| = 5: 150 KB of transitions + 75 KB of reward indices is beyond the CU's LDS for the plain form; the
    transitions alone, one tree and the tables are 159 of its 160 KB -- the MW form with one planning wavefront, budget 1000."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 15, 100, seed=2)
    assert cfg["reward"].shape == (15000, 5)
    p = np.ones(5) / 5
    for n_roots in (1, 40):
        _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 10.0, p, p, seed=4 + n_roots, expect="uct_lone_mw", trees=(0, n_roots - 1))
