"""The SHARED-model form of the row kernel (uct_row_kernel<AT, true>, round 6): four roots per wavefront, a DPP row of sixteen lanes
each, one copy of the model's transitions and the workgroup's trees in LDS, rewards from the 16-byte records in L2 added in
order one level / one round later -- the default between one root per CU and sixteen (SURVEY 8(d)'s 4096 roots).  Same plans,
trees, env-step counts and generator states as the oracle.  Reference: MCTS.run / evaluate (mcts.py:132-184), Node.random_argmax
(abstract.py:296-311)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _rows_not_the_multi_wave_lone_form(monkeypatch):
    """These are the row kernel's tests: since later in round 6 batches of up to 8 roots per CU take uct_lone_kernel<.., MW> by
    default (tests/test_gpu_uct_lone.py covers it); with that form off the row kernel is the default from one root per CU up."""
    monkeypatch.setenv("MP_UCT_LONE_WAVES", "0")


def _cmp(ctx, cfg, n_roots, episodes, horizon, gamma, temperature, prior, rollout, seed=0, max_steps=0, steps0=None,
         done_rule="source", expect="uct_row_shared"):
    from oracle import oracle
    from rl_agents_amd import native
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term, done_rule=done_rule, max_steps=max_steps)
    s0 = np.random.Generator(np.random.PCG64(seed)).integers(0, r.shape[0], size=n_roots).astype(np.int32)
    rng = native.seed_sequence_states((), 1000 * seed, n_roots)
    rng_ref = rng.copy()
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
    model.close()
    return out


@pytest.mark.parametrize("n_roots", [257, 1000, 1024, 2049, 4096])
def test_rows_headline_geometry_default(ctx, monkeypatch, n_roots):
    """Headline table (S = 10 000, |A| = 5), budget 1000 as 33 x 30: the default kernel from 2049 to 4096 roots, and from 257 with
    the multi-wavefront lone form turned off (one, two or four waves per workgroup by batch size)."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    if n_roots <= 2048:
        monkeypatch.setenv("MP_UCT_LONE_WAVES", "0")     # (round 6, later: up to 8 roots per CU plan on uct_lone_kernel<.., MW>)
    _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 2 / (1 - 0.8), p, p, seed=n_roots)


@pytest.mark.parametrize("n_roots,waves,rpw", [(1, 1, 4), (5, 4, 2), (64, 2, 4), (777, 8, 2), (6000, 4, 4), (6001, 8, 2), (20000, 4, 4)])
def test_rows_forced_any_batch(ctx, monkeypatch, n_roots, waves, rpw):
    """Forced (MP_UCT_ROWS=1) below and beyond its default range: ragged last workgroups, several rounds of workgroups, two or four
    roots per wavefront (MP_UCT_ROW_ROOTS) in one to eight planning wavefronts per workgroup."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_ROWS", "1")
    monkeypatch.setenv("MP_UCT_LONE", "0")
    monkeypatch.setenv("MP_UCT_ROW_WAVES", str(waves))
    monkeypatch.setenv("MP_UCT_ROW_ROOTS", str(rpw))
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, n_roots, 12, 30, 0.8, 10.0, p, p, seed=n_roots + 1)


def test_rows_range_and_switch(ctx, monkeypatch):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 256, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_lone")
    _cmp(ctx, cfg, 4097, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_quad")
    monkeypatch.setenv("MP_UCT_ROWS", "0")
    _cmp(ctx, cfg, 1000, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_quad")


@pytest.mark.parametrize("n_actions", [2, 3, 4, 6, 7, 8])
def test_rows_every_action_count(ctx, n_actions):
    """|A| = 2 .. 8, many distinct rewards (no reward dictionary is involved: rewards come from the records), skewed policies."""
    g = np.random.Generator(np.random.PCG64(n_actions))
    s = 300
    cfg = dict(transition=g.integers(0, s, size=(s, n_actions)), reward=g.random((s, n_actions)), terminal=g.random(s) < 0.05)
    pr = g.random(n_actions) + 0.1
    pr /= pr.sum()
    ro = g.random(n_actions) + 0.1
    ro /= ro.sum()
    _cmp(ctx, cfg, 333, 40, 12, 0.9, 5.0, pr, ro, seed=n_actions)


@pytest.mark.parametrize("horizon", [1, 2, 15, 16, 17, 33, 120, 255])
def test_rows_horizons(ctx, monkeypatch, horizon):
    """Rollouts of every length around the rounds of sixteen draws, up to the 255 steps the jump table covers."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_ROWS", "1")     # (a model this small plans 300 roots on uct_lone_kernel by default since round 6)
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 300, 20, horizon, 0.95, 10.0, p, p, seed=horizon)


def test_rows_truncation_terminal_conventions_and_zero_probabilities(ctx, monkeypatch):
    """TimeLimit truncation with per-root step counts, both terminal conventions, rollout policies with zero-probability actions."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_ROWS", "1")     # (as above: the small model's default is uct_lone_kernel)
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    n = 500
    steps0 = (np.arange(n) % 9).astype(np.int32)
    prior = np.array([0.1, 0.5, 0.1, 0.2, 0.1])
    for done_rule in ("source", "next"):
        for rollout in (np.array([0.0, 0.25, 0.5, 0.25, 0.0]), np.array([0.0, 0.0, 1.0, 0.0, 0.0]), np.ones(5) / 5):
            _cmp(ctx, cfg, n, 30, 8, 0.8, 10.0, prior, rollout, seed=4, max_steps=10, steps0=steps0, done_rule=done_rule)


def test_rows_whole_trees_equal_the_one_lane_kernel(ctx, monkeypatch):
    """Every node of every tree (value, count, first child): the export of the row kernel against the one-lane-per-root kernel."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    n = 600
    s0 = np.random.Generator(np.random.PCG64(8)).integers(0, 10000, size=n).astype(np.int32)
    p = np.ones(5) / 5

    def trees():
        rng = native.seed_sequence_states((), 77, n)
        ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, p, p, rng, max_plan_len=4)
        return ctx.last_kernel_variant(), [ctx.uct_tree(i, 1 + 33 * 5) for i in (0, 1, 63, 64, 255, 599)], rng
    v1, t1, r1 = trees()
    monkeypatch.setenv("MP_UCT_ROWS", "0")
    monkeypatch.setenv("MP_UCT_QUAD", "0")
    v2, t2, r2 = trees()
    assert v1 == "uct_row_shared" and v2 == "uct_global", (v1, v2)
    np.testing.assert_array_equal(r1, r2)
    for a, b in zip(t1, t2):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    model.close()
