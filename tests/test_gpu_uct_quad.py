"""The four-lanes-per-root UCT kernel (uct_kernel<..., QD>, round 5: small batches of an LDS-resident model -- the lanes of a
root draw the rollout's actions with PCG64 jump-ahead, the root's first lane walks the model through them): same plans,
trees, env-step counts and generator states as the oracle and as the one-lane-per-root kernels.  Reference: MCTS.evaluate
(mcts.py:160-177), Node.random_argmax (abstract.py:296-311)."""
import os

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
def _no_lone_kernel(monkeypatch):
    """(batches of at most one root per CU take uct_lone_kernel by default since it exists -- tests/test_gpu_uct_lone.py; this
    file is about the kernel that served them before and still serves 257 .. 16 384 roots)"""
    monkeypatch.setenv("MP_UCT_LONE", "0")
    monkeypatch.setenv("MP_UCT_ROWS", "0")      # (257 .. 4096 roots take the shared-model row kernel since round 6: test_gpu_uct_rows.py)


def _rng_states(n, base=0):
    from rl_agents_amd import native
    return native.seed_sequence_states((), base, n)


def _cmp(ctx, cfg, n_roots, episodes, horizon, gamma, temperature, prior, rollout, seed=0, max_steps=0, steps0=None,
         done_rule="source", expect="uct_quad"):
    from oracle import oracle
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term, done_rule=done_rule, max_steps=max_steps)
    s0 = np.random.Generator(np.random.PCG64(seed)).integers(0, r.shape[0], size=n_roots).astype(np.int32)
    rng = _rng_states(n_roots, base=1000 * seed)
    rng_ref = rng.copy()
    out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, prior, rollout, rng, root_steps=steps0,
                       max_plan_len=max(horizon, 1))
    if n_roots < 16 and expect == "uct_quad" and not os.environ.get("MP_UCT_QUAD"):
        expect = "uct_global"          # (fewer than one wave's sixteen roots: the one-lane-per-root kernel is as fast)
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


@pytest.mark.parametrize("n_roots", [1, 3, 16, 17, 1000, 4096, 16384])
def test_quad_headline_geometry(ctx, n_roots, monkeypatch):
    """Headline table (S = 10 000, |A| = 5), budget 1000 as 33 x 30, the batch sizes SURVEY 8(d) names and ragged ones
    (fewer than sixteen roots: forced, the default there is the one-lane kernel)."""
    from rl_agents_amd.envs import generators
    if n_roots < 16:
        monkeypatch.setenv("MP_UCT_QUAD", "1")
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, n_roots, 33, 30, 0.8, 2 / (1 - 0.8), p, p, seed=n_roots)


def test_quad_is_off_beyond_one_wave_per_simd_and_by_request(ctx, monkeypatch):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 16384 + 16, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_global")
    monkeypatch.setenv("MP_UCT_QUAD", "0")
    _cmp(ctx, cfg, 64, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_global")
    monkeypatch.setenv("MP_UCT_QUAD", "1")
    _cmp(ctx, cfg, 40000, 6, 10, 0.8, 10.0, p, p, seed=2, expect="uct_quad")


@pytest.mark.parametrize("n_actions", [2, 3, 4, 6, 7, 8])
def test_quad_every_action_count(ctx, n_actions):
    g = np.random.Generator(np.random.PCG64(n_actions))
    s = 300
    cfg = dict(transition=g.integers(0, s, size=(s, n_actions)), reward=g.choice(np.linspace(0, 1, 17), size=(s, n_actions)),
               terminal=g.random(s) < 0.05)
    pr = g.random(n_actions) + 0.1
    pr /= pr.sum()
    ro = g.random(n_actions) + 0.1
    ro /= ro.sum()
    _cmp(ctx, cfg, 333, 40, 12, 0.9, 5.0, pr, ro, seed=n_actions)


@pytest.mark.parametrize("horizon", [1, 2, 4, 5, 39, 40, 41, 120, 255])
def test_quad_horizons(ctx, horizon):
    """Rollouts of every length (rounds of four draws): up to the 255 steps the jump table in LDS covers; longer horizons
    take the one-lane-per-root kernels."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 77, 25, horizon, 0.95, 10.0, p, p, seed=horizon)


def test_quad_long_horizon_falls_back(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 50, collision_rate=0.01, seed=9)
    p = np.ones(5) / 5
    _cmp(ctx, cfg, 77, 10, 256, 0.95, 10.0, p, p, seed=1, expect="uct_global_spill")


def test_quad_truncation_terminal_conventions_and_zero_probabilities(ctx):
    """TimeLimit truncation with per-root step counts, both terminal conventions, a rollout policy with zero-probability
    actions at either end (thresholds that can never be reached), a preference-like prior."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    n = 500
    steps0 = (np.arange(n) % 9).astype(np.int32)
    prior = np.array([0.1, 0.5, 0.1, 0.2, 0.1])
    for done_rule in ("source", "next"):
        for rollout in (np.array([0.0, 0.25, 0.5, 0.25, 0.0]), np.array([0.0, 0.0, 1.0, 0.0, 0.0]), np.ones(5) / 5):
            _cmp(ctx, cfg, n, 30, 8, 0.8, 10.0, prior, rollout, seed=4, max_steps=10, steps0=steps0, done_rule=done_rule)


def test_quad_kept_subtrees(ctx):
    """step_strategy 'subtree' on the four-lanes kernel: three plans on re-rooted trees equal the oracle's."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    p = np.ones(5) / 5
    s, rng = 5, _rng_states(1, base=11)
    ref_rng = rng.copy()
    tree = None
    os.environ["MP_UCT_QUAD"] = "1"       # (a single root takes the one-lane kernel by default)
    ctx.uct_reset_tree()
    for step in range(3):
        out = ctx.uct_plan(model, [s], 25, 12, 0.8, 10.0, p, p, rng, max_plan_len=12)
        assert ctx.last_kernel_variant() == "uct_quad"
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
    os.environ.pop("MP_UCT_QUAD", None)
    model.close()
