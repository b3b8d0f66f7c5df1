"""Batched HIP planners vs the CPU oracle on seeded synthetic tables (many roots per launch), bit for bit,
plus size-independent properties at the BASELINE.json sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def _rng_states(n, base=0):
    from rl_agents_amd import native
    return np.stack([native.rng_state_from_generator(
        np.random.Generator(np.random.PCG64(np.random.SeedSequence(base + i)))) for i in range(n)])


def _cmp_uct(ctx, cfg, n_roots, episodes, horizon, gamma, temperature, prior, rollout, seed=0, max_steps=0,
             steps0=None, done_rule="source"):
    from oracle import oracle
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term, done_rule=done_rule, max_steps=max_steps)
    s0 = np.random.Generator(np.random.PCG64(seed)).integers(0, r.shape[0], size=n_roots).astype(np.int32)
    rng = _rng_states(n_roots, base=1000 * seed)
    rng_ref = rng.copy()
    out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, prior, rollout, rng, root_steps=steps0,
                       max_plan_len=horizon)
    ref = oracle.uct_plan_batch(t, r, term, s0, episodes, horizon, gamma, temperature, prior, rollout, rng_ref,
                                steps0=steps0, max_steps=max_steps, done_rule=done_rule, max_plan_len=horizon,
                                n_threads=8)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["plan_len"], ref["plan_len"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"])
    assert np.array_equal(out["root_child_value"], ref["root_child_value"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()
    return out


@pytest.mark.parametrize("variant", ["global", "lds", "ldsr"])
def test_uct_batch_highway_headline_shape(ctx, variant, monkeypatch):
    """Headline configuration (highway-shaped S=10 000, A=5, 33 episodes x horizon 30), 1536 ragged roots,
    with the model gathered from HBM/L2 records (default at this size), with the transition table staged in LDS, and with
    the whole model resident in LDS (round 4: the default from 65 536 roots upwards)."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_MODEL", variant)
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    p = np.ones(5) / 5
    out = _cmp_uct(ctx, cfg, 1536 + 17, 33, 30, 0.8, 2 / (1 - 0.8), p, p, seed=1)
    assert out["env_steps"].max() <= 33 * 30
    assert ctx.last_kernel_variant() == "uct_" + variant


def _coarse(cfg, levels=16):
    """The same model with rewards on a grid of `levels` + 1 values: at most 256 distinct rewards, what the LDS-resident
    variant needs (its reward table is indexed by a byte)."""
    return dict(cfg, reward=np.round(np.asarray(cfg["reward"]) * levels) / levels)


@pytest.mark.parametrize("n_actions", [2, 3, 4, 5, 6, 7, 8])
def test_uct_lds_resident_model_action_counts(ctx, n_actions, monkeypatch):
    """ENV_TABLE_LDSR for every |A| it is compiled for: preference policies, TimeLimit truncation with steps already
    taken, both terminal conventions, ragged batch sizes around the workgroup size."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_MODEL", "ldsr")
    cfg = _coarse(generators.random_deterministic(257, n_actions, seed=n_actions, terminal_rate=0.05))
    p = np.ones(n_actions) / n_actions
    _cmp_uct(ctx, cfg, 200, 25, 9, 0.9, 7.5, p, p, seed=n_actions)
    assert ctx.last_kernel_variant() == "uct_ldsr"
    pref = np.ones(n_actions) / (n_actions - 1 + 3.0)
    pref[1] *= 3.0
    steps0 = np.random.Generator(np.random.PCG64(5)).integers(0, 6, size=1100).astype(np.int32)
    monkeypatch.setenv("MP_UCT_LDSR_WAVES", "16")          # whole 1024-thread workgroups and a ragged last one
    _cmp_uct(ctx, cfg, 1100, 40, 12, 0.95, 3.0, pref, pref, seed=3, max_steps=10, steps0=steps0)
    _cmp_uct(ctx, cfg, 1100, 40, 12, 0.95, 3.0, pref, p, seed=4, done_rule="next")
    assert ctx.last_kernel_variant() == "uct_ldsr"


def test_uct_lds_resident_model_deep_trees_spill_the_path(ctx, monkeypatch):
    """Paths deeper than the register-held levels (depths 1..5) go through the global spill array: 600 episodes at a
    small exploration constant on a two-action chain drive the tree to depth > 12."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_MODEL", "ldsr")
    cfg = _coarse(generators.random_deterministic(97, 2, seed=5, terminal_rate=0.0), levels=4)
    p = np.ones(2) / 2
    out = _cmp_uct(ctx, cfg, 130, 600, 25, 0.97, 0.05, p, p, seed=6)
    assert ctx.last_kernel_variant() == "uct_ldsr" and out["plan_len"].max() > 8
    cfg3 = _coarse(generators.random_deterministic(64, 3, seed=8, terminal_rate=0.02), levels=8)
    p3 = np.ones(3) / 3
    out = _cmp_uct(ctx, cfg3, 70, 900, 30, 0.99, 0.02, p3, p3, seed=7)
    assert out["plan_len"].max() > 8


@pytest.mark.parametrize("n_actions", [3, 5, 7])
def test_uct_long_horizons_spill_the_path_stack(ctx, n_actions, monkeypatch):
    """The reference has no horizon limit (mcts.py:116-118); a [H + 1][64] path stack stops fitting 64 KB of LDS near
    H = 180 (round 3 refused there).  Beyond, the stack lives in registers (depths 1..5) + a global spill array; also forced
    at a small horizon (MP_UCT_PATH=spill), for compile-time and generic |A|."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(300, n_actions, seed=40 + n_actions, terminal_rate=0.0)
    p = np.ones(n_actions) / n_actions
    _cmp_uct(ctx, cfg, 70, 30, 400, 0.995, 0.01, p, p, seed=2)                # horizon 400: deep, narrow trees
    assert ctx.last_kernel_variant() == "uct_global_spill"
    monkeypatch.setenv("MP_UCT_PATH", "spill")
    _cmp_uct(ctx, cfg, 130, 200, 20, 0.97, 0.05, p, p, seed=3)
    assert ctx.last_kernel_variant() == "uct_global_spill"


def test_uct_lds_resident_model_falls_back(ctx, monkeypatch):
    """More than 256 distinct rewards: no byte-indexed reward table exists and the record-gather kernel runs, also when
    the variant is asked for; a model that does not fit the LDS is refused when forced."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_MODEL", "ldsr")
    cfg = generators.random_deterministic(257, 4, seed=4, terminal_rate=0.05)     # 1028 distinct rewards
    p = np.ones(4) / 4
    _cmp_uct(ctx, cfg, 100, 20, 8, 0.9, 5.0, p, p, seed=1)
    assert ctx.last_kernel_variant() == "uct_global"
    big = _coarse(generators.random_deterministic(20000, 4, seed=4, terminal_rate=0.05))   # 240 KB of tables
    model = ctx.load_table(big["transition"], big["reward"], big["terminal"])
    with pytest.raises(native.NativeError):
        ctx.uct_plan(model, np.zeros(4, np.int32), 5, 5, 0.9, 5.0, p, p, _rng_states(4), max_plan_len=5)
    monkeypatch.delenv("MP_UCT_MODEL")
    ctx.uct_plan(model, np.zeros(4, np.int32), 5, 5, 0.9, 5.0, p, p, _rng_states(4), max_plan_len=5)
    # (round 6: its TRANSITIONS alone -- 160 000 B -- fit the LDS beside this small plan's tree: the multi-wavefront lone form with
    # one planning wavefront takes it, the rewards from the records; tests/test_gpu_uct_lone.py compares that form with the oracle)
    assert ctx.last_kernel_variant() == "uct_lone_mw"
    monkeypatch.setenv("MP_UCT_LONE_WAVES", "0")
    ctx.uct_plan(model, np.zeros(4, np.int32), 5, 5, 0.9, 5.0, p, p, _rng_states(4), max_plan_len=5)
    assert ctx.last_kernel_variant() == "uct_global"
    model.close()


@pytest.mark.parametrize("variant", ["global", "lds"])
@pytest.mark.parametrize("n_actions", [2, 3, 4, 5, 6, 7, 8, 11])
def test_uct_batch_action_counts(ctx, n_actions, variant, monkeypatch):
    """Every compile-time |A| specialisation and the generic-|A| kernel, both model placements."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_UCT_MODEL", variant)
    cfg = generators.random_deterministic(257, n_actions, seed=n_actions, terminal_rate=0.05)
    p = np.ones(n_actions) / n_actions
    _cmp_uct(ctx, cfg, 200, 25, 9, 0.9, 7.5, p, p, seed=n_actions)


def test_uct_batch_preference_policy_truncation_and_next_rule(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(500, 4, seed=21, terminal_rate=0.1)
    pref = np.ones(4) / (4 - 1 + 3.0)
    pref[1] *= 3.0                                     # mcts.py:76-97 preference policy, ratio 3
    steps0 = np.random.Generator(np.random.PCG64(5)).integers(0, 6, size=300).astype(np.int32)
    _cmp_uct(ctx, cfg, 300, 40, 12, 0.95, 3.0, pref, pref, seed=3, max_steps=10, steps0=steps0)
    _cmp_uct(ctx, cfg, 300, 40, 12, 0.95, 3.0, pref, np.ones(4) / 4, seed=4, done_rule="next")


@pytest.mark.parametrize("n_actions,episodes", [(5, 5000), (20, 1200)])
def test_uct_more_episodes_than_the_quotient_tables_hold(ctx, n_actions, episodes):
    """budget 50 000 at horizon 10: 5 000 episodes -- visit counts far beyond the 16 KB of 1/n and prior/n tables in LDS
    (the kernel's own IEEE divisions from there on; before round 3's second half such a plan was refused)."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(150, n_actions, seed=3 + n_actions, terminal_rate=0.02)
    g = np.random.Generator(np.random.PCG64(n_actions))
    prior = g.random(n_actions) + 0.2
    prior /= prior.sum()
    p = np.ones(n_actions) / n_actions
    _cmp_uct(ctx, cfg, 70, episodes, 10, 0.9, 3.0, prior, p, seed=2)


def test_uct_zero_episodes_and_single_root(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(50, 3, seed=2)
    p = np.ones(3) / 3
    out = _cmp_uct(ctx, cfg, 1, 0, 5, 0.8, 10.0, p, p)
    assert out["plan_len"][0] == 0
    _cmp_uct(ctx, cfg, 1, 90, 11, 0.8, 10.0, p, p)


def test_uct_small_plans_survive_the_eviction_of_their_pinned_arrays(ctx):
    """Host-array plans of few roots go through pinned arrays the context keeps per plan shape, their addresses remembered beside
    them (round 6); the context holds a handful of shapes and frees the oldest: a dozen shapes in turn, then the first again --
    every plan (with step counts on every other one) equal to the oracle."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(200, 4, seed=12, terminal_rate=0.03)
    p = np.ones(4) / 4
    for rnd in range(2):
        for horizon in list(range(2, 14)) + [2]:
            n = 1 + horizon % 3
            _cmp_uct(ctx, cfg, n, 6, horizon, 0.9, 3.0, p, p, seed=horizon + rnd, max_steps=9 if horizon % 2 else 0,
                     steps0=(np.arange(n) % 4).astype(np.int32) if horizon % 2 else None)


def test_uct_stream_continues_across_calls(ctx):
    """Two plan() calls on one agent continue one PCG64 stream (tree reset in between), as the reference does."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(100, 5, seed=9)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    p = np.ones(5) / 5
    rng, rng_ref = _rng_states(64), _rng_states(64)
    s0 = np.arange(64, dtype=np.int32)
    for _ in range(3):
        out = ctx.uct_plan(model, s0, 14, 6, 0.8, 10.0, p, p, rng, max_plan_len=6)
        ref = oracle.uct_plan_batch(t, r, term, s0, 14, 6, 0.8, 10.0, p, p, rng_ref, max_plan_len=6)
        rng_ref = ref["rng_after"]
        np.testing.assert_array_equal(out["plans"], ref["plans"])
        np.testing.assert_array_equal(rng, rng_ref)
        s0 = t[s0, np.maximum(out["plans"][:, 0], 0)].astype(np.int32)


def _cmp_opd(ctx, cfg, n_roots, budget, gamma, terminal_reward=0.0, seed=0, done_rule="source"):
    from oracle import oracle
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term, done_rule=done_rule)
    s0 = np.random.Generator(np.random.PCG64(seed)).integers(0, r.shape[0], size=n_roots).astype(np.int32)
    rng = _rng_states(n_roots, base=77)
    rng_ref, rng0 = rng.copy(), rng.copy()
    mpl = budget // r.shape[1] + 2
    out = ctx.opd_plan(model, s0, budget, gamma, terminal_reward, rng, max_plan_len=mpl)
    ref = oracle.opd_plan_batch(t, r, term, s0, budget, gamma, terminal_reward, rng_ref, done_rule=done_rule,
                                max_plan_len=mpl, n_threads=8)
    np.testing.assert_array_equal(out["status"], ref["status"])
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    assert np.array_equal(out["root_lower"], ref["root_lower"])
    assert np.array_equal(out["root_upper"], ref["root_upper"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    # the whole tree of the first and the last root (bounds after the deferred backup, counts, links), node for node
    cap = 1 + (budget // r.shape[1]) * r.shape[1]
    for root in sorted({0, n_roots - 1}):
        if out["status"][root] != 0:
            continue
        tree = ctx.opd_tree(root, cap)
        one = oracle.opd_plan(t, r, term, int(s0[root]), budget, gamma, terminal_reward, rng0[root].copy(),
                              done_rule=done_rule, max_plan_len=mpl)["tree"]
        for k in one:
            np.testing.assert_array_equal(tree[k], one[k], err_msg="tree[{}] of root {}".format(k, root))
    model.close()
    return out


def _opd_variant(monkeypatch, variant):
    """MP_OPD_MODEL = lds | ldsx | global; "global_cls": the high-occupancy kernel with the residue-class layout of its
    bounds array (MP_OPD_WIDE=cls; the default is the sibling layout, round 5)."""
    monkeypatch.setenv("MP_OPD_MODEL", variant.split("_")[0])
    if variant.endswith("_cls"):
        monkeypatch.setenv("MP_OPD_WIDE", "cls")


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global", "global_cls"])
def test_opd_batch_highway_budget5000(ctx, variant, monkeypatch):
    """C4 shape: highway-shaped S=10 000, A=5, budget 5000 (1000 expansions), with the upper-bound array in LDS
    (40 KB per root; "ldsx": parent map in HBM so that four roots fit a CU) and in HBM/L2 (the high-occupancy
    variant used for big batches; both layouts)."""
    from rl_agents_amd.envs import generators
    _opd_variant(monkeypatch, variant)
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    _cmp_opd(ctx, cfg, 96, 5000, 0.8, seed=5)


def test_opd_budget_beyond_lds(ctx):
    """budget 25 000 (40 K of LDS would be 200 KB): only the HBM-resident bounds array can hold it."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(500, 5, seed=77, terminal_rate=0.02)
    _cmp_opd(ctx, cfg, 6, 25000, 0.9, seed=6)


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global", "global_cls"])
@pytest.mark.parametrize("n_actions,budget", [(2, 101), (3, 200), (4, 100), (5, 500), (7, 300), (13, 1300), (20, 2000),
                                              (64, 640), (5, 10000)])     # (the last: more than 128 entries per class)
def test_opd_batch_action_counts(ctx, n_actions, budget, variant, monkeypatch):
    from rl_agents_amd.envs import generators
    _opd_variant(monkeypatch, variant)
    cfg = generators.random_deterministic(300, n_actions, seed=40 + n_actions, terminal_rate=0.05)
    _cmp_opd(ctx, cfg, 70, budget, 0.9, terminal_reward=0.25, seed=n_actions)


@pytest.mark.parametrize("variant", ["lds", "global", "global_cls"])
@pytest.mark.parametrize("gamma", [0.8, 0.9, 0.95, 0.6])
def test_opd_children_an_ulp_above_their_parent(ctx, gamma, variant, monkeypatch):
    """Rewards of exactly 1: in exact arithmetic a child's upper bound EQUALS its parent's (gamma^(d-1) + gamma^d / (1 - gamma) =
    gamma^(d-1) / (1 - gamma)), so every leaf ties with every other -- and rounded, some children land an ulp ABOVE the
    maximum the selection holds (6 of 59 depths at gamma = 0.8).  The wide kernel's select_drain / rescan_best keep that
    maximum across expansions and must notice; plans, bounds, generator states and whole trees vs the oracle."""
    from rl_agents_amd.envs import generators
    _opd_variant(monkeypatch, variant)
    cfg = generators.random_deterministic(200, 5, seed=310, terminal_rate=0.02)
    ones = dict(cfg, reward=np.ones_like(cfg["reward"]))
    _cmp_opd(ctx, ones, 40, 1500, gamma, seed=int(gamma * 100))
    two = dict(cfg, reward=np.where(cfg["reward"] > 0.5, 1.0, 0.5))
    _cmp_opd(ctx, two, 40, 1500, gamma, terminal_reward=0.5, seed=int(gamma * 100) + 1)


@pytest.mark.parametrize("n_actions,budget", [(65, 650), (100, 1999), (130, 1300), (257, 2000)])
def test_opd_more_actions_than_lanes(ctx, n_actions, budget):
    """Round 4: |A| > 64 (the reference has no bound) runs on the plain kernel -- children 64 at a time, leaf argmax as a scan;
    rewards with many ties (one decimal), a terminal reward, negative terminal reward, plans / bounds / trees vs the oracle."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(120, n_actions, seed=90 + n_actions, terminal_rate=0.05)
    cfg = dict(cfg, reward=np.round(cfg["reward"], 1))
    _cmp_opd(ctx, cfg, 40, budget, 0.9, terminal_reward=0.25, seed=n_actions)
    _cmp_opd(ctx, cfg, 5, budget, 0.7, terminal_reward=-0.5, seed=n_actions + 1, done_rule="next")
    # more than 64 children tie for the plan's choice (constant rewards; gamma = 0 makes whole levels tie): the fuzz sweep found
    # the ORACLE capping its tie list at 64 entries there -- random_argmax draws among all of them (abstract.py:304-311)
    flat = dict(cfg, reward=np.full_like(cfg["reward"], 0.5))
    _cmp_opd(ctx, flat, 6, budget, 0.9, seed=n_actions + 2)
    _cmp_opd(ctx, cfg, 6, budget, 0.0, seed=n_actions + 3)


@pytest.mark.parametrize("variant", ["lds", "ldsx"])
@pytest.mark.parametrize("n_actions,budget", [(4, 100), (5, 2500), (13, 1300), (64, 640)])
def test_opd_closing_passes_on_the_node_array(ctx, n_actions, budget, variant, monkeypatch):
    """MP_OPD_CLOSING=chain: the closing passes the kernel falls back to when the tables of opd_closing.hpp do not fit
    (|A| < 4), forced where they do -- both forms must give the oracle's trees and plans."""
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_OPD_MODEL", variant)
    monkeypatch.setenv("MP_OPD_CLOSING", "chain")
    cfg = generators.random_deterministic(300, n_actions, seed=140 + n_actions, terminal_rate=0.05)
    _cmp_opd(ctx, cfg, 70, budget, 0.9, terminal_reward=0.25, seed=n_actions)
    cfg = generators.highway_shaped(6, 8, 40, seed=2)
    if n_actions == 5:
        _cmp_opd(ctx, cfg, 70, budget, 0.95, seed=3)


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global", "global_cls"])
def test_opd_general_main_loop_where_the_fast_one_applies(ctx, variant, monkeypatch):
    """MP_OPD_LOOP=0: the main loop for arbitrary bounds (taken by itself when the terminal reward is negative or gamma is
    outside [0, 1)), forced where the loop for bounds >= 0 applies."""
    from rl_agents_amd.envs import generators
    _opd_variant(monkeypatch, variant)
    monkeypatch.setenv("MP_OPD_LOOP", "0")
    _cmp_opd(ctx, generators.highway_shaped(6, 8, 40, seed=2), 70, 2500, 0.95, seed=3)
    _cmp_opd(ctx, generators.random_deterministic(300, 7, seed=11, terminal_rate=0.05), 70, 700, 0.9, terminal_reward=0.25, seed=5)


def test_opd_negative_terminal_reward_lowers_an_expanded_node(ctx):
    """terminal_reward < 0: a done child's bound lies BELOW its parent's creation-time bound, so an expanded node's final
    lower bound can be smaller than the value it had as a leaf -- the backups must not keep the old value."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(60, 4, seed=9, terminal_rate=0.4)
    _cmp_opd(ctx, cfg, 70, 400, 0.9, terminal_reward=-2.0, seed=4)
    _cmp_opd(ctx, cfg, 70, 400, 0.9, terminal_reward=-2.0, seed=4, done_rule="next")


def test_opd_ties_draw_from_the_generator(ctx):
    """Constant rewards make every lower bound tie: the plan is drawn through the PCG64 tie-break."""
    cfg = dict(transition=np.tile(np.arange(6).reshape(6, 1), (1, 3)), reward=np.full((6, 3), 0.5),
               terminal=np.zeros(6, bool))
    out = _cmp_opd(ctx, cfg, 40, 60, 0.8, seed=8)
    assert len(set(map(tuple, out["plans"]))) > 1


def test_opd_gridworld_c1_and_next_rule(ctx):
    from rl_agents_amd.envs import generators
    _cmp_opd(ctx, generators.gridworld(), 100, 100, 0.8, seed=1)
    cfg = generators.random_deterministic(200, 4, seed=3, terminal_rate=0.2)
    _cmp_opd(ctx, cfg, 50, 400, 0.85, terminal_reward=1.0, seed=2, done_rule="next")


def test_opd_budget_smaller_than_actions(ctx):
    from rl_agents_amd.envs import generators
    out = _cmp_opd(ctx, generators.random_deterministic(20, 5, seed=1), 3, 4, 0.8)
    assert (out["plan_len"] == 0).all() and (out["env_steps"] == 0).all()


def test_vi_highway_c2_and_robust_c5_shapes(ctx):
    """C2 (S=10 000) and a C5-shaped robust pair (S=50 000, M=2): bit-exact with the oracle."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    q, sweeps = ctx.vi_solve(model, 0.95, 200)
    q_ref, sweeps_ref = oracle.vi_solve("deterministic", cfg["transition"], cfg["reward"], cfg["terminal"], gamma=0.95,
                                        iterations=200)
    assert sweeps == sweeps_ref and np.array_equal(q, q_ref)
    big = generators.highway_shaped(10, 50, 100, seed=2)
    big2 = generators.rewire(big, 0.1, seed=3)
    tt = np.stack([big["transition"], big2["transition"]])
    rr = np.stack([big["reward"], big2["reward"] * 0.97])
    model = ctx.load_table(tt, rr)
    q, sweeps = ctx.vi_solve(model, 0.9, 150, robust=True)
    q_ref, sweeps_ref = oracle.vi_solve("deterministic", tt, rr, None, gamma=0.9, iterations=150, robust=True)
    assert sweeps == sweeps_ref and np.array_equal(q, q_ref)


def test_vi_sparse_and_dense_vs_oracle(ctx):
    from oracle import oracle
    from rl_agents_amd.envs import generators
    for b in (1, 2, 7, 8, 9, 23, 128, 129, 300, 1000, 8200):   # (> 128: numpy's recursion; > 8192: its buffer pieces)
        cfg = generators.random_sparse(211, 3, b, seed=b, terminal_rate=0.1)
        model = ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])
        q, sweeps = ctx.vi_solve(model, 0.9, 60)
        q_ref, sweeps_ref = oracle.vi_solve("sparse", cfg["transition"], cfg["reward"], cfg["terminal"], gamma=0.9,
                                            iterations=60, next_states=cfg["next"])
        assert sweeps == sweeps_ref and np.array_equal(q, q_ref), b
    for s, a in ((17, 2), (64, 3), (333, 5), (1029, 2)):       # ragged tiles: S*A % 64 != 0, S % 16 != 0
        cfg = generators.random_stochastic(s, a, seed=s, terminal_rate=0.1)
        model = ctx.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
        q, sweeps = ctx.vi_solve(model, 0.9, 40)
        q_ref, sweeps_ref = oracle.vi_solve("stochastic", cfg["transition"], cfg["reward"], cfg["terminal"], gamma=0.9,
                                            iterations=40)
        assert sweeps == sweeps_ref and np.array_equal(q, q_ref)     # default: numpy's order of additions
        ctx.vi_dense_mode("mfma")
        try:
            q, sweeps = ctx.vi_solve(model, 0.9, 40)
        finally:
            ctx.vi_dense_mode("exact")
        # matrix-core accumulation order differs from numpy's pairwise sum: 1e-12 relative
        np.testing.assert_allclose(q, q_ref, rtol=1e-12, atol=1e-12)
        assert abs(sweeps - sweeps_ref) <= 1


def test_vi_properties_at_scale(ctx):
    """Size-independent properties on a dense S=2000 model: gamma-contraction fixed point and monotonicity in R."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_stochastic(2000, 5, seed=1)
    model = ctx.load_dense(cfg["transition"], cfg["reward"], None)
    q, _ = ctx.vi_solve(model, 0.9, 400, rtol=0.0, atol=1e-13)
    v = q.max(axis=1)
    np.testing.assert_allclose(q, cfg["reward"] + 0.9 * (cfg["transition"] @ v), rtol=1e-10, atol=1e-10)
    model2 = ctx.load_dense(cfg["transition"], cfg["reward"] + 0.1, None)
    q2, _ = ctx.vi_solve(model2, 0.9, 400, rtol=0.0, atol=1e-13)
    np.testing.assert_allclose(q2 - q, 0.1 / (1 - 0.9), rtol=1e-9)   # constant reward shift -> shift / (1 - gamma)


@pytest.mark.parametrize("dense_mode", ["exact", "mfma"])
def test_row_block_backup_and_sharded_driver_world1(ctx, dense_mode, request):
    """mp_vi_backup on row blocks reassembles the full backup; the sharded driver at world size 1 returns
    exactly what mp_vi_solve returns (same kernel, same order) -- in both forms of the dense contraction."""
    from rl_agents_amd.distributed import vi_solve_row_sharded
    from rl_agents_amd.envs import generators
    ctx.vi_dense_mode(dense_mode)
    request.addfinalizer(lambda: ctx.vi_dense_mode("exact"))
    cfg = generators.random_stochastic(301, 3, seed=5, terminal_rate=0.1)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    full = ctx.load_dense(t, r, term)
    v = np.random.Generator(np.random.PCG64(1)).random(301)
    q_full = ctx.vi_backup(full, 0.9, v)
    parts = []
    for lo, hi in ((0, 100), (100, 101), (101, 301)):
        blk = ctx.load_dense_rows(t[lo:hi], r[lo:hi], term[lo:hi])
        parts.append(ctx.vi_backup(blk, 0.9, v))
    assert np.array_equal(np.concatenate(parts), q_full)
    np.testing.assert_allclose(q_full, r + 0.9 * np.where(term[:, None], 0.0, t @ v), rtol=1e-13)
    q, sweeps = ctx.vi_solve(full, 0.9, 80)
    q2, sweeps2 = vi_solve_row_sharded(ctx, t, r, term, gamma=0.9, iterations=80)
    assert sweeps == sweeps2 and np.array_equal(q, q2)
    # robust pair
    cfg2 = generators.random_stochastic(301, 3, seed=6)
    tt, rr = np.stack([t, cfg2["transition"]]), np.stack([r, cfg2["reward"]])
    both = ctx.load_dense(tt, rr)
    qr, sr = ctx.vi_solve(both, 0.9, 60, robust=True)
    qr2, sr2 = vi_solve_row_sharded(ctx, tt, rr, None, gamma=0.9, iterations=60, robust=True)
    assert sr == sr2 and np.array_equal(qr, qr2)


def test_limits_and_error_codes(ctx):
    """Unsupported configurations fail loudly with the documented error codes (no silent fallback)."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    small = generators.random_deterministic(30, 3, seed=1)
    model = ctx.load_table(small["transition"], small["reward"], small["terminal"])
    rng = _rng_states(2)
    p = np.ones(3) / 3
    ctx.uct_plan(model, [0, 1], 4, 5000, 0.9, 1.0, p, p, rng, max_plan_len=4)   # (round 4: a deep horizon spills the path stack)
    assert ctx.last_kernel_variant() == "uct_global_spill"
    with pytest.raises(native.NativeError) as e:                      # ... until the gamma ** h table itself outgrows the LDS
        ctx.uct_plan(model, [0, 1], 4, 20000, 0.9, 1.0, p, p, rng, max_plan_len=4)
    assert e.value.code == native.ERR_ARG
    with pytest.raises(native.NativeError) as e:                      # gamma = 1: the reference divides by 1 - gamma
        ctx.opd_plan(model, [0, 1], 30, 1.0, 0.0, rng)
    assert e.value.code == native.ERR_ARG
    with pytest.raises(ValueError):                                   # wrong policy length
        ctx.uct_plan(model, [0, 1], 4, 5, 0.9, 1.0, np.ones(2) / 2, p, rng)
    with pytest.raises(native.NativeError) as e:                      # transition index out of range
        ctx.load_table([[0, 7]], [[0.0, 0.0]])
    assert e.value.code == native.ERR_ARG
    dense = generators.random_stochastic(12, 2, seed=1)
    dmodel = ctx.load_dense(dense["transition"], dense["reward"], None)
    with pytest.raises(native.NativeError) as e:                      # tree search needs a deterministic table
        ctx.uct_plan(dmodel, [0, 1], 4, 5, 0.9, 1.0, np.ones(2) / 2, np.ones(2) / 2, rng)
    assert e.value.code == native.ERR_MODE
    sparse = generators.random_sparse(12, 2, 2, seed=1)
    smodel = ctx.load_sparse(sparse["transition"], sparse["next"], sparse["reward"], None)
    with pytest.raises(native.NativeError) as e:                      # robust VI has no sparse mode ("Unknown mode")
        ctx.vi_solve(smodel, 0.9, 5, robust=True)
    assert e.value.code == native.ERR_MODE


def test_uct_large_state_space_without_compact_table(ctx):
    """S >= 32768: no uint16 transition table exists; the record-gather kernel handles it (also with MP_UCT_MODEL=lds)."""
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(40000, 4, seed=12, terminal_rate=0.02)
    p = np.ones(4) / 4
    _cmp_uct(ctx, cfg, 300, 20, 10, 0.9, 5.0, p, p, seed=2)


def test_uct_many_actions_generic_kernel(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(64, 40, seed=13)
    p = np.ones(40) / 40
    _cmp_uct(ctx, cfg, 100, 12, 4, 0.8, 10.0, p, p, seed=3)


def test_row_sharded_driver_device_resident():
    """The device-resident sharded driver (torch tensors, ctx on torch's stream) equals mp_vi_solve at world size 1."""
    torch = pytest.importorskip("torch")
    from rl_agents_amd import native
    from rl_agents_amd.distributed import vi_solve_row_sharded_device
    from rl_agents_amd.envs import generators
    cfg = generators.random_stochastic(257, 3, seed=9, terminal_rate=0.1)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c = native.Context(0, torch.cuda.current_stream().cuda_stream)
        t = torch.from_numpy(cfg["transition"]).cuda()
        r = torch.from_numpy(cfg["reward"]).cuda()
        term = torch.from_numpy(cfg["terminal"].astype(np.uint8)).cuda()
        q_dev, sweeps = vi_solve_row_sharded_device(c, t, r, term, 257, (0, 257), gamma=0.9, iterations=80)
        full = c.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
        q, sweeps_ref = c.vi_solve(full, 0.9, 80)
        assert sweeps == sweeps_ref and np.array_equal(q_dev.cpu().numpy(), q)
        c.close()


def _random_policy_tables(s, a, seed, zero_rate=0.15):
    """Seeded per-state distributions with some exactly-zero entries (never a whole row)."""
    g = np.random.Generator(np.random.PCG64(seed))
    w = g.random((2, s, a)) ** 3
    drop = g.random((2, s, a)) < zero_rate
    drop[:, np.arange(s), g.integers(0, a, size=s)] = False
    w = np.where(drop, 0.0, w)
    return w[0] / w[0].sum(axis=1, keepdims=True), w[1] / w[1].sum(axis=1, keepdims=True)


@pytest.mark.parametrize("coarse_bits", [None, 5, "packed"])
@pytest.mark.parametrize("n_actions", [2, 3, 4, 5, 6, 7, 8])
def test_uct_state_policies_batch_action_counts(ctx, n_actions, coarse_bits, monkeypatch):
    """Per-state prior / rollout tables (mcts_with_prior.py:47-62), every |A| specialisation, 300 roots vs the oracle.
    coarse_bits = 5 keeps only 5 bits of each threshold in the fused records, so that the exact-row fallback (taken
    once in ~2^32 steps otherwise) runs every few steps."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    if coarse_bits == "packed":     # the 16-byte records of saturated batches (|A| <= 5), forced on this small one
        monkeypatch.setenv("MP_UCT_POLICY_RECORD", "packed")
    elif coarse_bits:
        monkeypatch.setenv("MP_UCT_COARSE_BITS", str(coarse_bits))
    cfg = generators.random_deterministic(301, n_actions, seed=40 + n_actions, terminal_rate=0.04)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    prior, rollout = _random_policy_tables(301, n_actions, seed=n_actions)
    model = ctx.load_table(t, r, term, max_steps=40)
    policy = ctx.load_policy(model, prior, rollout)
    n = 300
    s0 = np.random.Generator(np.random.PCG64(5)).integers(0, 301, size=n).astype(np.int32)
    rng = _rng_states(n, base=777)
    rng_ref = rng.copy()
    out = ctx.uct_plan(model, s0, 30, 12, 0.9, 6.5, None, None, rng, max_plan_len=12, policy=policy)
    ref = oracle.uct_plan_batch(t, r, term, s0, 30, 12, 0.9, 6.5, prior, rollout, rng_ref, max_steps=40,
                                max_plan_len=12, n_threads=8)
    for k in ("plans", "plan_len", "root_child_count", "env_steps"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert np.array_equal(out["root_value"], ref["root_value"])
    assert np.array_equal(out["root_child_value"], ref["root_child_value"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    policy.close()
    model.close()


def test_uct_state_policies_headline_shape_and_uniform_equivalence(ctx):
    """Highway-shaped S=10 000, A=5: (i) Boltzmann-like per-state tables vs the oracle on 1100 ragged roots;
    (ii) uniform tables give exactly what the state-independent uniform policy gives (same stream, same plans)."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    prior, rollout = _random_policy_tables(10000, 5, seed=9, zero_rate=0.05)
    policy = ctx.load_policy(model, prior, rollout)
    n = 1100 + 13
    s0 = np.random.Generator(np.random.PCG64(3)).integers(0, 10000, size=n).astype(np.int32)
    rng = _rng_states(n, base=31)
    rng_ref = rng.copy()
    out = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, None, None, rng, max_plan_len=30, policy=policy)
    ref = oracle.uct_plan_batch(t, r, term, s0, 33, 30, 0.8, 10.0, prior, rollout, rng_ref, max_plan_len=30, n_threads=8)
    for k in ("plans", "plan_len", "root_child_count", "env_steps"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    uni = np.full((10000, 5), 0.2)
    upol = ctx.load_policy(model, uni, uni)
    rng_a, rng_b = _rng_states(n, base=99), _rng_states(n, base=99)
    a = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, None, None, rng_a, max_plan_len=30, policy=upol)
    b = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, np.full(5, 0.2), np.full(5, 0.2), rng_b, max_plan_len=30)
    for k in ("plans", "root_child_count", "env_steps"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.array_equal(a["root_value"], b["root_value"])
    np.testing.assert_array_equal(rng_a, rng_b)
    for x in (policy, upol, model):
        x.close()


def test_policy_load_errors(ctx):
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(50, 9, seed=1)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    # |A| = 9: no register-resident specialisation in uct.hip (2..8 have one) -- the policy loads (round 4) and plans through
    # the loop forms of the other kernel, on this deterministic table too; mp_uct_plan_policy itself says where to go
    pol9 = ctx.load_policy(model, np.full((50, 9), 1 / 9), np.full((50, 9), 1 / 9))
    with pytest.raises(native.NativeError, match="mp_uct_plan_stochastic_policy"):
        ctx.uct_plan(model, [0], 5, 5, 0.8, 10.0, None, None, _rng_states(1), policy=pol9)
    out = ctx.uct_plan_stochastic(model, [0, 3], 5, 5, 0.8, 10.0, None, None, _rng_states(2), policy=pol9)
    assert (out["plan_len"] >= 1).all()
    pol9.close()
    model.close()
    cfg = generators.random_deterministic(50, 4, seed=1)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    bad = np.full((50, 4), 0.25)
    bad[3, 1] = -0.1
    with pytest.raises(native.NativeError):
        ctx.load_policy(model, bad, np.full((50, 4), 0.25))
    with pytest.raises(native.NativeError):
        ctx.load_policy(model, np.full((50, 4), 0.25), np.zeros((50, 4)))
    # a policy carries the fused records of ITS model: another model of the same shape is refused
    policy = ctx.load_policy(model, np.full((50, 4), 0.25), np.full((50, 4), 0.25))
    other = generators.random_deterministic(50, 4, seed=2)
    model2 = ctx.load_table(other["transition"], other["reward"], other["terminal"])
    with pytest.raises(native.NativeError):
        ctx.uct_plan(model2, [0], 5, 5, 0.8, 10.0, None, None, _rng_states(1), policy=policy)
    policy.close()
    model2.close()
    model.close()


def test_lds_atomics_apply_in_lane_order(ctx):
    """What the state-aware kernel's grouped backup relies on (csrc/saopd.hip apply_vec): same-address LDS atomics of one
    wave instruction apply in lane order -- checked on THIS device over 131 072 wave instructions with random masks."""
    assert ctx.selftest_lds_atomic_order(131072) == 0


@pytest.mark.parametrize("mapping", ["wave", "wave-serial-prune", "wave-par-backup", "wave-seq-backup", "wave-global", "wave-ordered",
                                     "wave-flat", "wave-buckets", "lane"])
@pytest.mark.parametrize("shape", ["grid", "garnet", "highway"])
def test_state_aware_batch_vs_oracle(ctx, shape, mapping, monkeypatch):
    """200 planners per launch, three consecutive plans each (planner state kept on the device), vs the oracle run
    planner by planner; ragged outcomes included (planners whose leaves all get pruned report MP_ERR_ARG)."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_SAOPD_MODEL", mapping.split("-")[0])
    if mapping == "wave-serial-prune":      # one scratch entry per lane: states with two leaves take the serial prune pass
        monkeypatch.setenv("MP_SAOPD_LANE_SCRATCH", "1")
    if mapping == "wave-par-backup":        # the grouped parallel backup in EVERY plan (default: the first plan only): the
        monkeypatch.setenv("MP_SAOPD_PAR_BACKUP", "1")   # chunked lists are rebuilt from the linked ones by the later plans
    if mapping == "wave-seq-backup":        # ... and in none: the element-by-element loop
        monkeypatch.setenv("MP_SAOPD_PAR_BACKUP", "0")
    if mapping == "wave-global":            # round 4: the dictionaries stay in global memory (default: LDS where they fit)
        monkeypatch.setenv("MP_SAOPD_DICT", "0")
    if mapping == "wave-ordered":           # round 4: dispatch by expected cost also for this small batch (default: > 32 per CU)
        monkeypatch.setenv("MP_SAOPD_ORDER", "1")
    if mapping == "wave-flat":              # round 4: the earlier plans' rows as one flat list in every plan ...
        monkeypatch.setenv("MP_SAOPD_CSR", "0")
    if mapping == "wave-buckets":           # ... and bucketed by state from the second plan on (default: from the fourth)
        monkeypatch.setenv("MP_SAOPD_CSR", "1")
    cfg, budget, gamma = {"grid": (generators.gridworld(), 120, 0.8),
                          "garnet": (generators.random_deterministic(40, 3, seed=5, terminal_rate=0.1), 90, 0.7),
                          "highway": (generators.highway_shaped(3, 4, 10, seed=3), 150, 0.9)}[shape]
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    n = 200
    model = ctx.load_table(t, r, term)
    g = np.random.Generator(np.random.PCG64(17))
    states = g.integers(0, r.shape[0], size=n).astype(np.int32)
    if mapping == "wave-ordered":           # an earlier batch on the model: the first plans below start longest first
        warm = native.StateAwarePlanners(ctx, model, n)
        warm.plan(states[::-1].copy(), budget, gamma, 0.0, _rng_states(n, base=1))
        warm.close()
    planners = native.StateAwarePlanners(ctx, model, n)
    rng = _rng_states(n, base=4242)
    ref_rng = rng.copy()
    ref_planner = [None] * n
    dead = np.zeros(n, bool)
    for step in range(3):
        out = planners.plan(states, budget, gamma, 0.0, rng)
        for i in range(n):
            if dead[i]:
                continue
            try:
                o = oracle.saopd_plan(t, r, term, int(states[i]), budget, gamma, rng_state=ref_rng[i],
                                      planner=ref_planner[i], max_plan_len=budget + 1)
            except ValueError:
                assert out["status"][i] == native.MP_ERR_ARG, (step, i)
                dead[i] = True
                continue
            assert out["status"][i] == 0, (step, i)
            np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"], err_msg=str((step, i)))
            assert out["env_steps"][i] == o["env_steps"] and out["updates"][i] == o["updates"], (step, i)
            np.testing.assert_array_equal(rng[i], o["rng_after"])
            ref_rng[i], ref_planner[i] = o["rng_after"], o["planner"]
            if i % 37 == 0:
                tree, sv = planners.export(i)
                assert np.array_equal(sv, o["state_values"]), (step, i)
                for k in ("parent", "first_child", "state", "depth", "lower", "reward", "alive", "count"):
                    assert np.array_equal(tree[k], o["tree"][k]), (step, i, k)
        # every planner moves on along its own plan (dead ones stay put; their results are no longer compared)
        states = np.where(out["plan_len"] > 0, t[states, np.maximum(out["plans"][:, 0], 0)], states).astype(np.int32)
    assert (~dead).sum() > 20      # garnets prune themselves empty often; most grids and highways survive
    planners.close()
    model.close()


@pytest.mark.parametrize("n_actions", [9, 16, 33, 40, 70])   # (70: more actions than lanes -- the one-planner-per-lane kernel)
def test_state_aware_many_actions_vs_oracle(ctx, n_actions):
    """|A| from 9 to 40: the grouped parallel backup runs 7, 4 and (|A| > 32) no groups per pass -- the last is the
    element-by-element loop."""
    from oracle import oracle
    from rl_agents_amd import native
    g = np.random.Generator(np.random.PCG64(300 + n_actions))
    n_states = 14
    t = g.integers(0, n_states, size=(n_states, n_actions), dtype=np.int64)
    r = np.round(g.random((n_states, n_actions)), 2)
    term = g.random(n_states) < 0.1
    n, budget = 20, 12 * n_actions
    model = ctx.load_table(t, r, term)
    planners = native.StateAwarePlanners(ctx, model, n)
    states = g.integers(0, n_states, size=n).astype(np.int32)
    rng = _rng_states(n, base=31)
    ref_rng = rng.copy()
    ref_planner = [None] * n
    dead = np.zeros(n, bool)
    compared = 0
    for step in range(3):
        out = planners.plan(states, budget, 0.85, 0.0, rng)
        for i in range(n):
            if dead[i]:
                continue
            try:
                o = oracle.saopd_plan(t, r, term, int(states[i]), budget, 0.85, rng_state=ref_rng[i], planner=ref_planner[i],
                                      max_plan_len=budget + 1)
            except ValueError:
                assert out["status"][i] == native.MP_ERR_ARG, (step, i)
                dead[i] = True
                continue
            assert out["status"][i] == 0, (step, i)
            np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"], err_msg=str((step, i)))
            assert out["env_steps"][i] == o["env_steps"] and out["updates"][i] == o["updates"], (step, i)
            np.testing.assert_array_equal(rng[i], o["rng_after"])
            ref_rng[i], ref_planner[i] = o["rng_after"], o["planner"]
            compared += 1
        states = np.where(out["plan_len"] > 0, t[states, np.maximum(out["plans"][:, 0], 0)], states).astype(np.int32)
    assert compared > 20
    planners.close()
    model.close()


@pytest.mark.parametrize("dict_lds", ["1", "0"])
@pytest.mark.parametrize("shape", ["grid-12400", "ring-2700"])
def test_state_aware_large_budgets_vs_oracle(ctx, shape, dict_lds, monkeypatch):
    """Round 4: the budget is no longer bounded by LDS.  The wave kernels keep the first depth-table entries in LDS (37 with
    the dictionaries there, 2 560 without) and read deeper nodes' entries from the global copy: the 10x10 grid at budget 12 400
    reaches depth 223, a one-action ring at budget 2 700 depth 2 700.  Plans, Bellman-backup counts, trees vs the oracle."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_SAOPD_DICT", dict_lds)
    if shape == "grid-12400":
        cfg = generators.gridworld()
        t, r, term, budget = cfg["transition"], cfg["reward"], cfg["terminal"], 12400
    else:
        g = np.random.Generator(np.random.PCG64(8))
        t, r, term, budget = ((np.arange(30) + 1) % 30).reshape(30, 1), np.round(g.random((30, 1)), 2), np.zeros(30, bool), 2700
    states = np.array([5, 17, 29], dtype=np.int32)
    n = len(states)
    model = ctx.load_table(t, r, term)
    planners = native.StateAwarePlanners(ctx, model, n)
    rng = _rng_states(n, base=77)
    ref_rng = rng.copy()
    out = planners.plan(states, budget, 0.8, 0.0, rng, max_plan_len=budget + 1)
    for i in range(n):
        o = oracle.saopd_plan(t, r, term, int(states[i]), budget, 0.8, rng_state=ref_rng[i], max_plan_len=budget + 1)
        assert out["status"][i] == 0
        np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"])
        assert out["env_steps"][i] == o["env_steps"] and out["updates"][i] == o["updates"]
        np.testing.assert_array_equal(rng[i], o["rng_after"])
        tree, sv = planners.export(i)
        assert np.array_equal(sv, o["state_values"])
        for k in ("parent", "depth", "lower", "alive"):
            assert np.array_equal(tree[k], o["tree"][k]), k
        assert tree["depth"].max() == o["tree"]["depth"].max() and tree["depth"].max() > (200 if shape == "grid-12400" else 2600)
    planners.close()
    model.close()


@pytest.mark.parametrize("prune_rows", [None, 64, 0])
@pytest.mark.parametrize("n_states,n_actions,budget", [(2, 3, 420), (3, 2, 500), (6, 4, 480)])
def test_state_aware_long_lists_vs_oracle(ctx, n_states, n_actions, budget, prune_rows, monkeypatch):
    """Tiny state spaces over several plans: a state's node list grows to many hundred rows, which takes the prune pass
    off its register sets (more than 256 rows of changed states -> one state per round; one state with more than 256 rows
    -> the streamed form).  Every plan, alive flag and state value vs the oracle."""
    from oracle import oracle
    from rl_agents_amd import native
    if prune_rows is not None:   # test knob: fewer rows through the register sets (0: everything through the streamed form)
        monkeypatch.setenv("MP_SAOPD_PRUNE_ROWS", str(prune_rows))
    g = np.random.Generator(np.random.PCG64(100 + n_states))
    t = g.integers(0, n_states, size=(n_states, n_actions), dtype=np.int64)
    r = np.round(g.random((n_states, n_actions)), 1)
    term = np.zeros(n_states, bool)
    n = 24
    model = ctx.load_table(t, r, term)
    planners = native.StateAwarePlanners(ctx, model, n)
    states = g.integers(0, n_states, size=n).astype(np.int32)
    rng = _rng_states(n, base=99)
    ref_rng = rng.copy()
    ref_planner = [None] * n
    dead = np.zeros(n, bool)
    compared = 0
    for step in range(5):
        out = planners.plan(states, budget, 0.9, 0.0, rng)
        for i in range(n):
            if dead[i]:
                continue
            try:
                o = oracle.saopd_plan(t, r, term, int(states[i]), budget, 0.9, rng_state=ref_rng[i], planner=ref_planner[i],
                                      max_plan_len=budget + 1)
            except ValueError:
                assert out["status"][i] == native.MP_ERR_ARG, (step, i)
                dead[i] = True
                continue
            assert out["status"][i] == 0, (step, i)
            np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"], err_msg=str((step, i)))
            assert out["env_steps"][i] == o["env_steps"] and out["updates"][i] == o["updates"], (step, i)
            np.testing.assert_array_equal(rng[i], o["rng_after"])
            ref_rng[i], ref_planner[i] = o["rng_after"], o["planner"]
            if i % 5 == 0:
                tree, sv = planners.export(i)
                assert np.array_equal(sv, o["state_values"]), (step, i)
                for k in ("parent", "first_child", "state", "depth", "lower", "reward", "alive", "count"):
                    assert np.array_equal(tree[k], o["tree"][k]), (step, i, k)
            compared += 1
        states = np.where(out["plan_len"] > 0, t[states, np.maximum(out["plans"][:, 0], 0)], states).astype(np.int32)
    assert compared > 40
    planners.close()
    model.close()


def test_state_aware_queue_overflow_is_reported(ctx, monkeypatch):
    """A backup queue that is too small is a per-planner MP_ERR_ALLOC status, not a silent truncation."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.gridworld()
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    monkeypatch.setenv("MP_SAOPD_QUEUE", "512")      # >= 1 + budget (the prune pass lists candidates in it)
    monkeypatch.setenv("MP_SAOPD_QUEUE_LIMIT_MB", "0")   # ... and no room to grow it
    planners = native.StateAwarePlanners(ctx, model, 64)
    rng = _rng_states(64, base=5)
    out = planners.plan(np.arange(64, dtype=np.int32), 400, 0.8, 0.0, rng, max_plan_len=4)
    assert (out["status"] == native.MP_ERR_ALLOC).any() and set(np.unique(out["status"])) <= {0, native.MP_ERR_ALLOC}
    # a planner left with a full queue has lost backups: it stays failed, loudly, in every later call
    again = planners.plan(np.arange(64, dtype=np.int32), 8, 0.8, 0.0, rng, max_plan_len=4)
    full = out["status"] == native.MP_ERR_ALLOC
    assert (again["status"][full] == native.MP_ERR_ALLOC).all() and (again["plan_len"][full] == 0).all()
    assert (again["status"][~full] == 0).any()   # (the others go on; some may fill the tiny queue now)
    planners.close()
    model.close()


def test_state_aware_device_mode_is_asynchronous_and_loud(ctx, monkeypatch):
    """mem = MP_MEM_DEVICE: one launch, nothing read back.  Results equal the host-mode call's; a planner whose queue
    fills up reports MP_ERR_ALLOC (no roll-back possible then) and keeps reporting it, the others are unaffected."""
    torch = pytest.importorskip("torch")
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.gridworld()
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    n, budget, mpl = 64, 400, 4
    s0 = np.arange(n, dtype=np.int32)
    dev = torch.device("cuda", 0)

    def device_plan(planners, rng_np, b):
        d = dict(root=torch.from_numpy(s0).to(dev), rng=torch.from_numpy(rng_np.view(np.int64)).to(dev),
                 plans=torch.zeros((n, mpl), dtype=torch.int32, device=dev), plan_len=torch.zeros(n, dtype=torch.int32, device=dev),
                 env_steps=torch.zeros(n, dtype=torch.int64, device=dev), updates=torch.zeros(n, dtype=torch.int64, device=dev),
                 status=torch.zeros(n, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()  # (the uploads ran on torch's stream, the plan runs on the context's)
        planners.plan_device(d["root"], b, 0.8, 0.0, d["rng"], mpl, plans=d["plans"], plan_len=d["plan_len"],
                             env_steps=d["env_steps"], updates=d["updates"], status=d["status"])
        ctx.synchronize()
        return {k: v.cpu().numpy() for k, v in d.items()}

    # a comfortable queue: the asynchronous call gives what the synchronous one gives
    ref_planners = native.StateAwarePlanners(ctx, model, n)
    rng = _rng_states(n, base=5)
    ref = ref_planners.plan(s0, budget, 0.8, 0.0, rng.copy(), max_plan_len=mpl)
    ref_planners.close()
    planners = native.StateAwarePlanners(ctx, model, n)
    out = device_plan(planners, rng.copy(), budget)
    for k in ("plans", "plan_len", "env_steps", "updates", "status"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    planners.close()
    # a queue that is too small: reported per planner, sticky, the others still equal the reference
    monkeypatch.setenv("MP_SAOPD_QUEUE", "512")
    planners = native.StateAwarePlanners(ctx, model, n)
    out = device_plan(planners, rng.copy(), budget)
    full = out["status"] == native.MP_ERR_ALLOC
    assert full.any() and not full.all()
    for k in ("plans", "plan_len", "env_steps", "updates"):
        np.testing.assert_array_equal(out[k][~full], ref[k][~full], err_msg=k)
    again = device_plan(planners, rng.copy(), 8)
    assert (again["status"][full] == native.MP_ERR_ALLOC).all() and (again["status"][~full] == 0).any()
    planners.close()
    model.close()


def test_results_do_not_depend_on_batch_composition(ctx):
    """Size-independent property at BASELINE shapes: a root's plan depends on its own state and random stream only --
    planning a strided subset of a 20 000-root batch (other wave positions, other launch geometry) reproduces the
    subset of the full batch's results, for UCT, UCT with per-state policies, OPD and state-aware OPD."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    n = 20000
    s0 = np.random.Generator(np.random.PCG64(8)).integers(0, 10000, size=n).astype(np.int32)
    rng0 = np.random.Generator(np.random.PCG64(9)).integers(1, 2 ** 62, size=(n, 6)).astype(np.uint64)
    rng0[:, 3] |= 1
    rng0[:, 4:] = 0
    sub = np.arange(0, n, 7)
    p = np.ones(5) / 5
    q, _ = ctx.vi_solve(model, 0.95, 100)
    z = np.exp((q - q.max(axis=1, keepdims=True)) / 0.5)
    policy = ctx.load_policy(model, z / z.sum(axis=1, keepdims=True), z / z.sum(axis=1, keepdims=True))

    def uct(idx, pol):
        rng = np.ascontiguousarray(rng0[idx])
        out = ctx.uct_plan(model, s0[idx], 33, 30, 0.8, 10.0, p, p, rng, max_plan_len=6, policy=pol)
        return out, rng
    for pol in (None, policy):
        full, rng_full = uct(np.arange(n), pol)
        part, rng_part = uct(sub, pol)
        for k in ("plans", "root_child_count", "env_steps"):
            np.testing.assert_array_equal(full[k][sub], part[k], err_msg=k)
        assert np.array_equal(full["root_value"][sub], part["root_value"])
        np.testing.assert_array_equal(rng_full[sub], rng_part)

    def opd(idx):
        rng = np.ascontiguousarray(rng0[idx])
        return ctx.opd_plan(model, s0[idx], 400, 0.8, 0.0, rng, max_plan_len=81)
    full, part = opd(np.arange(4000)), opd(np.arange(0, 4000, 7))
    for k in ("plans", "env_steps", "status"):
        np.testing.assert_array_equal(full[k][::7], part[k], err_msg=k)
    assert np.array_equal(full["root_lower"][::7], part["root_lower"]) and np.array_equal(full["root_upper"][::7], part["root_upper"])

    grid = generators.gridworld()
    gmodel = ctx.load_table(grid["transition"], grid["reward"], grid["terminal"])

    def sa(idx):
        planners = native.StateAwarePlanners(ctx, gmodel, len(idx))
        rng = np.ascontiguousarray(rng0[idx])
        out = planners.plan((s0[idx] % 100).astype(np.int32), 200, 0.8, 0.0, rng, max_plan_len=8)
        planners.close()
        return out
    full, part = sa(np.arange(3000)), sa(np.arange(0, 3000, 7))
    for k in ("plans", "env_steps", "updates", "status"):
        np.testing.assert_array_equal(full[k][::7], part[k], err_msg=k)
    for x in (policy, model, gmodel):
        x.close()


@pytest.mark.parametrize("mapping", ["wave", "lane"])
def test_state_aware_queue_grows_by_rollback(ctx, mapping, monkeypatch):
    """A backup queue that fills up is not fatal: the plan is rolled back (per-state dictionaries, list links,
    generator states) and run again with a queue four times as large -- same results as the oracle, on fresh
    planners and on planners that carry state from earlier plans.  Zero rewards make the state values decay to
    floating-point underflow: tens of thousands of backups per plan."""
    from oracle import oracle
    from rl_agents_amd import native
    monkeypatch.setenv("MP_SAOPD_MODEL", mapping)
    monkeypatch.setenv("MP_SAOPD_QUEUE", "256")
    t = np.array([[0, 2, 2, 1], [1, 1, 2, 2], [1, 1, 0, 1]])
    r = np.zeros((3, 4))
    r[2, 1] = 0.25
    term = np.zeros(3, bool)
    model = ctx.load_table(t, r, term)
    n = 6
    planners = native.StateAwarePlanners(ctx, model, n)
    rng = _rng_states(n, base=77)
    ref_rng, ref_pl = rng.copy(), [None] * n
    states = (np.arange(n) % 3).astype(np.int32)
    most = 0
    for step, budget in enumerate((8, 120, 120)):   # the small first plan leaves state behind: the next one rolls back non-fresh planners
        out = planners.plan(states, budget, 0.9, 0.0, rng)
        assert (out["status"] == 0).all(), out["status"]
        most = max(most, int(out["updates"].max()))
        for i in range(n):
            o = oracle.saopd_plan(t, r, term, int(states[i]), budget, 0.9, rng_state=ref_rng[i], planner=ref_pl[i], max_plan_len=121)
            np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"], err_msg=str((step, i)))
            assert out["updates"][i] == o["updates"], (step, i)
            np.testing.assert_array_equal(rng[i], o["rng_after"])
            tree, sv = planners.export(i)
            assert np.array_equal(sv, o["state_values"]) and np.array_equal(tree["alive"], o["tree"]["alive"])
            ref_rng[i], ref_pl[i] = o["rng_after"], o["planner"]
        states = t[states, out["plans"][:, 0]].astype(np.int32)
    assert most > 2000
    planners.close()
    model.close()


def test_small_plans_through_pinned_scratch_equal_pageable_ones(ctx, monkeypatch):
    """Host-array plans of at most 64 roots run through pinned scratch arrays of the context (zero-copy; round 5): same plans,
    statistics, env steps and advanced generator records as the pageable path (MP_NO_PINNED_SCRATCH=1), results owned by the
    caller (a second call does not overwrite the first one's arrays)."""
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(6, 8, 40, seed=2)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    p = np.ones(5) / 5
    for n in (1, 7, 64):
        s0 = np.random.Generator(np.random.PCG64(n)).integers(0, cfg["reward"].shape[0], size=n).astype(np.int32)
        steps = np.arange(n, dtype=np.int32) % 3
        rng_a, rng_b = _rng_states(n, base=5), _rng_states(n, base=5)
        a = ctx.uct_plan(model, s0, 20, 12, 0.9, 1.0, p, p, rng_a, root_steps=steps)
        a2 = ctx.uct_plan(model, s0, 20, 12, 0.9, 1.0, p, p, rng_a.copy(), root_steps=steps)      # (same scratch arrays again)
        monkeypatch.setenv("MP_NO_PINNED_SCRATCH", "1")
        b = ctx.uct_plan(model, s0, 20, 12, 0.9, 1.0, p, p, rng_b, root_steps=steps)
        monkeypatch.delenv("MP_NO_PINNED_SCRATCH")
        assert set(a) == set(b)
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
            assert a[k] is not a2[k]
        np.testing.assert_array_equal(rng_a, rng_b)
        rng_a, rng_b = _rng_states(n, base=9), _rng_states(n, base=9)
        a = ctx.opd_plan(model, s0, 300, 0.9, 0.0, rng_a, max_plan_len=70)
        monkeypatch.setenv("MP_NO_PINNED_SCRATCH", "1")
        b = ctx.opd_plan(model, s0, 300, 0.9, 0.0, rng_b, max_plan_len=70)
        monkeypatch.delenv("MP_NO_PINNED_SCRATCH")
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        np.testing.assert_array_equal(rng_a, rng_b)
    model.close()
