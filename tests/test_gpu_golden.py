"""HIP path vs the reference's golden vectors, through the C ABI (libmi355plan.so), bit for bit.

The goldens (tests/golden/*.npz) are outputs of the unmodified Python reference; everything must be
identical, dense VI included (its default form restates numpy's order of additions); the
matrix-core form of dense VI is compared with the tolerance stated in the test.
"""
import numpy as np
import pytest

from tests.helpers import mdp_from_golden, assert_tree_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def _load(ctx, cfg):
    if cfg["mode"] == "deterministic":
        return ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"], max_steps=cfg["max_steps"])
    if cfg["mode"] == "stochastic":
        return ctx.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
    return ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])


def test_library_loaded_is_in_tree():
    from rl_agents_amd import native
    lib = native.load()
    assert lib.mp_abi_version() == 7
    assert "rl_agents_amd/lib/libmi355plan.so" in native.lib_path()


def test_value_iteration_golden(ctx, golden):
    z = golden["vi"]
    for name in [str(n) for n in z["vi/names"]]:
        p = "vi/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        model = _load(ctx, cfg)
        gamma, iters = float(z[p + "/gamma"]), int(z[p + "/iterations"])
        q, sweeps = ctx.vi_solve(model, gamma, iters)
        v = ctx.vi_solve_v(model, gamma, iters)
        # (dense models: the default contraction restates numpy's order of additions, see tests/test_gpu_vi_dense_exact.py)
        assert sweeps == int(z[p + "/sweeps"]), name
        assert np.array_equal(q, z[p + "/Q"]), name
        assert np.array_equal(v, z[p + "/V"]), name
        if cfg["mode"] == "stochastic":
            # the matrix-core form: f64 MFMA accumulation order != numpy's: tolerance 1e-12 relative, sweep count within one
            ctx.vi_dense_mode("mfma")
            try:
                q, sweeps = ctx.vi_solve(model, gamma, iters)
                v = ctx.vi_solve_v(model, gamma, iters)
            finally:
                ctx.vi_dense_mode("exact")
            np.testing.assert_allclose(q, z[p + "/Q"], rtol=1e-12, atol=1e-12, err_msg=name)
            np.testing.assert_allclose(v, z[p + "/V"], rtol=1e-12, atol=1e-12, err_msg=name)
            assert abs(sweeps - int(z[p + "/sweeps"])) <= 1, name
        np.testing.assert_array_equal(q.argmax(axis=1), z[p + "/actions"], err_msg=name)
        model.close()


def test_robust_value_iteration_golden(ctx, golden):
    z = golden["vi"]
    for name in [str(n) for n in z["rvi/names"]]:
        p = "rvi/" + name
        mode = str(z[p + "/mode"])
        if mode == "deterministic":
            model = ctx.load_table(z[p + "/transitions"], z[p + "/rewards"])
        else:
            model = ctx.load_dense(z[p + "/transitions"], z[p + "/rewards"])
        q, sweeps = ctx.vi_solve(model, float(z[p + "/gamma"]), int(z[p + "/iterations"]), robust=True)
        assert sweeps == int(z[p + "/sweeps"]), name
        assert np.array_equal(q, z[p + "/Q"]), name
        if mode != "deterministic":
            ctx.vi_dense_mode("mfma")
            try:
                q, sweeps = ctx.vi_solve(model, float(z[p + "/gamma"]), int(z[p + "/iterations"]), robust=True)
            finally:
                ctx.vi_dense_mode("exact")
            np.testing.assert_allclose(q, z[p + "/Q"], rtol=1e-12, atol=1e-12, err_msg=name)
        n = len(z[p + "/actions"])
        np.testing.assert_array_equal(q.argmax(axis=1)[:n], z[p + "/actions"], err_msg=name)
        model.close()


def test_opd_golden(ctx, golden):
    z = golden["opd"]
    for name in [str(n) for n in z["opd/names"]]:
        p = "opd/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        model = _load(ctx, cfg)
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        budget = int(z[p + "/budget"])
        out = ctx.opd_plan(model, [int(z[p + "/s0"])], budget, float(z[p + "/gamma"]),
                           float(z[p + "/terminal_reward"]), rng, max_plan_len=budget + 1)
        assert out["status"][0] == 0
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[p + "/plan"], err_msg=name)
        assert out["root_lower"][0] == float(z[p + "/root_lower"]), name
        assert out["root_upper"][0] == float(z[p + "/root_upper"]), name
        assert out["env_steps"][0] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(rng[0], z[p + "/rng_after"], err_msg=name)
        a = cfg["reward"].shape[1]
        tree = ctx.opd_tree(0, 1 + (budget // a) * a)
        assert tree["count"][0] == int(z[p + "/root_count"])
        assert_tree_equal(z, p + "/tree", tree, a, dict(count="count", lower="lower", upper="upper", reward="reward",
                                                        done="done", depth="depth"))
        model.close()


def test_opd_reward_range_status(ctx):
    from rl_agents_amd import native
    t = [[1, 2], [1, 1], [3, 4], [3, 3], [4, 4]]
    r = [[0, 0], [0, 0], [0, 0], [1, 1], [-1, -1]]
    model = ctx.load_table(t, r, [0, 1, 0, 1, 1])
    out = ctx.opd_plan(model, [0], 20, 0.8, 0.0, np.array([[1, 2, 3, 5, 0, 0]], np.uint64))
    assert out["status"][0] == native.ERR_REWARD_RANGE


def test_uct_golden(ctx, golden):
    z = golden["uct"]
    for name in [str(n) for n in z["uct/names"]]:
        p = "uct/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        model = _load(ctx, cfg)
        a = cfg["reward"].shape[1]
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        episodes, horizon = int(z[p + "/episodes"]), int(z[p + "/horizon"])
        out = ctx.uct_plan(model, [int(z[p + "/s0"])], episodes, horizon, float(z[p + "/gamma"]),
                           float(z[p + "/temperature"]), z[p + "/prior_p"], z[p + "/rollout_p"], rng,
                           root_steps=[int(z[p + "/steps0"])])
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[p + "/plan"], err_msg=name)
        assert out["env_steps"][0] == int(z[p + "/env_steps"]), name
        assert out["root_value"][0] == float(z[p + "/root_value"]), name
        np.testing.assert_array_equal(rng[0], z[p + "/rng_after"], err_msg=name)
        tree = ctx.uct_tree(0, 1 + episodes * a)
        assert tree["count"][0] == int(z[p + "/root_count"])
        assert_tree_equal(z, p + "/tree", tree, a, dict(count="count", value="value"))
        model.close()


@pytest.mark.parametrize("variant", ["default", "ldsr"])
@pytest.mark.parametrize("tag", ["subtree_large1", "subtree_highway"])
def test_uct_subtree_strategy_golden(ctx, golden, tag, variant, monkeypatch):
    """step_strategy 'subtree': mp_uct_step_tree re-roots the kept tree, the next mp_uct_plan continues on it;
    plans, trees and generator state equal the reference's over a 5-step episode -- with the record-gather kernel and with
    the LDS-resident model (the highway table has few distinct rewards; large1 has 500 and falls back)."""
    if variant != "default":
        monkeypatch.setenv("MP_UCT_MODEL", variant)
    z = golden["uct"]
    p = "uct/" + tag
    cfg = mdp_from_golden(z, p + "/mdp")
    model = _load(ctx, cfg)
    a = cfg["reward"].shape[1]
    rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
    prob = np.ones(a) / a
    ctx.uct_reset_tree()
    prev = None
    for step in range(int(z[p + "/n_steps"])):
        if prev is not None:
            ctx.uct_step_tree([prev])
        out = ctx.uct_plan(model, [int(z[p + "/states"][step])], int(z[p + "/episodes"]), int(z[p + "/horizon"]),
                           float(z[p + "/gamma"]), float(z[p + "/temperature"]), prob, prob, rng)
        q = "{}/step{}".format(p, step)
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[q + "/plan"], err_msg=q)
        np.testing.assert_array_equal(rng[0], z[q + "/rng_after"], err_msg=q)
        tree = ctx.uct_tree(0)
        assert tree["count"][0] == int(z[q + "/root_count"]) and tree["value"][0] == float(z[q + "/root_value"])
        assert_tree_equal(z, q + "/tree", tree, a, dict(count="count", value="value"))
        prev = int(out["plans"][0, 0])
    model.close()


def test_uct_state_policies_golden(ctx, golden):
    """mp_uct_plan_policy vs the unmodified MCTSWithPriorPolicyAgent (mcts_with_prior.py) driven by a prior agent:
    plans, trees, env-step counts and generator states, bit for bit."""
    z = golden["uct_prior"]
    for name in [str(n) for n in z["uct_prior/names"]]:
        p = "uct_prior/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        model = _load(ctx, cfg)
        policy = ctx.load_policy(model, z[p + "/prior_table"], z[p + "/rollout_table"])
        a = cfg["reward"].shape[1]
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        episodes, horizon = int(z[p + "/episodes"]), int(z[p + "/horizon"])
        ctx.uct_reset_tree()
        out = ctx.uct_plan(model, [int(z[p + "/s0"])], episodes, horizon, float(z[p + "/gamma"]),
                           float(z[p + "/temperature"]), None, None, rng, policy=policy)
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[p + "/plan"], err_msg=name)
        assert out["env_steps"][0] == int(z[p + "/env_steps"]), name
        assert out["root_value"][0] == float(z[p + "/root_value"]), name
        np.testing.assert_array_equal(rng[0], z[p + "/rng_after"], err_msg=name)
        tree = ctx.uct_tree(0, 1 + episodes * a)
        assert tree["count"][0] == int(z[p + "/root_count"])
        assert_tree_equal(z, p + "/tree", tree, a, dict(count="count", value="value"))
        policy.close()
        model.close()


def test_uct_state_policies_subtree_golden(ctx, golden):
    z = golden["uct_prior"]
    p = "uct_prior/subtree_highway"
    cfg = mdp_from_golden(z, p + "/mdp")
    model = _load(ctx, cfg)
    policy = ctx.load_policy(model, z[p + "/prior_table"], z[p + "/prior_table"])
    a = cfg["reward"].shape[1]
    rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
    ctx.uct_reset_tree()
    prev = None
    for step in range(int(z[p + "/n_steps"])):
        if prev is not None:
            ctx.uct_step_tree([prev])
        out = ctx.uct_plan(model, [int(z[p + "/states"][step])], int(z[p + "/episodes"]), int(z[p + "/horizon"]),
                           float(z[p + "/gamma"]), float(z[p + "/temperature"]), None, None, rng, policy=policy)
        q = "{}/step{}".format(p, step)
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[q + "/plan"], err_msg=q)
        np.testing.assert_array_equal(rng[0], z[q + "/rng_after"], err_msg=q)
        tree = ctx.uct_tree(0)
        assert tree["count"][0] == int(z[q + "/root_count"]) and tree["value"][0] == float(z[q + "/root_value"])
        assert_tree_equal(z, q + "/tree", tree, a, dict(count="count", value="value"))
        prev = int(out["plans"][0, 0])
    ctx.uct_reset_tree()
    policy.close()
    model.close()


@pytest.mark.parametrize("mapping", ["wave", "wave-global", "lane"])
def test_state_aware_planner_golden_episodes(ctx, golden, mapping, monkeypatch):
    """(one planner per wavefront / per lane) mp_saopd_plan vs the unmodified StateAwarePlannerAgent over multi-plan episodes: plans, trees, leaves sets,
    state-value tables, env-step counts and generator state, with the planner state carried across plans; where
    the reference raises (every leaf pruned) the planner reports MP_ERR_ARG."""
    from rl_agents_amd import native
    from tests.helpers import replay_state_aware_episode
    monkeypatch.setenv("MP_SAOPD_MODEL", mapping.split("-")[0])
    if mapping == "wave-global":    # (round 4: the default wave kernel keeps the per-state dictionaries in LDS; this one does not)
        monkeypatch.setenv("MP_SAOPD_DICT", "0")
    z = golden["state_aware"]
    for name in [str(n) for n in z["sa/names"]]:
        cfg = mdp_from_golden(z, "sa/{}/mdp".format(name))
        model = _load(ctx, cfg)
        planners = native.StateAwarePlanners(ctx, model, 1)

        def plan_fn(cfg, s0, params, rng, planner_state):
            rng = np.array(rng, dtype=np.uint64).reshape(1, 6)
            out = planners.plan([s0], params["budget"], params["gamma"], params["terminal_reward"], rng,
                                accuracy=params["accuracy"], backup_aggregated_nodes=params["backup_aggregated_nodes"],
                                prune_suboptimal_leaves=params["prune_suboptimal_leaves"])
            if out["status"][0] == native.MP_ERR_ARG:
                raise ValueError("max() arg is an empty sequence")
            assert out["status"][0] == 0
            tree, sv = planners.export(0)
            return dict(plan=out["plans"][0, :out["plan_len"][0]], env_steps=out["env_steps"][0], rng_after=rng[0],
                        tree=tree, state_values=sv, planner=planners)
        replay_state_aware_episode(z, name, plan_fn)
        planners.close()
        model.close()
