"""The device against the round-3 golden vectors of the unmodified reference (tests/golden/round3.npz) and against the
oracle on seeded batches: robust V-form value iteration, discrete robust OPD on restricted action sets (all kernel
variants)."""
import os

import numpy as np
import pytest

from tests.helpers import assert_keyed_tree_equal, mdp_from_golden
from tests.test_oracle_round3 import names, robust_models

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RVI = "<class 'rl_agents_amd.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent'>"
DRP = "<class 'rl_agents_amd.agents.robust.robust.DiscreteRobustPlannerAgent'>"


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(REPO, "tests", "golden", "round3.npz"))


@pytest.fixture(scope="module")
def zvi():
    return np.load(os.path.join(REPO, "tests", "golden", "vi.npz"))


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def test_robust_state_value_goldens_agent_and_c_abi(ctx, z, zvi):
    """RobustValueIterationAgent.get_state_value (robust_value_iteration.py:32-37) no longer raises: mp_vi_solve_v_robust."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    for name in names(z, "rvi_v"):
        p = "rvi/" + name
        mode = str(zvi[p + "/mode"])
        t, r = zvi[p + "/transitions"], zvi[p + "/rewards"]
        want = z["rvi_v/{}/V".format(name)]
        model = ctx.load_table(t, r) if mode == "deterministic" else ctx.load_dense(t, r)
        v = ctx.vi_solve_v(model, float(zvi[p + "/gamma"]), int(zvi[p + "/iterations"]), robust=True)
        if mode == "deterministic":
            assert np.array_equal(v, want), name
        else:
            np.testing.assert_allclose(v, want, rtol=1e-12, atol=1e-12, err_msg=name)
        model.close()
        env = FiniteMDPEnv(dict(mode="deterministic", transition=[[0]], reward=[[0.0]]))
        models = [dict(mode=mode, transition=tm.tolist(), reward=rm.tolist()) for tm, rm in zip(t, r)]
        agent = agent_factory(env, dict(__class__=RVI, models=models, gamma=float(zvi[p + "/gamma"]),
                                        iterations=int(zvi[p + "/iterations"])))
        np.testing.assert_allclose(agent.get_state_value(), want, rtol=1e-12, atol=1e-12, err_msg=name)


@pytest.mark.parametrize("n_states,n_models", [(10000, 2), (700, 3), (20000, 1)])
def test_robust_state_value_all_device_paths_vs_oracle(ctx, n_states, n_models, monkeypatch):
    """Persistent launch (S <= 16 384), single-workgroup launch (small S) and chained launches vs the oracle, bit for bit."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfgs = [generators.random_deterministic(n_states, 4, seed=40 + i) for i in range(n_models)]
    t = np.stack([c["transition"] for c in cfgs])
    r = np.stack([c["reward"] * (1 - 0.05 * i) for i, c in enumerate(cfgs)])
    want = oracle.vi_solve("deterministic", t, r, None, gamma=0.9, iterations=150, robust=True, state_value=True)
    model = ctx.load_table(t, r)
    assert np.array_equal(ctx.vi_solve_v(model, 0.9, 150, robust=True), want)
    monkeypatch.setenv("MP_VI_NO_PERSIST", "1")
    monkeypatch.setenv("MP_VI_NO_SMALL", "1")
    assert np.array_equal(ctx.vi_solve_v(model, 0.9, 150, robust=True), want)
    model.close()


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global"])
def test_robust_planner_restricted_actions_goldens(ctx, z, variant, monkeypatch):
    """mp_ropd_plan on joint models that restrict their actions (union over the models, robust.py:22-25): plans, bounds,
    env steps, generator state and whole trees of the reference's DiscreteRobustPlanner, in every kernel variant."""
    monkeypatch.setenv("MP_OPD_MODEL", variant)
    for name in names(z, "robust_masked"):
        p = "robust_masked/" + name
        t, r, term = robust_models(z, p)
        m, _, a = r.shape
        budget = int(z[p + "/budget"])
        model = ctx.load_joint(t, r, term, available=z[p + "/available"])
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        out = ctx.ropd_plan(model, [int(z[p + "/s0"])], budget, float(z[p + "/gamma"]), float(z[p + "/terminal_reward"]), rng,
                            max_plan_len=budget // a + 1)
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[p + "/plan"], err_msg=name)
        assert out["root_lower"][0] == float(z[p + "/root_lower"]) and out["root_upper"][0] == float(z[p + "/root_upper"]), name
        assert int(out["env_steps"][0]) == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(rng[0], z[p + "/rng_after"], err_msg=name)
        tree = ctx.ropd_tree(0, 1 + (budget // a) * a, m)
        tree["obs"] = np.where(np.arange(len(tree["parent"]))[:, None] == 0, -1, tree["state"])
        tree["lower_min"], tree["upper_min"] = tree["lower"].min(axis=1), tree["upper"].min(axis=1)
        assert_keyed_tree_equal(z, p + "/tree", tree, dict(count="count", depth="depth", lower_min="lower_min",
                                                          upper_min="upper_min", reward="reward", done="done", obs="obs",
                                                          n_children="n_children"))
        model.close()


def test_robust_planner_agent_on_restricted_models(z):
    """DiscreteRobustPlannerAgent through agent_factory with `models` = preprocessor lists that yield masked envs."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import MaskedFiniteMDPEnv
    for name in names(z, "robust_masked"):
        p = "robust_masked/" + name
        if not bool(np.all(z[p + "/has_mask"])):
            continue                                      # (a model without get_available_actions: C-ABI test above)
        m = int(z[p + "/n_models"])
        cfgs = [mdp_from_golden(z, "{}/mdp{}".format(p, i)) for i in range(m)]

        def table(c, av):
            return dict(mode="deterministic", transition=c["transition"].tolist(), reward=c["reward"].tolist(),
                        terminal=np.asarray(c["terminal"]).astype(int).tolist(), available=np.asarray(av).astype(int).tolist())
        env = MaskedFiniteMDPEnv(dict(table(cfgs[0], z[p + "/available"][0]), state=int(z[p + "/s0"])))
        env.reset()
        models = [[{"method": "copy_with_config", "args": table(c, z[p + "/available"][i])}] for i, c in enumerate(cfgs)]
        agent = agent_factory(env, dict(__class__=DRP, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                                        terminal_reward=float(z[p + "/terminal_reward"]), models=models))
        agent.seed(int(z[p + "/seed"]))
        np.testing.assert_array_equal(agent.plan(int(z[p + "/s0"])), z[p + "/plan"], err_msg=name)
        root = agent.planner.root
        assert root.count == int(z[p + "/root_count"]) and root.get_value() == float(z[p + "/root_upper"]), name


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global"])
@pytest.mark.parametrize("n_models,n_actions,budget", [(1, 3, 150), (2, 5, 500), (3, 4, 300), (5, 7, 280), (16, 2, 100)])
def test_robust_planner_restricted_actions_batch_vs_oracle(ctx, n_models, n_actions, budget, variant, monkeypatch):
    from oracle import oracle
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_OPD_MODEL", variant)
    s_ = 200
    cfgs = [generators.random_deterministic(s_, n_actions, seed=60 + i, terminal_rate=0.05) for i in range(n_models)]
    t, r = np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs])
    term = np.stack([c["terminal"] for c in cfgs])
    avail = np.stack([generators.random_available(s_, n_actions, seed=i + 1, rate=0.55) for i in range(n_models)])
    model = ctx.load_joint(t, r, term, available=avail)
    n = 70
    g = np.random.Generator(np.random.PCG64(n_models * 10 + n_actions))
    s0 = g.integers(0, s_, size=(n, n_models)).astype(np.int32)          # distinct joint states
    rng = g.integers(0, 2 ** 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
    rng[:, 3] |= np.uint64(1)
    rng[:, 4:] = 0
    rng_ref = rng.copy()
    mpl = budget // n_actions + 2
    out = ctx.ropd_plan(model, s0, budget, 0.9, 0.25, rng, max_plan_len=mpl)
    ref = oracle.ropd_plan_batch(t, r, term, s0, budget, 0.9, 0.25, rng_ref, max_plan_len=mpl, n_threads=8, available=avail)
    for k in ("status", "plans", "plan_len", "env_steps"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert np.array_equal(out["root_lower"], ref["root_lower"]) and np.array_equal(out["root_upper"], ref["root_upper"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()
