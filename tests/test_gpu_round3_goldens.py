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
        assert np.array_equal(v, want), name      # (dense models too: the default contraction is numpy's order)
        model.close()
        env = FiniteMDPEnv(dict(mode="deterministic", transition=[[0]], reward=[[0.0]]))
        models = [dict(mode=mode, transition=tm.tolist(), reward=rm.tolist()) for tm, rm in zip(t, r)]
        agent = agent_factory(env, dict(__class__=RVI, models=models, gamma=float(zvi[p + "/gamma"]),
                                        iterations=int(zvi[p + "/iterations"])))
        assert np.array_equal(agent.get_state_value(), want), name


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


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global", "global_cls"])
def test_robust_planner_restricted_actions_goldens(ctx, z, variant, monkeypatch):
    """mp_ropd_plan on joint models that restrict their actions (union over the models, robust.py:22-25): plans, bounds,
    env steps, generator state and whole trees of the reference's DiscreteRobustPlanner, in every kernel variant."""
    monkeypatch.setenv("MP_OPD_MODEL", variant.split("_")[0])      # "global_cls": the wide kernels' residue-class layout
    if variant.endswith("_cls"):
        monkeypatch.setenv("MP_OPD_WIDE", "cls")
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


@pytest.mark.parametrize("variant", ["lds", "ldsx", "global", "global_cls"])
@pytest.mark.parametrize("n_models,n_actions,budget", [(1, 3, 150), (2, 5, 500), (3, 4, 300), (5, 7, 280), (16, 2, 100), (24, 3, 150),
                                                        (32, 2, 100), (33, 2, 100), (70, 3, 150), (2, 65, 700), (3, 130, 1400)])   # (|A| > 64: the plain kernel;
                                                        # > 32 models: done flags beyond the node's 32 bits ride in the reward's sign)
def test_robust_planner_restricted_actions_batch_vs_oracle(ctx, n_models, n_actions, budget, variant, monkeypatch):
    from oracle import oracle
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_OPD_MODEL", variant.split("_")[0])      # "global_cls": the wide kernels' residue-class layout
    if variant.endswith("_cls"):
        monkeypatch.setenv("MP_OPD_WIDE", "cls")
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


@pytest.mark.parametrize("variant", ["lds", "global", "global_cls"])
@pytest.mark.parametrize("n_models", [33, 40, 70])
def test_robust_planner_more_than_32_models_tree_export(ctx, n_models, variant, monkeypatch):
    """The exported tree of a plan over more than 32 models (the reference has no bound: JointEnv steps a list,
    robust.py:9-16): per-model states, rewards, done flags and bound vectors of every node against the oracle's tree."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    monkeypatch.setenv("MP_OPD_MODEL", variant.split("_")[0])      # "global_cls": the wide kernels' residue-class layout
    if variant.endswith("_cls"):
        monkeypatch.setenv("MP_OPD_WIDE", "cls")
    s_, a_, budget = 120, 3, 90
    cfgs = [generators.random_deterministic(s_, a_, seed=300 + i, terminal_rate=0.15) for i in range(n_models)]
    t, r = np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs])
    r[:, ::7, :] = 0.0                                     # rewards of exactly 0. on terminal steps too (-0. carries the flag)
    term = np.stack([c["terminal"] for c in cfgs])
    model = ctx.load_joint(t, r, term)
    g = np.random.Generator(np.random.PCG64(n_models))
    s0 = g.integers(0, s_, size=(1, n_models)).astype(np.int32)
    rng = np.array([[5, 7, 0, 9, 0, 0]], dtype=np.uint64)
    out = ctx.ropd_plan(model, s0, budget, 0.85, 0.5, rng.copy(), max_plan_len=budget)
    ref = oracle.ropd_plan(t, r, term, s0[0], budget, 0.85, 0.5, rng_state=rng[0].copy(), max_plan_len=budget)
    n = int(out["plan_len"][0])
    assert np.array_equal(out["plans"][0, :n], ref["plan"]) and out["root_lower"][0] == ref["root_lower"]
    assert out["root_upper"][0] == ref["root_upper"] and int(out["env_steps"][0]) == ref["env_steps"]
    tree = ctx.ropd_tree(0, 1 + (budget // a_) * a_, n_models)
    assert len(tree["parent"]) == len(ref["tree"]["parent"])
    assert ref["tree"]["done"][:, 32:].any() and not ref["tree"]["done"][:, 32:].all()
    for k in ("parent", "action", "depth", "count", "n_children", "state", "done", "reward", "lower", "upper"):
        assert np.array_equal(np.asarray(tree[k]), ref["tree"][k]), k
    assert not np.signbit(np.asarray(tree["reward"])).any()
    model.close()


# ------------------------------------------------------------------ MCTS on stochastic finite MDPs (uct_stoch.hip)
UCT = "<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>"


def _stoch_model(ctx, cfg):
    if cfg["mode"] == "stochastic":
        model = ctx.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
    elif cfg["mode"] == "sparse":
        model = ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])
    else:
        return ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"], max_steps=cfg["max_steps"])
    model.set_episode_rules("source", cfg["max_steps"])
    return model


def test_uct_on_stochastic_models_goldens_c_abi(ctx, z):
    """mp_uct_plan_stochastic against the reference MCTSAgent on `stochastic` / `sparse` finite MDPs, open and closed
    loop: plans with observation keys, values, env steps, the planner's generator, whole trees (observation layer)."""
    from tests.helpers import assert_parent_tree_equal
    from tests.test_oracle_round3 import stoch_case
    for name in names(z, "uct_stoch"):
        p = "uct_stoch/" + name
        cfg, _, _ = stoch_case(z, p)
        model = _stoch_model(ctx, cfg)
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        closed = bool(z[p + "/closed_loop"])
        out = ctx.uct_plan_stochastic(model, [int(z[p + "/s0"])], int(z[p + "/episodes"]), int(z[p + "/horizon"]),
                                      float(z[p + "/gamma"]), float(z[p + "/temperature"]), z[p + "/prior_p"], z[p + "/rollout_p"],
                                      rng, env_rng_state=z[p + "/env_rng"].reshape(1, 6), closed_loop=closed,
                                      root_steps=[int(z[p + "/steps0"])])
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], z[p + "/plan"], err_msg=name)
        assert out["root_value"][0] == float(z[p + "/root_value"]), name
        assert int(out["env_steps"][0]) == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(rng[0], z[p + "/rng_after"], err_msg=name)
        tree = ctx.uct_stoch_tree(0)
        assert_parent_tree_equal(z, p + "/tree", tree, dict(count="count", value="value", is_obs="is_obs"))
        model.close()


def test_uct_on_stochastic_models_agent(z):
    """MCTSAgent through agent_factory on a stochastic FiniteMDPEnv: plan() with string observation keys, planner.root
    with the observation layer, the env's generator left untouched."""
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    from tests.test_gpu_variants import _agent_tree
    from tests.test_oracle_round3 import stoch_case
    for name in names(z, "uct_stoch"):
        p = "uct_stoch/" + name
        cfg, _, _ = stoch_case(z, p)
        c = dict(mode=cfg["mode"], transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"],
                 max_steps=cfg["max_steps"], state=int(z[p + "/s0"]))
        if "next" in cfg:
            c["next"] = cfg["next"]
        env = FiniteMDPEnv(c)
        env.reset()
        env.steps = int(z[p + "/steps0"])
        env.seed(1000 + int(z[p + "/seed"]))
        assert np.array_equal(native.rng_state_from_generator(env.np_random), z[p + "/env_rng"])
        acfg = dict(__class__=UCT, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]), temperature=float(z[p + "/temperature"]),
                    horizon=int(z[p + "/horizon"]), episodes=int(z[p + "/episodes"]), closed_loop=bool(z[p + "/closed_loop"]))
        if "pref" in name:
            acfg.update(prior_policy={"type": "preference", "action": 1, "ratio": 3},
                        rollout_policy={"type": "preference", "action": 1, "ratio": 3})
        agent = agent_factory(env, acfg)
        agent.seed(int(z[p + "/seed"]))
        plan = agent.plan(int(z[p + "/s0"]))
        np.testing.assert_array_equal([int(x) for x in plan], z[p + "/plan"], err_msg=name)
        assert [isinstance(x, str) for x in plan] == list(z[p + "/plan_is_obs"]), name
        assert plan == agent.planner.get_plan()
        assert agent.planner.env_steps == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), z[p + "/rng_after"])
        np.testing.assert_array_equal(native.rng_state_from_generator(env.np_random), z[p + "/env_rng"])    # never stepped
        t = _agent_tree(agent.planner.root)
        np.testing.assert_array_equal(t["parent"], z[p + "/tree/parent"])
        np.testing.assert_array_equal(t["action"], z[p + "/tree/action"])
        for f in ("count", "value", "prior", "is_obs"):
            assert np.array_equal(t[f].astype(z[p + "/tree/" + f].dtype), z[p + "/tree/" + f]), (name, f)


@pytest.mark.parametrize("mode,closed", [("stochastic", False), ("stochastic", True), ("sparse", False), ("sparse", True),
                                         ("deterministic", True), ("sparse2", True), ("sparse2", False), ("sparse6", True),
                                         ("sparse-unfused", True), ("sparse-many-actions", True), ("sparse-generic-a", True),
                                         ("dense-few2", True), ("dense-few4", False), ("dense-few4", True),
                                         ("sparse2-few-rewards", True), ("sparse2-few-rewards", False), ("dense-few2-few-rewards", True),
                                         ("sparse2-few-rewards-32", True)])
def test_uct_on_stochastic_models_batch_vs_oracle(ctx, mode, closed, monkeypatch):
    """Seeded batches of 300 roots (ragged last wave), distinct planner AND env generator records per root, a TimeLimit
    and both terminal conventions: plans, values, env steps and generator records equal the oracle's."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    few_rewards = "few-rewards" in mode     # at most 256 distinct rewards + two successors: the 16-byte records
    if mode.endswith("-32"):
        monkeypatch.setenv("MP_UCT_STOCH_FUSED", "2")       # ... kept at 32 bytes
        mode = mode[:-3]
    mode = mode.replace("-few-rewards", "")
    if mode == "stochastic":
        cfg = generators.random_stochastic(90, 4, seed=21, terminal_rate=0.05, concentration=0.1)
    elif mode.startswith("dense-few"):
        # a DENSE model whose rows hold at most 2 / 4 non-zero entries (zeros between, before and after them; successors
        # listed twice merge): the device samples it through the fused records, the oracle through the full rows
        b = int(mode[-1])
        sp = generators.random_sparse(150, 4, b, seed=30 + b, terminal_rate=0.05)
        dense = np.zeros((150, 4, 150))
        for j in range(b):
            np.add.at(dense, (np.arange(150)[:, None], np.arange(4)[None, :], sp["next"][:, :, j]), sp["transition"][:, :, j])
        cfg = dict(transition=dense, reward=sp["reward"], terminal=sp["terminal"])
        mode = "stochastic"
    elif mode.startswith("sparse"):
        # 3 / 2 successors: the fused 64- / 32-byte records; 6: rows by binary search; "unfused": MP_UCT_STOCH_FUSED=0;
        # 11 actions: the generic selection (more than eight children)
        b = {"sparse": 3, "sparse2": 2, "sparse6": 6, "sparse-unfused": 3, "sparse-many-actions": 2, "sparse-generic-a": 2}[mode]
        n_act = 11 if mode == "sparse-many-actions" else 5
        cfg = generators.random_sparse(400, n_act, b, seed=22 + b, terminal_rate=0.05)
        if mode == "sparse-unfused":
            monkeypatch.setenv("MP_UCT_STOCH_FUSED", "0")
        if mode == "sparse-generic-a":                       # the loop form of the selection where |A| <= 8
            monkeypatch.setenv("MP_UCT_STOCH_GENERIC_A", "1")
        mode = "sparse"
    else:
        cfg = generators.random_deterministic(200, 4, seed=23, terminal_rate=0.05)
    if few_rewards:
        cfg = dict(cfg, reward=np.round(cfg["reward"] * 37) / 37)      # 38 values, most of them not exact in binary
    a = cfg["reward"].shape[1]
    n = 300
    g = np.random.Generator(np.random.PCG64(77))
    s0 = g.integers(0, cfg["reward"].shape[0], size=n).astype(np.int32)
    steps0 = g.integers(0, 6, size=n).astype(np.int32)

    def records():
        r = g.integers(0, 2 ** 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
        r[:, 3] |= np.uint64(1)
        r[:, 4:] = 0
        return r
    rng, erng = records(), records()
    rng_ref = rng.copy()
    prior = g.random(a) + 0.1
    prior /= prior.sum()
    roll = g.random(a) + 0.1
    roll /= roll.sum()
    for rule in ("source", "next"):
        if mode == "stochastic":
            model = ctx.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
        elif mode == "sparse":
            model = ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])
        else:
            model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
        model.set_episode_rules(rule, 25)
        out = ctx.uct_plan_stochastic(model, s0, 24, 9, 0.9, 4.5, prior, roll, rng, env_rng_state=erng, closed_loop=closed,
                                      root_steps=steps0, max_plan_len=18)
        ref = oracle.uct_plan_stoch_batch(mode, cfg["transition"], cfg["reward"], cfg["terminal"], s0, 24, 9, 0.9, 4.5, prior,
                                          roll, rng_ref, erng, next_states=cfg.get("next"), closed_loop=closed, steps0=steps0,
                                          max_steps=25, done_rule=rule, max_plan_len=18, n_threads=8)
        for k in ("plans", "plan_len", "env_steps"):
            np.testing.assert_array_equal(out[k], ref[k], err_msg="{} {}".format(rule, k))
        assert np.array_equal(out["root_value"], ref["root_value"])
        np.testing.assert_array_equal(rng, ref["rng_after"])
        rng_ref = ref["rng_after"].copy()
        model.close()


@pytest.mark.parametrize("closed", [False, True])
def test_uct_on_stochastic_models_long_plans(ctx, closed):
    """2 300 episodes: more nodes than a 16-bit path entry can name (the int32 path stack) and visit counts beyond the
    quotient tables (the kernel's own IEEE divisions) -- against the oracle."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.random_sparse(120, 4, 2, seed=5, terminal_rate=0.03)
    n, episodes, horizon = 70, 2300, 25
    g = np.random.Generator(np.random.PCG64(5))
    s0 = g.integers(0, 120, size=n).astype(np.int32)

    def records():
        r = g.integers(0, 2 ** 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
        r[:, 3] |= np.uint64(1)
        r[:, 4:] = 0
        return r
    rng, erng = records(), records()
    rng_ref = rng.copy()
    p = np.ones(4) / 4
    model = ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])
    out = ctx.uct_plan_stochastic(model, s0, episodes, horizon, 0.9, 3.0, p, p, rng, env_rng_state=erng, closed_loop=closed,
                                  max_plan_len=2 * horizon)
    ref = oracle.uct_plan_stoch_batch("sparse", cfg["transition"], cfg["reward"], cfg["terminal"], s0, episodes, horizon, 0.9, 3.0,
                                      p, p, rng_ref, erng, next_states=cfg["next"], closed_loop=closed, max_plan_len=2 * horizon,
                                      n_threads=8)
    for k in ("plans", "plan_len", "env_steps"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()


def test_closed_loop_on_a_deterministic_model_through_the_literal_kernel(ctx):
    """Cross-check of round 2's argument (closed loop on a deterministic model = the open-loop statistics + a host-side
    observation layer): the literal kernel -- real observation nodes -- reproduces the reference's closed-loop goldens of
    variants.npz on deterministic tables."""
    zv = np.load(os.path.join(REPO, "tests", "golden", "variants.npz"))
    from tests.helpers import assert_parent_tree_equal
    done = 0
    for name in [str(n) for n in zv["closed/names"]]:
        p = "closed/" + name
        cfg = mdp_from_golden(zv, p + "/mdp")
        model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"], max_steps=cfg["max_steps"])
        rng = np.array(zv[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        out = ctx.uct_plan_stochastic(model, [int(zv[p + "/s0"])], int(zv[p + "/episodes"]), int(zv[p + "/horizon"]),
                                      float(zv[p + "/gamma"]), float(zv[p + "/temperature"]), zv[p + "/prior_p"],
                                      zv[p + "/rollout_p"], rng, closed_loop=True, root_steps=[int(zv[p + "/steps0"])])
        n = int(out["plan_len"][0])
        np.testing.assert_array_equal(out["plans"][0, :n], zv[p + "/plan"], err_msg=name)
        np.testing.assert_array_equal(rng[0], zv[p + "/rng_after"], err_msg=name)
        assert_parent_tree_equal(zv, p + "/tree", ctx.uct_stoch_tree(0), dict(count="count", value="value", is_obs="is_obs"))
        model.close()
        done += 1
    assert done >= 10


# ------------------------------------------------------------------ state-aware OPD on restricted action sets
SAOPD = "<class 'rl_agents_amd.agents.tree_search.state_aware.StateAwarePlannerAgent'>"


@pytest.mark.parametrize("mapping", ["wave", "lane", "lds"])
def test_state_aware_restricted_actions_goldens_c_abi(ctx, z, mapping, monkeypatch):
    """mp_saopd_plan on models carrying an availability table: the reference's StateAwarePlannerAgent episodes on
    MaskedFiniteMDPEnv (ascending listing) and on the highway-like env (IDLE first: planned in the permuted action space)
    -- plans, keyed trees, leaves, state values, env steps, generator -- in every kernel mapping."""
    from rl_agents_amd import native
    from tests.helpers import replay_state_aware_masked_episode
    if mapping == "lane":
        monkeypatch.setenv("MP_SAOPD_MODEL", "lane")
    if mapping == "lds":
        monkeypatch.setenv("MP_SAOPD_LDS", "1")
    held = {}

    def plan_fn(cfg, available, order, s0, params, rng, planner):
        t, r, av = cfg["transition"], cfg["reward"], np.asarray(available)
        o = None if order is None else np.asarray(order)
        if o is not None:
            t, r, av = t[:, o], r[:, o], av[:, o]
        if planner is None:
            for x in held.values():
                x.close()
            held["model"] = ctx.load_table(t, r, cfg["terminal"], available=av)
            held["planners"] = native.StateAwarePlanners(ctx, held["model"], 1)
        rs = np.array(rng, dtype=np.uint64).reshape(1, 6)
        out = held["planners"].plan([s0], params["budget"], params["gamma"], params["terminal_reward"], rs,
                                    accuracy=params["accuracy"], backup_aggregated_nodes=params["backup_aggregated_nodes"],
                                    prune_suboptimal_leaves=params["prune_suboptimal_leaves"])
        if out["status"][0] == native.MP_ERR_ARG:
            raise ValueError("max() arg is an empty sequence")
        assert out["status"][0] == 0
        tree, sv = held["planners"].export(0)
        plan = out["plans"][0, :out["plan_len"][0]]
        if o is not None:
            plan = o[plan]
            tree["action"] = np.where(tree["action"] >= 0, o[np.maximum(tree["action"], 0)], -1)
        return dict(plan=plan, env_steps=int(out["env_steps"][0]), rng_after=rs[0], tree=tree, state_values=sv, planner=True)
    for name in names(z, "sa_masked"):
        replay_state_aware_masked_episode(z, name, plan_fn)
    for x in held.values():
        x.close()


def test_state_aware_agent_on_restricted_action_envs(z):
    """StateAwarePlannerAgent through agent_factory on MaskedFiniteMDPEnv and on HighwayLikeEnv (no NotImplementedError any
    more): the reference's episode, plan by plan."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import HighwayLikeEnv, MaskedFiniteMDPEnv
    for name in names(z, "sa_masked"):
        p = "sa_masked/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        s_start = int(z[p + "/states"][0])
        if bool(z[p + "/listing_idle_first"]):
            env = HighwayLikeEnv(table=dict(transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"],
                                            original_shape=(3, 4, 10)), state=s_start)

            def current(e=env):
                return e.state_index
        else:
            env = MaskedFiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"],
                                          terminal=cfg["terminal"], available=z[p + "/available"], state=s_start))
            env.reset()

            def current(e=env):
                return e.mdp.state
        agent = agent_factory(env, dict(__class__=SAOPD, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                                        terminal_reward=float(z[p + "/terminal_reward"]), accuracy=float(z[p + "/accuracy"]),
                                        backup_aggregated_nodes=bool(z[p + "/backup_aggregated_nodes"]),
                                        prune_suboptimal_leaves=bool(z[p + "/prune_suboptimal_leaves"])))
        agent.seed(int(z[p + "/seed"]))
        raises_at = int(z[p + "/raises_at_step"]) if p + "/raises_at_step" in z.files else -1
        for step in range(int(z[p + "/n_steps"])):
            assert current() == int(z[p + "/states"][step]), name
            if step == raises_at:
                with pytest.raises(ValueError):
                    agent.plan(current())
                break
            plan = agent.plan(current())
            np.testing.assert_array_equal(plan, z["{}/step{}/plan".format(p, step)], err_msg="{} step {}".format(name, step))
            root = agent.planner.root
            assert root.count == int(z["{}/step{}/tree/count".format(p, step)][0]), name
            env.step(plan[0])


@pytest.mark.parametrize("mapping", ["wave", "lane"])
@pytest.mark.parametrize("shape", ["grid", "garnet", "highway"])
def test_state_aware_restricted_actions_batch_vs_oracle(ctx, shape, mapping, monkeypatch):
    """200 planners x 3 consecutive plans on restricted-action models vs the oracle: plans, env steps, Bellman-backup
    counts, status, generator records, and one whole exported arena per shape."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    if mapping == "lane":
        monkeypatch.setenv("MP_SAOPD_MODEL", "lane")
    if shape == "grid":
        cfg, budget, gamma = generators.gridworld(), 200, 0.8
        avail = generators.random_available(100, 4, seed=3, rate=0.35)
    elif shape == "garnet":
        cfg, budget, gamma = generators.random_deterministic(80, 5, seed=41, terminal_rate=0.05), 150, 0.85
        avail = generators.random_available(80, 5, seed=5, rate=0.5)
    else:
        cfg, budget, gamma = generators.highway_shaped(3, 4, 10, seed=3), 150, 0.8
        avail = generators.highway_available(cfg)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    n = 200
    model = ctx.load_table(t, r, term, available=avail)
    planners = native.StateAwarePlanners(ctx, model, n)
    g = np.random.Generator(np.random.PCG64(9))
    states = g.integers(0, s_, size=n).astype(np.int32)
    rng = native.seed_sequence_states([3], 0, n)
    ref_pl = [None] * n
    for step in range(3):
        rng_ref = rng.copy()
        out = planners.plan(states, budget, gamma, 0.25, rng, max_plan_len=budget // a_ + 1)
        for i in range(n):
            if out["status"][i] != 0:
                with pytest.raises(ValueError):
                    oracle.saopd_plan(t, r, term, int(states[i]), budget, gamma, 0.25, rng_state=rng_ref[i], planner=ref_pl[i],
                                      max_plan_len=budget + 1, available=avail)
                continue
            o = oracle.saopd_plan(t, r, term, int(states[i]), budget, gamma, 0.25, rng_state=rng_ref[i], planner=ref_pl[i],
                                  max_plan_len=budget + 1, available=avail)
            ref_pl[i] = o["planner"]
            np.testing.assert_array_equal(out["plans"][i, :out["plan_len"][i]], o["plan"], err_msg="planner {} step {}".format(i, step))
            assert int(out["env_steps"][i]) == o["env_steps"] and int(out["updates"][i]) == o["updates"]
            np.testing.assert_array_equal(rng[i], o["rng_after"])
            if i == 7:
                tree, sv = planners.export(i)
                for k in ("parent", "action", "state", "depth", "lower", "reward", "alive", "first_child", "n_children", "count"):
                    np.testing.assert_array_equal(tree[k], o["tree"][k], err_msg=k)
                assert np.array_equal(sv, o["state_values"])
        ok = out["status"] == 0
        first = np.where(ok & (out["plan_len"] > 0), out["plans"][:, 0], 0)
        states = np.where(ok, t[states, np.maximum(first, 0)], states).astype(np.int32)
        if not ok.all():
            break
    planners.close()
    model.close()


def test_mcts_subtree_strategy_on_stochastic_models_matches_reference():
    """step_strategy "subtree" on STOCHASTIC finite MDPs, open loop (round 4: the device refused before; the reference's
    step_by_subtree, tree_search/abstract.py:195-206, works on any env): multi-step episodes -- the live env is stepped,
    its generator advances, the device re-roots the kept tree (mp_uct_step_tree on the trees of mp_uct_plan_stochastic,
    also twice between two plans with receding_horizon = 2) -- plans, whole trees, generator states equal the unmodified
    reference's (tests/golden/stoch_subtree.npz, tests/golden/gen/make_golden_stoch_subtree.py)."""
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    from tests.test_gpu_variants import _agent_tree
    zz = np.load(os.path.join(REPO, "tests", "golden", "stoch_subtree.npz"))
    for name in [str(n) for n in zz["stoch_subtree/names"]]:
        p = "stoch_subtree/" + name
        cfg = mdp_from_golden(zz, p + "/mdp")
        c = dict(mode=cfg["mode"], transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"],
                 max_steps=cfg["max_steps"], state=int(zz[p + "/s0"]))
        if "next" in cfg:
            c["next"] = cfg["next"]
        env = FiniteMDPEnv(c)
        env.reset()
        env.seed(1000 + int(zz[p + "/seed"]))
        acfg = dict(__class__=UCT, budget=int(zz[p + "/budget"]), gamma=float(zz[p + "/gamma"]), temperature=float(zz[p + "/temperature"]),
                    horizon=int(zz[p + "/horizon"]), episodes=int(zz[p + "/episodes"]), step_strategy="subtree",
                    receding_horizon=int(zz[p + "/receding_horizon"]))
        if "pref" in name:
            acfg.update(prior_policy={"type": "preference", "action": 1, "ratio": 3},
                        rollout_policy={"type": "preference", "action": 1, "ratio": 3})
        agent = agent_factory(env, acfg)
        agent.seed(int(zz[p + "/seed"]))
        for step in range(int(zz[p + "/n_steps"])):
            q = "{}/step{}".format(p, step)
            assert env.mdp.state == int(zz[p + "/states"][step]), (name, step)
            np.testing.assert_array_equal(native.rng_state_from_generator(env.np_random), zz[q + "/env_rng"])
            plan = agent.plan(env.mdp.state)
            np.testing.assert_array_equal(plan, zz[q + "/plan"], err_msg=q)
            np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), zz[q + "/rng_after"], err_msg=q)
            if agent.planner.last is not None:                   # (a step that re-used the previous plan exports nothing new)
                root = agent.planner.root
                assert root.count == int(zz[q + "/root_count"]) and root.get_value() == float(zz[q + "/root_value"]), q
                t = _agent_tree(root)
                np.testing.assert_array_equal(t["parent"], zz[q + "/tree/parent"], err_msg=q)
                np.testing.assert_array_equal(t["action"], zz[q + "/tree/action"], err_msg=q)
                for f in ("count", "value"):
                    assert np.array_equal(t[f].astype(zz[q + "/tree/" + f].dtype), zz[q + "/tree/" + f]), (q, f)
            env.step(plan[0])


@pytest.mark.parametrize("golden_file", ["stoch_policies.npz", "many_actions.npz", "listing_order.npz"])
def test_per_state_policies_on_stochastic_models_match_reference(golden_file):
    """Round 4: restricted action sets (policies over get_available_actions(), mcts.py:59-97) and prior agents
    (mcts_with_prior.py:47-62) on STOCHASTIC finite MDPs, open and closed loop -- both refused before.  A node keeps the
    actions and priors of the state it was expanded in; plans, trees (stored priors included), env steps and generator
    states equal the unmodified reference's (tests/golden/stoch_policies.npz, make_golden_stoch_policies.py).
    many_actions.npz: 9 .. 40 actions -- the loop forms of the kernel (any number of actions) -- on deterministic tables too,
    which the planner routes there when their policies are per-state (make_golden_many_actions.py).
    listing_order.npz: environments that list their available actions in a non-ascending order, the restriction on the env
    object (OrderedMaskedFiniteMDPEnv, make_golden_listing_order.py): children and tie-breaks follow that order."""
    import json
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    from rl_agents_amd.envs.finite_mdp import MaskedFiniteMDPEnv
    from tests.test_gpu_variants import _agent_tree
    UCTP = "<class 'rl_agents_amd.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent'>"
    VI = "<class 'rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"
    zz = np.load(os.path.join(REPO, "tests", "golden", golden_file))
    for name in [str(n) for n in zz["stoch_policies/names"]]:
        p = "stoch_policies/" + name
        cfg = mdp_from_golden(zz, p + "/mdp")
        c = dict(mode=cfg["mode"], transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"],
                 max_steps=cfg["max_steps"], state=int(zz[p + "/s0"]))
        if "next" in cfg:
            c["next"] = cfg["next"]
        if (p + "/listing_order") in zz.files:
            from rl_agents_amd.envs import OrderedMaskedFiniteMDPEnv
            c["available"] = zz[p + "/available"]
            c["listing_order"] = [int(a) for a in zz[p + "/listing_order"]]
            env = OrderedMaskedFiniteMDPEnv(c)
        elif (p + "/available") in zz.files:
            c["available"] = zz[p + "/available"]
            env = MaskedFiniteMDPEnv(c)
        else:
            env = FiniteMDPEnv(c)
        env.reset()
        env.seed(1000 + int(zz[p + "/seed"]))
        assert np.array_equal(native.rng_state_from_generator(env.np_random), zz[p + "/env_rng"])
        acfg = dict(budget=int(zz[p + "/budget"]), gamma=float(zz[p + "/gamma"]), temperature=float(zz[p + "/temperature"]),
                    horizon=int(zz[p + "/horizon"]), episodes=int(zz[p + "/episodes"]), closed_loop=bool(zz[p + "/closed_loop"]),
                    prior_policy=json.loads(str(zz[p + "/prior_policy_json"])),
                    rollout_policy=json.loads(str(zz[p + "/rollout_policy_json"])))
        if bool(zz[p + "/with_prior_agent"]):
            agent = agent_factory(env, dict(acfg, __class__=UCTP, prior_agent=dict(__class__=VI, gamma=float(zz[p + "/prior_gamma"]),
                                                                                   temperature=float(zz[p + "/prior_temperature"]))))
            if not int(zz[p + "/prior_mask"]):   # the stock prior agent's own table: bit-exact for sparse models, 1e-12 for
                own = agent.prior_agent.policy_table()     # dense ones (matrix-core accumulation order, DESIGN.md parity bar)
                if cfg["mode"] == "sparse":
                    assert np.array_equal(own, zz[p + "/prior_table"]), name
                else:
                    np.testing.assert_allclose(own, zz[p + "/prior_table"], rtol=1e-9, atol=1e-12, err_msg=name)
            if int(zz[p + "/prior_mask"]) or cfg["mode"] != "sparse":
                # the planner's INPUT is the reference's table itself (zeroed entries / dense-VI rounding aside)
                table = np.array(zz[p + "/prior_table"])
                agent.prior_agent.policy_table = lambda table=table: table
        else:
            agent = agent_factory(env, dict(acfg, __class__=UCT))
        agent.seed(int(zz[p + "/seed"]))
        plan = agent.plan(int(zz[p + "/s0"]))
        np.testing.assert_array_equal([int(x) for x in plan], zz[p + "/plan"], err_msg=name)
        assert [isinstance(x, str) for x in plan] == list(zz[p + "/plan_is_obs"]), name
        assert agent.planner.env_steps == int(zz[p + "/env_steps"]), name
        np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), zz[p + "/rng_after"], err_msg=name)
        np.testing.assert_array_equal(native.rng_state_from_generator(env.np_random), zz[p + "/env_rng"])    # never stepped
        root = agent.planner.root
        assert root.count == int(zz[p + "/root_count"]) and root.get_value() == float(zz[p + "/root_value"]), name
        t = _agent_tree(root)
        np.testing.assert_array_equal(t["parent"], zz[p + "/tree/parent"], err_msg=name)
        np.testing.assert_array_equal(t["action"], zz[p + "/tree/action"], err_msg=name)
        for f in ("count", "value", "prior", "is_obs"):
            assert np.array_equal(t[f].astype(zz[p + "/tree/" + f].dtype), zz[p + "/tree/" + f]), (name, f)


@pytest.mark.parametrize("closed", [False, True], ids=["open", "closed"])
@pytest.mark.parametrize("mode,n_actions", [("sparse2", 3), ("sparse2", 5), ("sparse4", 7), ("sparse6", 4), ("stochastic", 2),
                                            ("stochastic", 8), ("sparse2", 12), ("stochastic", 9), ("sparse4", 40),
                                            ("deterministic", 12), ("deterministic", 33)])   # (> 8 actions: the loop forms)
def test_per_state_policies_on_stochastic_models_batch_vs_oracle(ctx, mode, n_actions, closed):
    """mp_uct_plan_stochastic_policy, seeded batches of 300 roots (ragged last wave), every |A| specialisation and record
    form: random availability tables, random per-state prior / rollout distributions over the listed actions, a TimeLimit,
    distinct planner and env generator records per root -- plans, values, env steps, generator records equal the oracle's."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    n, s_ = 300, 90
    if mode == "stochastic":
        cfg = generators.random_stochastic(s_, n_actions, seed=21 + n_actions, terminal_rate=0.05, concentration=0.1)
        kw, load = dict(), lambda: ctx.load_dense(cfg["transition"], cfg["reward"], cfg["terminal"])
    elif mode == "deterministic":   # a deterministic table through the same kernel (per-state policies over > 8 actions)
        cfg = generators.random_deterministic(s_, n_actions, seed=40 + n_actions, terminal_rate=0.05)
        kw, load = dict(), lambda: ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"],
                                                  done_rule="next" if n_actions % 2 else "source", max_steps=9)
    else:
        cfg = generators.random_sparse(s_, n_actions, int(mode[-1]), seed=30 + n_actions, terminal_rate=0.05)
        kw = dict(next_states=cfg["next"])
        load = lambda: ctx.load_sparse(cfg["transition"], cfg["next"], cfg["reward"], cfg["terminal"])
    model = load()
    if mode != "deterministic":     # (table models get their episode rules at load time)
        model.set_episode_rules("next" if n_actions % 2 else "source", 9)
    g = np.random.Generator(np.random.PCG64(100 + n_actions))
    avail = generators.random_available(s_, n_actions, seed=n_actions, rate=0.4)
    prior = np.where(avail, g.random((s_, n_actions)) + 0.1, 0.0)
    prior /= prior.sum(axis=1, keepdims=True)
    rollout = np.where(avail, g.random((s_, n_actions)) + 0.1, 0.0)
    rollout /= rollout.sum(axis=1, keepdims=True)
    lists = lambda t: dict(actions=[list(np.flatnonzero(avail[s])) for s in range(s_)],
                           p=[t[s, np.flatnonzero(avail[s])] for s in range(s_)])
    policy = ctx.load_policy(model, prior, rollout, listed=avail)
    s0 = g.integers(0, s_, size=n).astype(np.int32)
    steps0 = g.integers(0, 4, size=n).astype(np.int32)
    rng = np.stack([_fast_rng(i, 5) for i in range(n)])
    erng = np.stack([_fast_rng(i, 6) for i in range(n)])
    rng_ref = rng.copy()
    out = ctx.uct_plan_stochastic(model, s0, 40, 12, 0.9, 4.0, None, None, rng, env_rng_state=erng, closed_loop=closed,
                                  root_steps=steps0, max_plan_len=24, policy=policy)
    ref = oracle.uct_plan_stoch_batch(cfg["mode"], cfg["transition"], cfg["reward"], cfg["terminal"], s0, 40, 12, 0.9, 4.0,
                                      lists(prior), lists(rollout), rng_ref, erng, closed_loop=closed, steps0=steps0, max_steps=9,
                                      done_rule="next" if n_actions % 2 else "source", max_plan_len=24, n_threads=8, **kw)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["plan_len"], ref["plan_len"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    policy.close()
    model.close()


def _fast_rng(i, stream):
    from rl_agents_amd import native
    return native.seed_sequence_states([int(stream)], int(i), 1)[0]
