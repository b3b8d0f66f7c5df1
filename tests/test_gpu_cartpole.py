"""UCT on the closed-form CartPole (BASELINE config C3) on the device.

Parity statement (round 5): BIT-EXACT, like every other planner path.  Arithmetic and random stream are the reference's, and
sin / cos of the pole angle are the host libm's own algorithm restated on the device (csrc/libm_sincos.hpp: glibc's dbl-64
sin / cos for |x| < 0.855, in the form -- FMA-contracted or not -- that reproduces this host's sin / cos; compared with the
host's libm on 10^7 angles below).  Rounds 1-4 used the device math library and a 99.5 % tolerance; that form is kept as
MP_CARTPOLE_SINCOS=device and still has to meet it.  The reference's own functional test must pass on the device planner."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def test_cartpole_goldens(ctx, golden):
    from rl_agents_amd.envs import CartPoleEnv
    z = golden["uct_cartpole"]
    model = ctx.load_cartpole(CartPoleEnv().cartpole_params())
    exact = 0
    names = [str(n) for n in z["cartpole/names"]]
    for name in names:
        p = "cartpole/" + name
        rng = np.array(z[p + "/rng_before"], dtype=np.uint64).reshape(1, 6)
        out = ctx.uct_plan(model, z[p + "/state0"].reshape(1, 4), int(z[p + "/episodes"]), int(z[p + "/horizon"]),
                           float(z[p + "/gamma"]), float(z[p + "/temperature"]), z[p + "/prior_p"], z[p + "/rollout_p"],
                           rng, root_steps=[int(z[p + "/steps0"])])
        n = int(out["plan_len"][0])
        same = (np.array_equal(out["plans"][0, :n], z[p + "/plan"]) and out["env_steps"][0] == int(z[p + "/env_steps"])
                and out["root_value"][0] == float(z[p + "/root_value"]) and np.array_equal(rng[0], z[p + "/rng_after"]))
        exact += bool(same)
    assert exact == len(names), "{} of {} golden CartPole plans reproduced exactly".format(exact, len(names))


def test_cartpole_batch_c3_vs_oracle(ctx):
    """C3 shape: budget 1000 as 20 episodes x horizon 50, 4096 roots ~ U(-0.05, 0.05)^4."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    params = CartPoleEnv().cartpole_params()
    model = ctx.load_cartpole(params)
    n = 4096
    x0 = np.random.Generator(np.random.PCG64(0)).uniform(-0.05, 0.05, size=(n, 4))
    steps0 = (np.arange(n) % 200).astype(np.int32)                 # TimeLimit truncation inside some horizons
    rng = np.stack([native.rng_state_from_generator(np.random.Generator(np.random.PCG64(np.random.SeedSequence(i))))
                    for i in range(n)])
    rng_ref = rng.copy()
    p = np.ones(2) / 2
    out = ctx.uct_plan(model, x0, 20, 50, 0.8, 10.0, p, p, rng, root_steps=steps0, max_plan_len=50)
    ref = oracle.uct_plan_batch(None, None, None, x0, 20, 50, 0.8, 10.0, p, p, rng_ref, steps0=steps0, max_plan_len=50,
                                n_threads=8, cartpole=params)
    assert native.libm_sincos_variant() in (1, 2), "neither restated form reproduces this host's libm sin / cos"
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"])
    np.testing.assert_array_equal(rng, ref["rng_after"])


@pytest.mark.parametrize("n", [1, 5, 300, 1100, 4096 + 37, 9000, 20000, 70000])
def test_cartpole_every_replication_factor_vs_oracle(ctx, n):
    """Round 6: a CartPole root is replicated over 64 / (roots per wave) lanes (the launch code picks roots per wave by batch
    size: 1 root x 64 lanes ... 16 x 4, then the one-lane form from 32 roots per wave on); the replicas share the generator's
    work by jump-ahead.  Every batch size class against the oracle, with skewed policies (a draw decides more than a coin)
    and step limits inside some horizons."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    params = CartPoleEnv().cartpole_params()
    model = ctx.load_cartpole(params)
    x0 = np.random.Generator(np.random.PCG64(n)).uniform(-0.08, 0.08, size=(n, 4))
    steps0 = ((np.arange(n) * 7) % 200).astype(np.int32)
    rng = native.seed_sequence_states((), 1000 * n, n)            # root i <- Generator(PCG64(SeedSequence(1000 n + i)))
    rng_ref = rng.copy()
    prior, roll = np.array([0.5, 0.5]), np.array([0.35, 0.65])
    out = ctx.uct_plan(model, x0, 12, 40, 0.9, 5.0, prior, roll, rng, root_steps=steps0, max_plan_len=8)
    assert ctx.last_kernel_variant() == "uct_cartpole"
    ref = oracle.uct_plan_batch(None, None, None, x0, 12, 40, 0.9, 5.0, prior, roll, rng_ref, steps0=steps0, max_plan_len=8,
                                n_threads=8, cartpole=params)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"])
    np.testing.assert_array_equal(rng, ref["rng_after"])


@pytest.mark.parametrize("lanes,rep,fastdiv", [(16, 0, 1), (16, 2, 1), (8, 3, 0), (4, 2, 1), (4, 4, 0), (2, 5, 1), (1, 6, 0), (64, 0, 1)])
def test_cartpole_forced_layouts_agree(ctx, monkeypatch, lanes, rep, fastdiv):
    """MP_UCT_LANES x MP_UCT_CART_REP (roots per wave x lanes per root, tools/cart_sweep.sh's knobs), with the short exact divisions
    and the hand-ordered step (default) or the IEEE divisions and the compiler-ordered step (MP_CART_FASTDIV=0): the same plans,
    statistics and generator records as the default layout."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    model = ctx.load_cartpole(CartPoleEnv().cartpole_params())
    n = 777
    x0 = np.random.Generator(np.random.PCG64(5)).uniform(-0.05, 0.05, size=(n, 4))
    rng0 = np.stack([native.rng_state_from_generator(np.random.Generator(np.random.PCG64(np.random.SeedSequence(i)))) for i in range(n)])
    p = np.ones(2) / 2

    def plan():
        rng = rng0.copy()
        out = ctx.uct_plan(model, x0, 20, 50, 0.8, 10.0, p, p, rng, max_plan_len=6)
        return out, rng
    base, rng_base = plan()
    monkeypatch.setenv("MP_UCT_LANES", str(lanes))
    monkeypatch.setenv("MP_UCT_CART_REP", str(rep))
    monkeypatch.setenv("MP_CART_FASTDIV", str(fastdiv))
    got, rng_got = plan()
    for k in ("plans", "plan_len", "env_steps", "root_value", "root_child_count", "root_child_value"):
        np.testing.assert_array_equal(got[k], base[k], err_msg=k)
    np.testing.assert_array_equal(rng_got, rng_base)


@pytest.mark.parametrize("horizon", [60, 120, 220, 400])
def test_cartpole_long_horizons(ctx, horizon):
    """Horizons whose per-lane path stack does not fit 64 KB of LDS with four wavefronts per workgroup (round 6's default): the
    launch raises the kernel's LDS limit and, beyond the CU's 160 KB, takes fewer waves per workgroup instead of refusing."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    params = CartPoleEnv().cartpole_params()
    model = ctx.load_cartpole(params)
    n = 200
    x0 = np.random.Generator(np.random.PCG64(horizon)).uniform(-0.02, 0.02, size=(n, 4))
    rng = native.seed_sequence_states((), 5 * horizon, n)
    rng_ref = rng.copy()
    p = np.ones(2) / 2
    out = ctx.uct_plan(model, x0, 8, horizon, 0.99, 5.0, p, p, rng, max_plan_len=8)
    ref = oracle.uct_plan_batch(None, None, None, x0, 8, horizon, 0.99, 5.0, p, p, rng_ref, max_plan_len=8, n_threads=8, cartpole=params)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    assert np.array_equal(out["root_value"], ref["root_value"])
    np.testing.assert_array_equal(rng, ref["rng_after"])


def test_cartpole_extreme_roots_and_wide_threshold(ctx):
    """What the fast forms of the rollout must hand back to the careful ones (round 6): roots whose angular velocity leaves the range
    the short exact division is proven for (1e200, inf, nan: the rollout is redone with IEEE divisions), roots handed over with the
    pole already far down and a threshold beyond the restated sin / cos range (the generic rollout tests the range per call)."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    p = np.ones(2) / 2
    n = 96
    x0 = np.random.Generator(np.random.PCG64(3)).uniform(-0.05, 0.05, size=(n, 4))
    x0[5, 3], x0[6, 3], x0[7, 3], x0[8, 3] = 1e200, -1e120, np.inf, np.nan      # velocities beyond the proven range
    x0[20, 2], x0[21, 2], x0[22, 2] = 1.0, -2.5, 0.81                           # angles outside the restated range
    x0[40, 1] = 1e300                                                            # (cart velocity: not in any division)
    for thr in (None, 1.2):
        params = CartPoleEnv().cartpole_params()
        if thr is not None:
            params = dict(params, theta_threshold=thr)
        model = ctx.load_cartpole(params)
        rng = native.seed_sequence_states((), 4242, n)
        rng_ref = rng.copy()
        with np.errstate(all="ignore"):
            out = ctx.uct_plan(model, x0, 10, 30, 0.9, 5.0, p, p, rng, max_plan_len=6)
            ref = oracle.uct_plan_batch(None, None, None, x0, 10, 30, 0.9, 5.0, p, p, rng_ref, max_plan_len=6, n_threads=4, cartpole=params)
        np.testing.assert_array_equal(out["plans"], ref["plans"])
        np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
        assert np.array_equal(out["root_value"], ref["root_value"], equal_nan=True)
        np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"])
        np.testing.assert_array_equal(rng, ref["rng_after"])
        model.close()


def test_cartpole_random_parameters_and_layouts(ctx, monkeypatch):
    """Forty random CartPole models (masses, length, force, gravity, tau, thresholds -- some outside the range the short exact
    division is proven for, some beyond the restated sin / cos range), random budgets, skewed policies, step limits and root
    layouts: plans, statistics and generator records of the device against the oracle."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    g = np.random.Generator(np.random.PCG64(2026))
    base = CartPoleEnv().cartpole_params()
    for case in range(40):
        params = dict(base)
        params["masspole"] = float(g.choice([0.1, 0.05, 1.0, 1e-20, 3.0]))          # (1e-20: below 2^-50 -> IEEE divisions)
        params["masscart"] = float(g.choice([1.0, 0.5, 10.0]))
        params["length"] = float(g.choice([0.5, 0.25, 2.0]))
        params["force_mag"] = float(g.choice([10.0, 1.0, 30.0]))
        params["gravity"] = float(g.choice([9.8, 1.62, 24.8]))
        params["tau"] = float(g.choice([0.02, 0.01, 0.05]))
        params["theta_threshold"] = float(g.choice([12 * 2 * np.pi / 360, 0.1, 0.5, 0.79, 0.9, 2.0]))
        params["x_threshold"] = float(g.choice([2.4, 0.5, 100.0]))
        params["max_steps"] = int(g.choice([0, 30, 200]))
        n = int(g.choice([1, 3, 17, 64, 300, 1500]))
        episodes, horizon = int(g.choice([1, 5, 20])), int(g.choice([1, 7, 16, 17, 50, 70]))
        x0 = g.uniform(-0.08, 0.08, size=(n, 4)) * np.array([1.0, 1.0, float(g.choice([0.5, 1.0, 4.0])), float(g.choice([1.0, 30.0]))])
        steps0 = g.integers(0, 25, size=n).astype(np.int32)
        pr = g.random(2) + 0.05
        pr /= pr.sum()
        ro = g.random(2) + 0.05
        ro /= ro.sum()
        temperature = float(g.choice([1.0, 10.0]))
        for knob in ("MP_UCT_LANES", "MP_UCT_CART_REP"):
            monkeypatch.delenv(knob, raising=False)
        if g.random() < 0.5:
            monkeypatch.setenv("MP_UCT_LANES", str(int(g.choice([1, 2, 4, 8, 16, 32, 64]))))
            monkeypatch.setenv("MP_UCT_CART_REP", str(int(g.integers(0, 7))))
        desc = dict(case=case, n=n, episodes=episodes, horizon=horizon, params=params, lanes=__import__("os").environ.get("MP_UCT_LANES"),
                    rep=__import__("os").environ.get("MP_UCT_CART_REP"))
        model = ctx.load_cartpole(params)
        rng = native.seed_sequence_states((), 7000 + case, n)
        rng_ref = rng.copy()
        with np.errstate(all="ignore"):
            out = ctx.uct_plan(model, x0, episodes, horizon, 0.9, temperature, pr, ro, rng, root_steps=steps0, max_plan_len=5)
            ref = oracle.uct_plan_batch(None, None, None, x0, episodes, horizon, 0.9, temperature, pr, ro, rng_ref, steps0=steps0,
                                        max_plan_len=5, n_threads=4, cartpole=params)
        np.testing.assert_array_equal(out["plans"], ref["plans"], err_msg=str(desc))
        np.testing.assert_array_equal(out["env_steps"], ref["env_steps"], err_msg=str(desc))
        assert np.array_equal(out["root_value"], ref["root_value"], equal_nan=True), desc
        np.testing.assert_array_equal(out["root_child_count"], ref["root_child_count"], err_msg=str(desc))
        np.testing.assert_array_equal(rng, ref["rng_after"], err_msg=str(desc))
        model.close()


def test_device_sincos_equals_host_libm_on_ten_million_angles(ctx):
    """The device's restated sin / cos (the form mp_libm_sincos_variant picked for this host) against the host libm's --
    math.sin / math.cos, what gymnasium's CartPole calls -- on 10^7 angles: the pole's range, the whole restated range
    |x| < 0.855469, the Taylor / table boundary at 0.126, tiny arguments and the grid points k / 128."""
    from rl_agents_amd import native
    variant = native.libm_sincos_variant()
    assert variant in (1, 2)
    g = np.random.Generator(np.random.PCG64(12))
    x = np.concatenate([g.uniform(-0.3, 0.3, 5_000_000), g.uniform(-0.855468, 0.855468, 3_000_000),
                        g.uniform(0.12, 0.132, 500_000) * g.choice([-1.0, 1.0], 500_000), g.uniform(-2e-7, 2e-7, 500_000),
                        g.normal(0, 0.05, 1_000_000 - 220), np.arange(-110, 110) / 128.0])
    assert x.size == 10_000_000
    s_dev, c_dev = ctx.selftest_sincos(x, variant)
    s_host, c_host = native.libm_sincos(x, 0)            # sin(), cos() of the C library this process runs on
    assert np.array_equal(s_dev, s_host) and np.array_equal(c_dev, c_host)
    import math
    probe = x[::5000]
    assert [math.sin(v) for v in probe] == s_dev[::5000].tolist() and [math.cos(v) for v in probe] == c_dev[::5000].tolist()
    # the BRANCH-FREE form the rollouts evaluate (round 6: every regime computed, the table paths' common work shared): same bits
    inside = np.abs(x) < 0.855468                         # (it has no range test: the rollouts use it under their own, uct.hip cart_flat)
    s_flat, c_flat = ctx.selftest_sincos(x[inside], variant + 2)
    assert inside.sum() > 9_999_000 and np.array_equal(s_flat, s_host[inside]) and np.array_equal(c_flat, c_host[inside])
    # outside the restated range (no pole angle gets there) the device math library answers: close, not claimed equal
    far = g.uniform(0.9, 6.0, 1000)
    s_far, c_far = ctx.selftest_sincos(far, variant)
    assert np.allclose(s_far, np.sin(far), rtol=0, atol=1e-15) and np.allclose(c_far, np.cos(far), rtol=0, atol=1e-15)


def test_cartpole_device_math_library_form_keeps_its_tolerance(ctx, monkeypatch):
    """MP_CARTPOLE_SINCOS=device: the rounds 1-4 form (device sincos) still returns the oracle's results for >= 99.5 % of roots."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    monkeypatch.setenv("MP_CARTPOLE_SINCOS", "device")
    params = CartPoleEnv().cartpole_params()
    model = ctx.load_cartpole(params)
    n = 1024
    x0 = np.random.Generator(np.random.PCG64(1)).uniform(-0.05, 0.05, size=(n, 4))
    rng = native.seed_sequence_states((), 0, n)
    rng_ref = rng.copy()
    p = np.ones(2) / 2
    out = ctx.uct_plan(model, x0, 20, 50, 0.8, 10.0, p, p, rng, max_plan_len=50)
    ref = oracle.uct_plan_batch(None, None, None, x0, 20, 50, 0.8, 10.0, p, p, rng_ref, max_plan_len=50, n_threads=8, cartpole=params)
    same = ((out["plans"] == ref["plans"]).all(axis=1) & (out["env_steps"] == ref["env_steps"]) & (out["root_value"] == ref["root_value"]))
    assert same.mean() >= 0.995


def test_reference_functional_test_on_device():
    """tests/agents/tree_search/test_mcts.py of the reference: MCTSAgent(budget=400, temperature=200) balances
    CartPole-v0 for the whole 200-step episode, choosing a valid action at every step."""
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.envs import CartPoleEnv
    env = CartPoleEnv()
    env.seed(0)
    state, _ = env.reset()
    agent = MCTSAgent(env, config=dict(budget=400, temperature=200, max_depth=10))
    agent.seed(0)
    done, steps = False, 0
    while not done:
        action = agent.act(state)
        assert action is not None
        state, reward, terminated, truncated, info = env.step(action)
        done = terminated or truncated
        steps += 1
    assert steps == env.max_episode_steps == 200
