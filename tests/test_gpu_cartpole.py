"""UCT on the closed-form CartPole (BASELINE config C3) on the device.

Parity statement: arithmetic and random stream are the reference's, except sin/cos, which come from the
device math library (last-bit differences from glibc are possible).  Tolerance, as north_star allows for
stochastic UCT: at a fixed seed at least 99.5 % of roots must return the oracle's plan, env-step count and root
value exactly (observed: 100 %); the reference's own functional test must pass on the device planner."""
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
    same = ((out["plans"] == ref["plans"]).all(axis=1) & (out["env_steps"] == ref["env_steps"])
            & (out["root_value"] == ref["root_value"]))
    assert same.mean() >= 0.995, "only {:.4f} of roots identical to the oracle".format(same.mean())
    assert abs(out["root_value"].mean() - ref["root_value"].mean()) <= 1e-6


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
