"""PerEpisodeEvaluation: N environments that each own a finite MDP which changes at every step, one batched launch per
step -- equal to the unmodified reference's per-episode agents (tests/golden/per_episode.npz) and to N sequential
(environment, agent) loops of this package's single agents (trainer/evaluation.py:139-194 runs one per process)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E, T_STEPS = 6, 3


def _scheduled_envs(z):
    from rl_agents_amd.envs import ScheduledTableEnv
    envs = []
    for e in range(E):
        tables = [dict(mode="deterministic", transition=z["transition"][e, t], reward=z["reward"][e, t],
                       terminal=z["terminal"][e, t]) for t in range(T_STEPS)]
        envs.append(ScheduledTableEnv(tables, state=int(z["s0"][e])))
    return envs


@pytest.mark.parametrize("kind", ["vi", "uct", "opd"])
def test_golden_per_episode_evaluation(golden, kind):
    """Six episodes of the reference, each with its own agent object and a table replaced before every step: the first
    action of every plan / act of every step, from one batched launch per step."""
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    from rl_agents_amd.agents.tree_search.deterministic import DeterministicPlannerAgent
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.trainer.per_episode_evaluation import PerEpisodeEvaluation
    z = golden["per_episode"]
    envs = _scheduled_envs(z)
    if kind == "vi":
        agent = ValueIterationAgent(envs[0], dict(gamma=float(z["vi/gamma"]), iterations=int(z["vi/iterations"])))
    elif kind == "uct":
        agent = MCTSAgent(envs[0], dict(budget=int(z["uct/budget"]), gamma=float(z["uct/gamma"])))
    else:
        agent = DeterministicPlannerAgent(envs[0], dict(budget=int(z["opd/budget"]), gamma=float(z["opd/gamma"])))
    ev = PerEpisodeEvaluation(envs, agent, sim_seed=100, max_steps=T_STEPS)
    out = ev.run()
    for e in range(E):
        for t in range(T_STEPS):
            p = "{}/e{}/t{}".format(kind, e, t)
            want = int(z[p + "/action"]) if kind == "vi" else int(z[p + "/plan"][0])
            assert int(out["actions"][e, t]) == want, p
            assert int(z["{}/e{}/states".format(kind, e)][t]) >= 0
    if kind != "vi":
        total = sum(int(z["{}/e{}/t{}/env_steps_total".format(kind, e, T_STEPS - 1)]) for e in range(E))
        assert out["planner_env_steps"] == total
        for e in range(E):      # the generator of episode e ends where the reference's planner's does
            np.testing.assert_array_equal(ev.rng[e], z["{}/e{}/t{}/rng_after".format(kind, e, T_STEPS - 1)])
    assert out["uploads"] == E * T_STEPS          # every table changed at every step: each was sent exactly once
    ev.close()


def _sequential(envs, make_agent, sim_seed, max_steps):
    """N separate (environment, agent) loops, one agent object per episode as the reference runs them."""
    acts = np.full((len(envs), max_steps), -1, np.int32)
    returns = np.zeros(len(envs))
    for i, env in enumerate(envs):
        obs, _ = env.reset()
        agent = make_agent(env)
        if hasattr(agent, "seed"):
            agent.seed(sim_seed + i)
        for t in range(max_steps):
            a = int(agent.act(obs))
            obs, r, term, trunc, _ = env.step(a)
            acts[i, t] = a
            returns[i] += r
            if term or trunc:
                break
    return acts, returns


@pytest.mark.parametrize("kind", ["vi", "uct", "uct_random", "opd"])
def test_changing_highway_batch_equals_sequential_agents(kind):
    """highway-env's surface (restricted action sets listed IDLE first, the restriction on the env object) with a table
    re-drawn after every step, 24 environments: batch == 24 sequential agent loops, action for action."""
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    from rl_agents_amd.agents.tree_search.deterministic import DeterministicPlannerAgent
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.envs import ChangingHighwayEnv
    from rl_agents_amd.trainer.per_episode_evaluation import PerEpisodeEvaluation
    n, steps = 24, 6

    def envs():
        return [ChangingHighwayEnv(3, 4, 10, table_seed=500 + 20 * i, state=((i % 3) * 4 + (i % 4)) * 10,
                                   collision_rate=0.03 + 0.02 * (i % 4)) for i in range(n)]
    cfgs = dict(vi=(ValueIterationAgent, dict(gamma=0.95, iterations=100)),
                uct=(MCTSAgent, dict(budget=150, gamma=0.8)),
                uct_random=(MCTSAgent, dict(budget=150, gamma=0.8, prior_policy={"type": "random"},
                                            rollout_policy={"type": "random"})),
                opd=(DeterministicPlannerAgent, dict(budget=120, gamma=0.8)))
    cls, cfg = cfgs[kind]
    batch_envs = envs()
    ev = PerEpisodeEvaluation(batch_envs, cls(batch_envs[0], dict(cfg)), sim_seed=7, max_steps=steps)
    out = ev.run()
    acts, returns = _sequential(envs(), lambda env: cls(env, dict(cfg)), 7, steps)
    np.testing.assert_array_equal(out["actions"], acts)
    assert np.array_equal(out["returns"], returns)
    assert (out["actions"][:, 0] >= 0).all() and out["uploads"] >= n
    ev.close()


def test_unchanged_tables_are_not_sent_again():
    """Environments whose tables stay what they were are not uploaded again (the delta of a step is the episodes whose
    table changed)."""
    from rl_agents_amd.agents.tree_search.deterministic import DeterministicPlannerAgent
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.per_episode_evaluation import PerEpisodeEvaluation
    envs = []
    for i in range(8):
        cfg = {k: v for k, v in generators.highway_shaped(3, 4, 10, seed=i).items() if k != "original_shape"}
        cfg["state"] = 10 * i
        envs.append(FiniteMDPEnv(cfg))
    ev = PerEpisodeEvaluation(envs, DeterministicPlannerAgent(envs[0], dict(budget=60, gamma=0.8)), sim_seed=1, max_steps=5)
    out = ev.run()
    assert out["uploads"] == 8 and out["lengths"].max() >= 2
    ev.close()
