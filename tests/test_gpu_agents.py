"""The agent classes (same names / configs as the reference, "__class__" pointing at rl_agents_amd) reproduce the
reference's golden results end to end: agent_factory -> seed -> plan()/act() on a FiniteMDPEnv."""
import numpy as np
import pytest

from tests.helpers import mdp_from_golden

pytestmark = pytest.mark.gpu

UCT = "<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>"
OPD = "<class 'rl_agents_amd.agents.tree_search.deterministic.DeterministicPlannerAgent'>"
VI = "<class 'rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"
RVI = "<class 'rl_agents_amd.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent'>"


def _env(cfg, state=0, steps=0):
    from rl_agents_amd.envs import FiniteMDPEnv
    c = dict(mode=cfg["mode"], transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"],
             max_steps=cfg["max_steps"], state=int(state))
    if "next" in cfg:
        c["next"] = cfg["next"]
    env = FiniteMDPEnv(c)
    env.reset()
    env.steps = int(steps)
    return env


def test_mcts_agent_matches_reference_goldens(golden):
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["uct"]
    for name in [str(n) for n in z["uct/names"]]:
        p = "uct/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        agent_cfg = dict(__class__=UCT, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                         temperature=float(z[p + "/temperature"]), horizon=int(z[p + "/horizon"]),
                         episodes=int(z[p + "/episodes"]))
        if "pref" in name:
            agent_cfg.update(prior_policy={"type": "preference", "action": 1, "ratio": 3},
                             rollout_policy={"type": "preference", "action": 1, "ratio": 3})
        env = _env(cfg, state=int(z[p + "/s0"]), steps=int(z[p + "/steps0"]))
        agent = agent_factory(env, agent_cfg)
        assert agent.seed(int(z[p + "/seed"])) == [int(z[p + "/seed"])]
        plan = agent.plan(int(z[p + "/s0"]))
        np.testing.assert_array_equal(plan, z[p + "/plan"], err_msg=name)
        assert agent.planner.env_steps == int(z[p + "/env_steps"])
        root = agent.planner.root
        assert root.count == int(z[p + "/root_count"]) and root.get_value() == float(z[p + "/root_value"])
        # the host generator was advanced exactly like the reference's planner.np_random
        from rl_agents_amd import native
        np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), z[p + "/rng_after"])


def test_mcts_agent_sequence_of_acts(golden):
    """plan() after plan(): the stream continues, the tree is reset (step_strategy 'reset')."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    z = golden["uct"]
    cfg = mdp_from_golden(z, "uct/large1_b100_seed0/mdp")
    env = FiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"],
                            terminal=cfg["terminal"]))
    env.reset()
    agent = agent_factory(env, dict(__class__=UCT, budget=200))
    agent.seed(0)
    firsts = []
    for _ in range(3):
        a = agent.act(env.mdp.state)
        firsts.append(a)
        env.step(a)
    np.testing.assert_array_equal(firsts, z["uct/sequence_large1_b200_seed0/first_actions"])


def test_opd_agent_matches_reference_goldens(golden):
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["opd"]
    for name in [str(n) for n in z["opd/names"]]:
        p = "opd/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        env = _env(cfg, state=int(z[p + "/s0"]))
        agent = agent_factory(env, dict(__class__=OPD, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                                        terminal_reward=float(z[p + "/terminal_reward"])))
        agent.seed(int(z[p + "/seed"]))
        np.testing.assert_array_equal(agent.plan(int(z[p + "/s0"])), z[p + "/plan"], err_msg=name)
        root = agent.planner.root
        assert root.value_lower == float(z[p + "/root_lower"]) and root.value_upper == float(z[p + "/root_upper"])
        assert root.count == int(z[p + "/root_count"])
    # rewards outside [0, 1] raise like the reference (deterministic.py:46-47)
    from rl_agents_amd.envs import FiniteMDPEnv
    trap = FiniteMDPEnv(dict(mode="deterministic", transition=[[1, 2], [1, 1], [3, 4], [3, 3], [4, 4]],
                             reward=[[0, 0], [0, 0], [0, 0], [1, 1], [-1, -1]], terminal=[0, 1, 0, 1, 1]))
    trap.reset()
    with pytest.raises(ValueError):
        agent_factory(trap, dict(__class__=OPD, budget=20)).plan(0)


def test_vi_and_rvi_agents_match_reference_goldens(golden):
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["vi"]
    for name in ("large1_g09", "large1_default", "trap1", "sparse_s60", "highway_small"):
        p = "vi/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        env = _env(cfg)
        agent = agent_factory(env, dict(__class__=VI, gamma=float(z[p + "/gamma"]), iterations=int(z[p + "/iterations"])))
        assert np.array_equal(agent.state_action_value, z[p + "/Q"]) and agent.sweeps == int(z[p + "/sweeps"])
        np.testing.assert_array_equal([agent.act(s) for s in range(len(z[p + "/actions"]))], z[p + "/actions"])
        assert np.array_equal(agent.get_state_value(), z[p + "/V"])
    p = "rvi/large_pair_g09"
    models = [dict(mode="deterministic", transition=t.tolist(), reward=r.tolist())
              for t, r in zip(z[p + "/transitions"], z[p + "/rewards"])]
    agent = agent_factory(_env(mdp_from_golden(z, "vi/large1_g09/mdp")),
                          dict(__class__=RVI, gamma=0.9, iterations=200, models=models))
    np.testing.assert_array_equal([agent.act(s) for s in range(len(z[p + "/actions"]))], z[p + "/actions"])
    assert agent.models.uploads == 1                       # the per-act re-solve is served from the cache
    with pytest.raises(ValueError):
        agent_factory(_env(mdp_from_golden(z, "vi/large1_g09/mdp")), dict(__class__=RVI))
    with pytest.raises(TypeError):
        agent_factory(object(), dict(__class__=VI))


def test_non_finite_mdp_env_is_reconverted_each_act(golden):
    """An env that only offers to_finite_mdp() (highway-style): act() re-extracts, uploads only on change."""
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    from rl_agents_amd.envs import generators
    from rl_agents_amd.envs.finite_mdp import MDP

    class HighwayLike(object):
        def __init__(self):
            self.cfg = generators.highway_shaped(3, 4, 10, seed=3)
            self.state = 5
            self.unwrapped = self

        def to_finite_mdp(self):
            return MDP.from_config(dict(self.cfg, state=self.state))

    env = HighwayLike()
    agent = ValueIterationAgent(env, dict(gamma=0.95, iterations=200))
    z = golden["vi"]
    acts = []
    for s in (5, 17, 30):
        env.state = s
        acts.append(agent.act(None))
    np.testing.assert_array_equal(acts, z["vi/highway_small/actions"][[5, 17, 30]])
    assert agent.models.uploads == 1
    env.cfg = generators.highway_shaped(3, 4, 10, seed=4)
    agent.act(None)
    assert agent.models.uploads == 2


def test_plan_batch_api(golden):
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    env = FiniteMDPEnv(generators.highway_shaped(10, 10, 100, seed=0))
    env.reset()
    agent = MCTSAgent(env, dict(budget=1000, horizon=30, episodes=33))
    agent.seed(3)
    out = agent.plan_batch(np.arange(512) * 7 % 10000)
    assert out["plans"].shape == (512, 30) and (out["plans"][:, 0] >= 0).all()
    again = MCTSAgent(env, dict(budget=1000, horizon=30, episodes=33))
    again.seed(3)
    np.testing.assert_array_equal(again.plan_batch(np.arange(512) * 7 % 10000)["plans"], out["plans"])


def test_batched_evaluation_equals_sequential_episodes():
    """N lock-step episodes (one batched plan per step) reproduce N sequential agent/env loops action for action."""
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.batched_evaluation import BatchedEvaluation
    cfg = dict(generators.highway_shaped(3, 4, 10, seed=3), state=2, max_steps=9)
    env = FiniteMDPEnv(cfg)
    env.reset()
    agent_cfg = dict(budget=120, gamma=0.9)
    out = BatchedEvaluation(env, MCTSAgent(env, dict(agent_cfg)), num_episodes=6, sim_seed=40).run()
    for i in range(6):
        e = FiniteMDPEnv(cfg)
        e.reset()
        agent = MCTSAgent(e, dict(agent_cfg))
        agent.seed(40 + i)
        actions, total, done = [], 0.0, False
        while not done:
            a = agent.act(e.mdp.state)
            _, r, term, trunc, _ = e.step(a)
            actions.append(a)
            total += r
            done = term or trunc
        assert out["lengths"][i] == len(actions)
        np.testing.assert_array_equal(out["actions"][i, :len(actions)], actions)
        assert out["returns"][i] == pytest.approx(total, abs=1e-12)


@pytest.mark.parametrize("tag,agent_cfg", [("subtree_large1", dict(budget=150)),
                                           ("subtree_highway", dict(budget=300, horizon=12, episodes=25))])
def test_mcts_agent_subtree_strategy(golden, tag, agent_cfg):
    """An agent configured with step_strategy 'subtree' reproduces the reference's episode plan for plan."""
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["uct"]
    p = "uct/" + tag
    cfg = mdp_from_golden(z, p + "/mdp")
    env = _env(cfg, state=int(z[p + "/states"][0]))
    agent = agent_factory(env, dict(agent_cfg, __class__=UCT, step_strategy="subtree"))
    agent.seed(11)
    for step in range(int(z[p + "/n_steps"])):
        assert env.mdp.state == int(z[p + "/states"][step])
        plan = agent.plan(env.mdp.state)
        q = "{}/step{}".format(p, step)
        np.testing.assert_array_equal(plan, z[q + "/plan"], err_msg=q)
        assert agent.planner.root.count == int(z[q + "/root_count"])
        env.step(plan[0])


def test_plan_trajectory_follows_greedy_policy(golden):
    """ValueIterationAgent.plan_trajectory (value_iteration.py:84-96): greedy roll-out through the model."""
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    z = golden["vi"]
    cfg = mdp_from_golden(z, "vi/highway_small/mdp")
    env = _env(cfg)
    agent = ValueIterationAgent(env, dict(gamma=0.95, iterations=200))
    states, actions = agent.plan_trajectory(0, horizon=8)
    q = z["vi/highway_small/Q"]
    s = 0
    for i, (st, a) in enumerate(zip(states, actions)):
        assert st == s
        if a is None:
            assert cfg["terminal"][st] and i == len(states) - 1
            break
        assert a == np.argmax(q[st])
        s = int(cfg["transition"][st, a])
    assert len(states) == len(actions) <= 9


UCTP = "<class 'rl_agents_amd.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent'>"


def _bfs_nodes(root):
    out, i = [root], 0
    while i < len(out):
        out.extend(out[i].children[a] for a in sorted(out[i].children))
        i += 1
    return out


def test_mcts_with_prior_agent_matches_reference_goldens(golden):
    """MCTSWithPriorPolicyAgent with this package's ValueIterationAgent (Boltzmann over Q) as prior agent vs the
    reference MCTSWithPriorPolicyAgent driven by the same distribution: the VI Q table is solved on the device, the
    policy tables come out bit-identical, and so do plan, tree statistics, priors and generator state."""
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["uct_prior"]
    done = 0
    for name in [str(n) for n in z["uct_prior/names"]]:
        if "masked" in name or "two_policies" in name or "highway_mid" in name:
            continue        # tables the stock prior agent cannot produce (covered through the C ABI tests)
        p = "uct_prior/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        env = _env(cfg, state=int(z[p + "/s0"]))
        agent = agent_factory(env, dict(__class__=UCTP, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                                        temperature=float(z[p + "/temperature"]), horizon=int(z[p + "/horizon"]),
                                        episodes=int(z[p + "/episodes"]),
                                        prior_agent=dict(__class__=VI, gamma=float(z[p + "/prior_gamma"]),
                                                         temperature=float(z[p + "/prior_temperature"]))))
        assert np.array_equal(agent.prior_agent.policy_table(), z[p + "/prior_table"]), name
        agent.seed(int(z[p + "/seed"]))
        plan = agent.plan(int(z[p + "/s0"]))
        np.testing.assert_array_equal(plan, z[p + "/plan"], err_msg=name)
        assert agent.planner.env_steps == int(z[p + "/env_steps"])
        nodes = _bfs_nodes(agent.planner.root)
        np.testing.assert_array_equal([n.count for n in nodes], z[p + "/tree/count"])
        assert np.array_equal(np.array([n.get_value() for n in nodes]), z[p + "/tree/value"])
        assert np.array_equal(np.array([n.prior for n in nodes]), z[p + "/tree/prior"])
        np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), z[p + "/rng_after"])
        done += 1
    assert done >= 8


def test_mcts_with_prior_agent_subtree_episode(golden):
    """The reference's vi_prior.json shape: step_strategy 'subtree' with a prior agent, over a 5-step episode."""
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["uct_prior"]
    p = "uct_prior/subtree_highway"
    cfg = mdp_from_golden(z, p + "/mdp")
    env = _env(cfg, state=int(z[p + "/states"][0]))
    agent = agent_factory(env, dict(__class__=UCTP, budget=300, horizon=12, episodes=25, step_strategy="subtree",
                                    prior_agent=dict(__class__=VI, gamma=0.95, temperature=0.3)))
    agent.seed(11)
    for step in range(int(z[p + "/n_steps"])):
        assert env.mdp.state == int(z[p + "/states"][step])
        plan = agent.plan(env.mdp.state)
        q = "{}/step{}".format(p, step)
        np.testing.assert_array_equal(plan, z[q + "/plan"], err_msg=q)
        nodes = _bfs_nodes(agent.planner.root)
        np.testing.assert_array_equal([n.count for n in nodes], z[q + "/tree/count"])
        assert np.array_equal(np.array([n.prior for n in nodes][1:]), z[q + "/tree/prior"][1:])
        env.step(plan[0])


SAOPD = "<class 'rl_agents_amd.agents.tree_search.state_aware.StateAwarePlannerAgent'>"


def test_state_aware_agent_episodes_match_reference(golden):
    """StateAwarePlannerAgent through agent_factory: consecutive plan() calls of one agent along an episode equal the
    reference agent's (the planner's state values and state-node lists persist across plans), and the agent raises
    the reference's ValueError where every leaf gets pruned."""
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    z = golden["state_aware"]
    for name in [str(n) for n in z["sa/names"]]:
        p = "sa/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        env = _env(cfg, state=int(z[p + "/states"][0]))
        agent = agent_factory(env, dict(__class__=SAOPD, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                                        terminal_reward=float(z[p + "/terminal_reward"]), accuracy=float(z[p + "/accuracy"]),
                                        backup_aggregated_nodes=bool(z[p + "/backup_aggregated_nodes"]),
                                        prune_suboptimal_leaves=bool(z[p + "/prune_suboptimal_leaves"])))
        agent.seed(int(z[p + "/seed"]))
        raises_at = int(z[p + "/raises_at_step"]) if p + "/raises_at_step" in z.files else -1
        for step in range(int(z[p + "/n_steps"])):
            assert env.mdp.state == int(z[p + "/states"][step])
            if step == raises_at:
                with pytest.raises(ValueError):
                    agent.plan(env.mdp.state)
                break
            plan = agent.plan(env.mdp.state)
            q = "{}/step{}".format(p, step)
            np.testing.assert_array_equal(plan, z[q + "/plan"], err_msg=q)
            assert agent.planner.env_steps == int(z[q + "/env_steps"])
            np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), z[q + "/rng_after"])
            nodes = _bfs_nodes(agent.planner.root)
            np.testing.assert_array_equal([n.count for n in nodes], z[q + "/tree/count"])
            assert np.array_equal(np.array([n.value_lower for n in nodes]), z[q + "/tree/lower"])
            np.testing.assert_array_equal([bool(n.alive) for n in nodes], z[q + "/tree/is_leaf"])
            want = z[q + "/state_values"]
            seen = ~np.isnan(want)
            assert np.array_equal(agent.planner.state_values[seen], want[seen]), q
            env.step(plan[0])


def test_gamma_one_raises_like_the_reference(golden):
    """OPD / state-aware planning divide by 1 - gamma (deterministic.py:53, state_aware.py:83): ZeroDivisionError."""
    from rl_agents_amd.agents.common.factory import agent_factory
    cfg = mdp_from_golden(golden["opd"], "opd/grid_c1/mdp")
    for cls in (OPD, SAOPD):
        agent = agent_factory(_env(cfg), dict(__class__=cls, budget=40, gamma=1.0))
        with pytest.raises(ZeroDivisionError):
            agent.plan(0)


@pytest.mark.parametrize("kind", ["mcts_subtree", "state_aware"])
def test_batched_evaluation_with_planners_that_carry_state(kind):
    """Planners that keep state between plans (kept UCT trees, state-aware dictionaries) hold it per batch slot:
    N lock-step episodes still reproduce N sequential agents action for action, also after some episodes ended."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.batched_evaluation import BatchedEvaluation
    if kind == "mcts_subtree":
        cfg = dict(generators.highway_shaped(3, 4, 10, seed=3), state=2, max_steps=9)
        agent_cfg = dict(__class__=UCT, budget=120, gamma=0.9, step_strategy="subtree")
    else:
        cfg = dict(generators.gridworld(), state=0, max_steps=6)
        agent_cfg = dict(__class__=SAOPD, budget=100, gamma=0.9, prune_suboptimal_leaves=False)   # (pruning can empty the
        # leaves list, where the reference raises -- and so does a batched run for a live episode)
    cfg.pop("original_shape", None)
    env = FiniteMDPEnv(cfg)
    env.reset()
    n = 5
    starts = [2, 5, 7, 11, 13] if kind == "mcts_subtree" else [0, 12, 55, 37, 99]
    out = BatchedEvaluation(env, agent_factory(env, dict(agent_cfg)), num_episodes=n, sim_seed=70).run(initial_states=starts)
    for i in range(n):
        e = FiniteMDPEnv(dict(cfg, state=starts[i]))
        e.reset()
        agent = agent_factory(e, dict(agent_cfg))
        agent.seed(70 + i)
        actions, done = [], False
        while not done:
            a = agent.act(e.mdp.state)
            _, _, term, trunc, _ = e.step(a)
            actions.append(a)
            done = term or trunc
        assert out["lengths"][i] == len(actions), (kind, i)
        np.testing.assert_array_equal(out["actions"][i, :len(actions)], actions, err_msg=str((kind, i)))
    assert len(set(out["lengths"].tolist())) >= 1


def test_mcts_subtree_with_receding_horizon_descends_every_step():
    """step_strategy 'subtree' with receding_horizon 2: the agent steps its tree on EVERY act (abstract.py:70-82), so
    two levels are descended between two plans -- compared with the oracle driven by the same bookkeeping."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    env = FiniteMDPEnv(dict(mode="deterministic", transition=t, reward=r, terminal=term, state=5))
    env.reset()
    agent = agent_factory(env, dict(__class__=UCT, budget=300, horizon=12, episodes=25, step_strategy="subtree",
                                    receding_horizon=2))
    agent.seed(4)
    rng = native.rng_state_from_generator(agent.planner.np_random)
    p = np.ones(5) / 5
    tree, prev, remaining, replans = None, [], 0, 0
    for step in range(8):
        replan = remaining == 0 or len(prev) <= 1
        remaining = 1 if replan else remaining - 1
        tree = oracle.uct_reroot(tree, prev[0], 5) if (prev and tree is not None) else None
        if replan:
            out = oracle.uct_plan(t, r, term, env.mdp.state, 25, 12, 0.8, 2 / (1 - 0.8), p, p, rng, max_plan_len=12,
                                  init_tree=tree)
            tree, rng, want = out["tree"], out["rng_after"], [int(a) for a in out["plan"]]
            replans += 1
        else:
            want = prev[1:]
        got = agent.plan(env.mdp.state)
        assert got == want, (step, got, want)
        prev = want
        _, _, done, trunc, _ = env.step(want[0])
        if done or trunc:
            break
    assert replans >= 3
    np.testing.assert_array_equal(native.rng_state_from_generator(agent.planner.np_random), rng)


def test_tree_export_checks_ownership():
    """planner.root after another planner of the process has planned: never that planner's tree -- the first planner's own
    last tree, handed over to the host just before the shared device workspaces were reused (round 4: it used to raise);
    a fresh export of a tree that is no longer on the device still raises."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    cfg = generators.gridworld()
    env = FiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"]))
    env.reset()
    a = agent_factory(env, dict(__class__=UCT, budget=100))
    b = agent_factory(env, dict(__class__=OPD, budget=100))
    a.seed(0), b.seed(0)
    a.plan(0)
    assert a.planner.root.count > 0
    a.plan(0)
    count_a = a.planner.root.count
    a.planner._root = None                         # not exported yet when the second agent plans
    b.plan(0)
    assert b.planner.root.count == 101
    root_a = a.planner.root                        # a's OWN tree (kept when b took the workspaces over), not b's
    assert root_a.count == count_a and not hasattr(root_a, "value_lower")
    with pytest.raises(RuntimeError):
        a.planner.export_tree(0)                   # ... but it is not on the device any more
    a.plan(0)
    assert a.planner.root.count > 0
    assert b.planner.root.count == 101 and hasattr(b.planner.root, "value_lower")
    with pytest.raises(RuntimeError):
        b.planner.export_tree(0)


def test_batched_benchmark_equals_individual_evaluations(tmp_path):
    """Benchmark mode (scripts/experiments.py:85-116): environments x agents (a base agent varied over a key), every
    experiment one batched evaluation -- same episodes as running each experiment alone."""
    import json
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.batched_evaluation import BatchedEvaluation, batched_benchmark
    envs = []
    for i, cfg in enumerate((generators.highway_shaped(3, 4, 10, seed=3), generators.gridworld())):
        path = tmp_path / "env_{}.json".format(i)
        path.write_text(json.dumps(dict(id="finite-mdp-v0", import_module="finite_mdp", mode="deterministic",
                                        transition=cfg["transition"].tolist(), reward=cfg["reward"].tolist(),
                                        terminal=cfg["terminal"].astype(int).tolist(), max_steps=12)))
        envs.append(str(path))
    base = tmp_path / "agent.json"
    base.write_text(json.dumps(dict(__class__=UCT, gamma=0.9)))
    bench = dict(environments=envs, base_agent=str(base), key="budget", values=[60, 150],
                 agents=[dict(__class__=OPD, budget=80, gamma=0.8)])
    results = batched_benchmark(bench, episodes=9, seed=5)
    assert len(results) == 2 * 3
    # one experiment re-run alone
    cfg = json.loads(open(envs[1]).read())
    env = FiniteMDPEnv({k: cfg[k] for k in ("mode", "transition", "reward", "terminal", "max_steps")})
    env.reset()
    alone = BatchedEvaluation(env, agent_factory(env, dict(__class__=UCT, gamma=0.9, budget=150)), num_episodes=9, sim_seed=5).run()
    r = results[3 + 2]      # second environment; agents: the OPD config, then budget 60, budget 150
    np.testing.assert_array_equal(r["actions"], alone["actions"])
    assert np.array_equal(r["returns"], alone["returns"]) and r["episodes"] == 9
    assert results[0]["agent"]["__class__"] == OPD and results[1]["agent"]["budget"] == 60


def _tree_agents(z, name):
    from rl_agents_amd.agents.common.factory import agent_factory
    p = "trees/" + name
    cfg = mdp_from_golden(z, p + "/mdp")
    env = _env(cfg, state=int(z[p + "/s0"]))
    if bool(z[p + "/is_uct"]):
        agent_cfg = dict(__class__=UCT, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                         temperature=float(z[p + "/temperature"]), horizon=int(z[p + "/horizon"]), episodes=int(z[p + "/episodes"]))
    else:
        agent_cfg = dict(__class__=OPD, budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]))
    agent = agent_factory(env, agent_cfg)
    agent.seed(int(z[p + "/seed"]))
    return agent, env, p


def test_exported_tree_consumers_match_the_reference(golden):
    """f-3 end to end on the device: plan() on the GPU, then planner.root feeds get_obs_visits / get_trajectories /
    breadth_first_search / planner.get_visits with the reference's answers for the same plans (tree_tools.npz)."""
    from rl_agents_amd.agents.tree_search.abstract import Node
    z = golden["tree_tools"]
    for name in [str(n) for n in z["trees/names"]]:
        agent, env, p = _tree_agents(z, name)
        plan = agent.plan(int(z[p + "/s0"]))
        np.testing.assert_array_equal(plan, z[p + "/plan"], err_msg=name)
        root = agent.planner.root
        assert root.planner is agent.planner
        visits, _ = root.get_obs_visits(state=env)
        ref = dict(zip([str(k) for k in z[p + "/visit_keys"]], [int(c) for c in z[p + "/visit_counts"]]))
        assert dict(visits) == ref, name
        assert [n.count for n in root.get_trajectories(False, False)] == [int(c) for c in z[p + "/flat_counts"]]
        counts = list(Node.breadth_first_search(root, operator=lambda n, path: n.count))
        assert counts == [int(c) for c in z[p + "/bfs_counts"]]
        # (the optimistic planners derive it from their tree; MCTS -- rollouts leave no trace there -- replays the plan with the
        # visit counter armed, tests/test_gpu_visits.py)
        ref = dict(zip([str(k) for k in z[p + "/planner_visit_keys"]], [int(c) for c in z[p + "/planner_visit_counts"]]))
        assert dict(agent.planner.get_visits()) == ref, name


def test_a_second_agent_does_not_take_the_first_agents_tree(golden):
    """Tree ownership per planner: the planner about to lose the shared device workspaces exports its last tree first,
    so planner.root keeps answering (benchmark mode + display_tree); write_tree sends the plot to the writer."""
    import types
    import matplotlib
    matplotlib.use("Agg")
    z = golden["tree_tools"]
    first, env1, p1 = _tree_agents(z, "uct_highway_small")
    second, env2, p2 = _tree_agents(z, "opd_grid_c1")
    first.config["display_tree"] = True
    writer = types.SimpleNamespace(images=[])
    writer.add_image = lambda title, image, epoch: writer.images.append((title, image.shape, epoch))
    first.set_writer(writer)
    first.plan(int(z[p1 + "/s0"]))
    assert writer.images and writer.images[0][0] == "Expanded_tree" and writer.images[0][2] == 1
    first.planner._root = None                     # (write_tree exported it: drop the cache to exercise the hand-over)
    second.plan(int(z[p2 + "/s0"]))                # the second agent plans on the same context ...
    assert not first.planner.owns_device_tree()
    root = first.planner.root                      # ... and the first agent's tree is still there
    counts = [n.count for n in root.get_trajectories(False, False)]
    assert counts == [int(c) for c in z[p1 + "/flat_counts"]]
    third, _, p3 = _tree_agents(z, "uct_large1_b100")
    third.plan(int(z[p3 + "/s0"]))
    assert [n.count for n in second.planner.root.get_trajectories(False, False)] == [int(c) for c in z[p2 + "/flat_counts"]]
