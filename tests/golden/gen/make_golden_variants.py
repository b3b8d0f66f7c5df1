#!/usr/bin/env python3
"""Golden vectors of the planner VARIANTS added in round 2, produced by the UNMODIFIED reference (same set-up as
make_golden.py; kept in a separate script so that the round-1 fixtures stay byte-identical):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_variants.py

  variants.npz
    closed/*       MCTSAgent with ``closed_loop: true`` (mcts.py:147, MCTSNode.get_child :267-273): trees with the
                   observation-keyed node layer, plans with the observation keys in them
    uct_masked/*   MCTSAgent on environments exposing ``get_available_actions`` (mcts.py:59-97 policies, expand :237-246)
    uct_prior_masked/*  MCTSWithPriorPolicyAgent on such environments (mcts_with_prior.py:56-62 renormalisation)
    opd_masked/*   DeterministicPlannerAgent on such environments (deterministic.py:32-35)
    robust/*       DiscreteRobustPlanner / RobustNode (agents/robust/robust.py:28-50) over the ndarray branch of
                   DeterministicNode.update (deterministic.py:54-59), driven through a joint-environment stand-in with the
                   gymnasium 5-tuple step (the reference's own JointEnv.step still returns the old 4-tuple, which
                   DeterministicNode.expand can no longer unpack)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference, the stubs and this repo)
from rl_agents.agents.common.factory import agent_factory  # noqa: E402
from rl_agents_amd.envs import MaskedFiniteMDPEnv, generators  # noqa: E402


def make_masked_env(cfg, available, state=0, steps=0):
    c = {k: v for k, v in cfg.items() if k != "original_shape"}
    c["state"] = int(state)
    c["available"] = np.asarray(available)
    env = MaskedFiniteMDPEnv(c)
    env.reset()
    env.steps = int(steps)
    return env


def keyed_tree(root, fields):
    """BFS listing like make_golden.bfs_tree, for trees whose child keys may be observation strings."""
    nodes, parents, keys, is_obs = [root], [-1], [-1], [False]
    i = 0
    while i < len(nodes):
        for k, c in nodes[i].children.items():
            nodes.append(c)
            parents.append(i)
            keys.append(int(k))
            is_obs.append(isinstance(k, str))
        i += 1
    out = dict(parent=np.asarray(parents, np.int32), action=np.asarray(keys, np.int32), is_obs=np.asarray(is_obs, bool))
    for name, fn, dt in fields:
        out[name] = np.asarray([fn(n) for n in nodes], dtype=dt)
    return out


UCT_FIELDS = [("count", lambda n: n.count, np.int64), ("value", lambda n: float(n.value), np.float64),
              ("prior", lambda n: float(n.prior), np.float64)]


def store_uct_case(store, p, cfg, env, agent, seed, s0, steps0, extra=None):
    agent.seed(seed)
    st0 = mg.rng_state(agent.planner.np_random)
    plan = agent.plan(s0)
    root = agent.planner.root
    pc = agent.planner.config
    mg.put_mdp(store, p + "/mdp", cfg)
    mg.put(store, p, dict(s0=s0, steps0=steps0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], episodes=pc["episodes"],
                          horizon=pc["horizon"], temperature=pc["temperature"], closed_loop=bool(pc["closed_loop"]),
                          plan=np.asarray([int(x) for x in plan], np.int32),
                          plan_is_obs=np.asarray([isinstance(x, str) for x in plan], bool),
                          root_count=root.count, root_value=float(root.value), env_steps=len(agent.planner.observations),
                          rng_before=st0, rng_after=mg.rng_state(agent.planner.np_random)))
    mg.put(store, p + "/tree", keyed_tree(root, UCT_FIELDS))
    if extra:
        mg.put(store, p, extra)


def golden_closed_loop(store):
    names = []
    large1 = {k: v for k, v in mg.load_env_config("large/env_1.json").items() if k != "max_steps"}
    large1_ms = mg.load_env_config("large/env_1.json")
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    hw_mid = generators.highway_shaped(5, 5, 20, seed=4)
    trap = mg.load_env_config("trap/env_1.json")
    pref = {"type": "preference", "action": 1, "ratio": 3}
    cases = [
        ("large1_b100", large1, 0, 0, dict(budget=100), [0, 1]),
        ("large1_b1000", large1, 0, 0, dict(budget=1000), [0]),
        ("large1_h30e33", large1, 7, 0, dict(budget=1000, horizon=30, episodes=33), [0, 5]),
        ("large1_maxsteps2", large1_ms, 0, 0, dict(budget=200), [9]),
        ("large1_pref", large1, 11, 0, dict(budget=300, prior_policy=pref, rollout_policy=pref), [4]),
        ("highway_small", hw, 0, 0, dict(budget=1000, horizon=30, episodes=33), [0, 1]),      # the shipped closed_loop.json
        ("highway_mid", hw_mid, 22, 0, dict(budget=1000, horizon=30, episodes=33), [2]),      # config: defaults + closed loop
        ("highway_default", hw, 13, 0, dict(), [0, 6]),
        ("trap", trap, 0, 0, dict(budget=200, temperature=3000), [0, 1]),
    ]
    for name, cfg, s0, steps0, acfg, seeds in cases:
        for seed in seeds:
            env = mg.make_env(cfg, state=s0, steps=steps0)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCT, closed_loop=True))
            prior_a, prior_p = agent.planner.prior_policy(env, None)
            roll_a, roll_p = agent.planner.rollout_policy(env, None)
            p = "closed/{}_seed{}".format(name, seed)
            store_uct_case(store, p, cfg, env, agent, seed, s0, steps0,
                           dict(prior_p=np.asarray(prior_p, np.float64), rollout_p=np.asarray(roll_p, np.float64)))
            names.append("{}_seed{}".format(name, seed))
    store["closed/names"] = np.asarray(names)
    # an episode of three consecutive plans (tree reset in between, one generator stream), first actions only
    env = mg.make_env(hw, state=5)
    agent = agent_factory(env, dict(__class__=mg.UCT, budget=300, horizon=12, episodes=25, closed_loop=True))
    agent.seed(3)
    firsts, states = [], []
    for _ in range(3):
        states.append(env.mdp.state)
        plan = agent.plan(env.mdp.state)
        firsts.append(int(plan[0]))
        env.step(plan[0])
    mg.put_mdp(store, "closed/episode_highway/mdp", hw)
    mg.put(store, "closed/episode_highway", dict(first_actions=np.asarray(firsts, np.int32), states=np.asarray(states, np.int32)))


def golden_uct_masked(store):
    names = []
    large1 = {k: v for k, v in mg.load_env_config("large/env_1.json").items() if k != "max_steps"}
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    hw_mid = generators.highway_shaped(5, 5, 20, seed=4)
    garnet = generators.random_deterministic(60, 4, seed=31, terminal_rate=0.05)
    av_large = generators.random_available(100, 5, seed=1, rate=0.35)
    av_hw, av_mid = generators.highway_available(hw), generators.highway_available(hw_mid)
    av_garnet = generators.random_available(60, 4, seed=2, rate=0.5)
    pref1 = {"type": "preference", "action": 1, "ratio": 3}
    pref3 = {"type": "preference", "action": 3, "ratio": 2.5}
    rnd = {"type": "random"}
    cases = [
        ("large1_b200", large1, av_large, 0, dict(budget=200), [0, 1, 2]),
        ("large1_b1000", large1, av_large, 4, dict(budget=1000), [0]),
        ("large1_pref", large1, av_large, 11, dict(budget=300, prior_policy=pref1, rollout_policy=pref3), [4, 5]),
        ("large1_prior_random", large1, av_large, 9, dict(budget=300, prior_policy=rnd), [3]),      # ignores availability
        ("large1_rollout_random", large1, av_large, 9, dict(budget=300, rollout_policy=rnd), [3]),
        ("highway_small", hw, av_hw, 0, dict(budget=1000, horizon=30, episodes=33), [0, 1]),
        ("highway_small_corner", hw, av_hw, 119 - 9, dict(budget=400), [2]),
        ("highway_mid", hw_mid, av_mid, 22, dict(budget=1000, horizon=30, episodes=33), [0]),
        ("highway_mid_closed", hw_mid, av_mid, 3, dict(budget=600, closed_loop=True), [1]),
        ("garnet_half", garnet, av_garnet, 5, dict(budget=400, gamma=0.9), [0, 7]),
    ]
    for name, cfg, avail, s0, acfg, seeds in cases:
        for seed in seeds:
            env = make_masked_env(cfg, avail, state=s0)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCT))
            p = "uct_masked/{}_seed{}".format(name, seed)
            pc = agent.planner.config
            store_uct_case(store, p, cfg, env, agent, seed, s0, 0,
                           dict(available=np.asarray(avail, bool), prior_policy=np.asarray(str(agent.config["prior_policy"])),
                                rollout_policy=np.asarray(str(agent.config["rollout_policy"]))))
            import json
            store[p + "/prior_policy_json"] = np.asarray(json.dumps(agent.config["prior_policy"]))
            store[p + "/rollout_policy_json"] = np.asarray(json.dumps(agent.config["rollout_policy"]))
            names.append("{}_seed{}".format(name, seed))
    store["uct_masked/names"] = np.asarray(names)
    # subtree re-use on a masked env
    env = make_masked_env(hw, av_hw, state=5)
    agent = agent_factory(env, dict(__class__=mg.UCT, budget=300, horizon=12, episodes=25, step_strategy="subtree"))
    agent.seed(11)
    p = "uct_masked/subtree_highway"
    mg.put_mdp(store, p + "/mdp", hw)
    store[p + "/available"] = np.asarray(av_hw, bool)
    pc = agent.planner.config
    store[p + "/rng_before"] = mg.rng_state(agent.planner.np_random)
    states = []
    for step in range(5):
        states.append(env.mdp.state)
        plan = agent.plan(env.mdp.state)
        root = agent.planner.root
        mg.put(store, "{}/step{}".format(p, step), dict(plan=np.asarray(plan, np.int32), root_count=root.count,
                                                        root_value=float(root.value),
                                                        rng_after=mg.rng_state(agent.planner.np_random)))
        mg.put(store, "{}/step{}/tree".format(p, step), keyed_tree(root, UCT_FIELDS))
        _, _, term, trunc, _ = env.step(plan[0])
        if term or trunc:
            break
    mg.put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), gamma=pc["gamma"],
                          episodes=pc["episodes"], horizon=pc["horizon"], temperature=pc["temperature"]))

    # MCTSWithPriorPolicyAgent on masked envs: the prior agent's distribution restricted to the available actions
    import prior_agents  # noqa: F401
    pnames = []
    for name, cfg, avail, s0, acfg, pcfg, seeds in [
            ("large1", large1, av_large, 0, dict(budget=300), dict(gamma=0.9, temperature=0.5), [0, 1]),
            ("highway_small", hw, av_hw, 0, dict(budget=1000, horizon=30, episodes=33), dict(gamma=0.95, temperature=0.3), [0])]:
        for seed in seeds:
            env = make_masked_env(cfg, avail, state=s0)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCTP, prior_agent=dict(pcfg, __class__=mg.PRIOR)))
            p = "uct_prior_masked/{}_seed{}".format(name, seed)
            store_uct_case(store, p, cfg, env, agent, seed, s0, 0,
                           dict(available=np.asarray(avail, bool), prior_table=np.array(agent.prior_agent.table),
                                q=np.array(agent.prior_agent.q), prior_gamma=pcfg["gamma"], prior_temperature=pcfg["temperature"]))
            pnames.append("{}_seed{}".format(name, seed))
    store["uct_prior_masked/names"] = np.asarray(pnames)


OPD_FIELDS = [("count", lambda n: n.count, np.int64), ("lower", lambda n: float(n.value_lower), np.float64),
              ("upper", lambda n: float(n.value_upper), np.float64), ("reward", lambda n: float(n.reward), np.float64),
              ("done", lambda n: bool(n.done), bool), ("depth", lambda n: n.depth, np.int32),
              ("obs", lambda n: -1 if n.observation is None else int(n.observation), np.int64)]


def golden_opd_masked(store):
    names = []
    large1 = mg.load_env_config("large/env_1.json")
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    grid = generators.gridworld()
    garnet = generators.random_deterministic(60, 4, seed=31, terminal_rate=0.05)
    cases = [
        ("large1_s0_b100", large1, generators.random_available(100, 5, seed=1, rate=0.35), 0, dict(budget=100, gamma=0.8), 0),
        ("large1_s7_b500", large1, generators.random_available(100, 5, seed=1, rate=0.35), 7, dict(budget=500, gamma=0.8), 1),
        ("grid_walls", grid, generators.random_available(100, 4, seed=3, rate=0.3), 0, dict(budget=400, gamma=0.9), 5),
        ("highway_small", hw, generators.highway_available(hw), 0, dict(budget=300, gamma=0.8), 0),
        ("highway_small_tr05", hw, generators.highway_available(hw), 41, dict(budget=300, gamma=0.9, terminal_reward=0.5), 3),
        ("garnet_half", garnet, generators.random_available(60, 4, seed=2, rate=0.5), 5, dict(budget=240, gamma=0.85), 2),
        ("garnet_single", garnet, generators.random_available(60, 4, seed=4, rate=0.9), 8, dict(budget=200, gamma=0.85), 2),
    ]
    for name, cfg, avail, s0, acfg, seed in cases:
        env = make_masked_env(cfg, avail, state=s0)
        agent = agent_factory(env, dict(acfg, __class__=mg.OPD))
        agent.seed(seed)
        st0 = mg.rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        root = agent.planner.root
        p = "opd_masked/" + name
        mg.put_mdp(store, p + "/mdp", cfg)
        mg.put(store, p, dict(s0=s0, seed=seed, budget=agent.config["budget"], gamma=agent.config["gamma"],
                              terminal_reward=agent.config["terminal_reward"], available=np.asarray(avail, bool),
                              plan=np.asarray(plan, np.int32), root_lower=float(root.value_lower),
                              root_upper=float(root.value_upper), root_count=root.count,
                              env_steps=len(agent.planner.observations), rng_before=st0,
                              rng_after=mg.rng_state(agent.planner.np_random)))
        mg.put(store, p + "/tree", keyed_tree(root, OPD_FIELDS))
        names.append(name)
    store["opd_masked/names"] = np.asarray(names)


def main():
    store = {}
    golden_closed_loop(store)
    golden_uct_masked(store)
    golden_opd_masked(store)
    try:
        import make_golden_robust
        make_golden_robust.golden_robust(store)
    except ImportError:
        pass
    path = os.path.join(mg.REPO, "tests", "golden", "variants.npz")
    np.savez_compressed(path, **store)
    print("{}: {} arrays, {:.1f} KB".format(path, len(store), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
