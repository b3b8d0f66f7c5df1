#!/usr/bin/env python3
"""Golden vectors of the round-3 additions, produced by the UNMODIFIED reference (same set-up as make_golden.py; a
separate script and file so that the earlier fixtures stay byte-identical):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_round3.py

  round3.npz
    rvi_v/*          RobustValueIterationAgent.get_state_value (robust_value_iteration.py:32-37) on the model lists of
                     vi.npz's rvi/* cases (inputs are read from there)
    robust_masked/*  DiscreteRobustPlanner over joint environments whose models restrict their available actions: the
                     joint env lists the union (agents/robust/robust.py:22-25), DeterministicNode.expand creates one
                     child per listed action (deterministic.py:32-35)
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import make_golden as mg  # noqa: E402  (sets up sys.path for the reference, the stubs and this repo)
import make_golden_robust as mgr  # noqa: E402
from make_golden_variants import make_masked_env  # noqa: E402
from rl_agents.agents.common.factory import agent_factory  # noqa: E402
from rl_agents.agents.robust.robust import DiscreteRobustPlanner  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402


def golden_rvi_v(store):
    z = np.load(os.path.join(mg.REPO, "tests", "golden", "vi.npz"))
    names = [str(n) for n in z["rvi/names"]]
    for name in names:
        p = "rvi/" + name
        mode = str(z[p + "/mode"])
        models = [dict(mode=mode, transition=t.tolist(), reward=r.tolist())
                  for t, r in zip(z[p + "/transitions"], z[p + "/rewards"])]
        env = mg.make_env(dict(mode="deterministic", transition=[[0]], reward=[[0.0]]))
        agent = agent_factory(env, dict(__class__=mg.RVI, models=models, gamma=float(z[p + "/gamma"]),
                                        iterations=int(z[p + "/iterations"])))
        v = np.array(agent.get_state_value(), dtype=np.float64)
        store["rvi_v/{}/V".format(name)] = v
    store["rvi_v/names"] = np.asarray(names)


def robust_keyed_tree(root, m):
    """make_golden_robust.robust_tree + the per-node child count (children are keyed by the listed actions)."""
    t = mgr.robust_tree(root, m)
    nodes, i = [root], 0
    while i < len(nodes):
        nodes.extend(nodes[i].children.values())
        i += 1
    t["n_children"] = np.asarray([len(n.children) for n in nodes], np.int32)
    return t


def golden_robust_masked(store):
    names = []
    large1 = {k: v for k, v in mg.load_env_config("large/env_1.json").items() if k != "max_steps"}
    large2 = {k: v for k, v in mg.load_env_config("large/env_2.json").items() if k != "max_steps"}
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    hw2 = generators.rewire(hw, 0.15, seed=10)
    g1 = generators.random_deterministic(60, 4, seed=31, terminal_rate=0.1)
    g2 = generators.random_deterministic(60, 4, seed=32, terminal_rate=0.1)
    g3 = generators.random_deterministic(60, 4, seed=33)
    av_l1 = generators.random_available(100, 5, seed=1, rate=0.5)
    av_l2 = generators.random_available(100, 5, seed=2, rate=0.5)
    av_hw = generators.highway_available(hw)
    av_g = [generators.random_available(60, 4, seed=s, rate=0.6) for s in (7, 8, 9)]
    cases = [
        # name, [(model config, availability table or None = the model env has no get_available_actions)], s0, planner config, seed
        ("large_pair_b100", [(large1, av_l1), (large2, av_l2)], 0, dict(budget=100, gamma=0.8), 0),
        ("large_pair_b500", [(large1, av_l1), (large2, av_l2)], 7, dict(budget=500, gamma=0.8), 1),
        ("large_same_mask", [(large1, av_l1), (large2, av_l1)], 42, dict(budget=300, gamma=0.9), 2),
        ("highway_pair", [(hw, av_hw), (hw2, av_hw)], 0, dict(budget=300, gamma=0.8), 0),
        ("highway_corner_tr05", [(hw, av_hw), (hw2, av_hw)], 119 - 9, dict(budget=300, gamma=0.9, terminal_reward=0.5), 3),
        ("garnet_triple", [(g1, av_g[0]), (g2, av_g[1]), (g3, av_g[2])], 5, dict(budget=240, gamma=0.85, terminal_reward=0.25), 2),
        ("garnet_one_unrestricted", [(g1, av_g[0]), (g2, None)], 9, dict(budget=200, gamma=0.85), 4),   # union = all actions
        ("single_model", [(large1, av_l2)], 3, dict(budget=200, gamma=0.8), 4),
    ]
    for name, models, s0, pcfg, seed in cases:
        envs = [mg.make_env(c, state=s0) if av is None else make_masked_env(c, av, state=s0) for c, av in models]
        joint = mgr.JointEnv5(envs)
        planner = DiscreteRobustPlanner(joint, dict(dict(terminal_reward=0), **pcfg))
        planner.seed(seed)
        st0 = mg.rng_state(planner.np_random)
        planner.step_by_reset()
        plan = planner.plan(joint, s0)
        root = planner.root
        m = len(models)
        p = "robust_masked/" + name
        s_, a_ = np.asarray(models[0][0]["reward"]).shape
        avail = np.stack([np.ones((s_, a_), bool) if av is None else np.asarray(av, bool) for _, av in models])
        for i, (c, _) in enumerate(models):
            mg.put_mdp(store, "{}/mdp{}".format(p, i), c)
        mg.put(store, p, dict(n_models=m, s0=s0, seed=seed, budget=planner.config["budget"], gamma=planner.config["gamma"],
                              terminal_reward=planner.config.get("terminal_reward", 0), available=avail,
                              has_mask=np.asarray([av is not None for _, av in models]),
                              plan=np.asarray(plan, np.int32), root_lower=float(np.min(root.value_lower)),
                              root_upper=float(np.min(root.value_upper)), root_count=root.count,
                              env_steps=len(planner.observations), rng_before=st0,
                              rng_after=mg.rng_state(planner.np_random)))
        mg.put(store, p + "/tree", robust_keyed_tree(root, m))
        names.append(name)
    store["robust_masked/names"] = np.asarray(names)


def golden_env_side(store):
    """The reference's agents planning DIRECTLY on an environment with highway-env's surface (HighwayLikeEnv: restriction
    on the env, listed IDLE-first, to_finite_mdp() without an availability table): what a device planner has to
    reproduce when it derives the table and the listing order from the env (device_model.availability_of)."""
    import prior_agents  # noqa: F401
    from make_golden_variants import OPD_FIELDS, UCT_FIELDS, keyed_tree, store_uct_case
    from rl_agents_amd.envs import HighwayLikeEnv
    small = generators.highway_shaped(3, 4, 10, seed=3)
    mid = generators.highway_shaped(5, 5, 20, seed=4)
    pref3 = {"type": "preference", "action": 3, "ratio": 2.5}
    rnd = {"type": "random"}
    unames = []
    ucases = [
        ("small_s0", small, 0, dict(budget=1000, horizon=30, episodes=33), [0, 1]),
        ("small_corner", small, 119 - 9, dict(budget=400), [2]),
        ("small_pref", small, 41, dict(budget=300, prior_policy=pref3, rollout_policy=pref3), [5]),
        ("small_rollout_random", small, 13, dict(budget=300, rollout_policy=rnd), [3]),
        ("mid_s22", mid, 22, dict(budget=1000, horizon=30, episodes=33), [0]),
        ("mid_closed", mid, 3, dict(budget=600, closed_loop=True), [1]),
    ]
    for name, table, s0, acfg, seeds in ucases:
        for seed in seeds:
            env = HighwayLikeEnv(table=table, state=s0)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCT))
            p = "env_side/uct/{}_seed{}".format(name, seed)
            store_uct_case(store, p, table, env, agent, seed, s0, 0,
                           dict(shape=np.asarray(table["original_shape"]), listing=np.asarray(env.get_available_actions())))
            store[p + "/prior_policy_json"] = np.asarray(json.dumps(agent.config["prior_policy"]))
            store[p + "/rollout_policy_json"] = np.asarray(json.dumps(agent.config["rollout_policy"]))
            assert env.state_index == s0 and env.steps == 0
            unames.append("{}_seed{}".format(name, seed))
    store["env_side/uct/names"] = np.asarray(unames)
    # receding horizon with tree re-use
    env = HighwayLikeEnv(table=small, state=5)
    agent = agent_factory(env, dict(__class__=mg.UCT, budget=300, horizon=12, episodes=25, step_strategy="subtree"))
    agent.seed(11)
    p = "env_side/uct_subtree"
    mg.put_mdp(store, p + "/mdp", small)
    states = []
    for step in range(5):
        states.append(env.state_index)
        plan = agent.plan(env.state_index)
        root = agent.planner.root
        mg.put(store, "{}/step{}".format(p, step), dict(plan=np.asarray(plan, np.int32), root_count=root.count,
                                                        root_value=float(root.value),
                                                        rng_after=mg.rng_state(agent.planner.np_random)))
        mg.put(store, "{}/step{}/tree".format(p, step), keyed_tree(root, UCT_FIELDS))
        _, _, term, trunc, _ = env.step(plan[0])
        if term or trunc:
            break
    mg.put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), shape=np.asarray(small["original_shape"])))
    # MCTSWithPriorPolicyAgent: the prior agent's distribution renormalised over the LISTED actions, in listing order
    pnames = []
    for name, table, s0, acfg, pcfg, seeds in [
            ("small", small, 0, dict(budget=1000, horizon=30, episodes=33), dict(gamma=0.95, temperature=0.3), [0]),
            ("mid", mid, 31, dict(budget=300), dict(gamma=0.9, temperature=0.5), [1])]:
        for seed in seeds:
            env = HighwayLikeEnv(table=table, state=s0)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCTP, prior_agent=dict(pcfg, __class__=mg.PRIOR)))
            p = "env_side/uct_prior/{}_seed{}".format(name, seed)
            store_uct_case(store, p, table, env, agent, seed, s0, 0,
                           dict(shape=np.asarray(table["original_shape"]), prior_table=np.array(agent.prior_agent.table),
                                prior_gamma=pcfg["gamma"], prior_temperature=pcfg["temperature"]))
            pnames.append("{}_seed{}".format(name, seed))
    store["env_side/uct_prior/names"] = np.asarray(pnames)
    onames = []
    for name, table, s0, acfg, seed in [
            ("small_s0", small, 0, dict(budget=300, gamma=0.8), 0),
            ("small_s41_tr05", small, 41, dict(budget=300, gamma=0.9, terminal_reward=0.5), 3),
            ("small_corner", small, 119 - 9, dict(budget=200, gamma=0.8), 1),
            ("mid_s22_b1000", mid, 22, dict(budget=1000, gamma=0.85), 2)]:
        env = HighwayLikeEnv(table=table, state=s0)
        agent = agent_factory(env, dict(acfg, __class__=mg.OPD))
        agent.seed(seed)
        st0 = mg.rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        root = agent.planner.root
        p = "env_side/opd/" + name
        mg.put_mdp(store, p + "/mdp", table)
        mg.put(store, p, dict(s0=s0, seed=seed, budget=agent.config["budget"], gamma=agent.config["gamma"],
                              terminal_reward=agent.config["terminal_reward"], shape=np.asarray(table["original_shape"]),
                              plan=np.asarray(plan, np.int32), root_lower=float(root.value_lower),
                              root_upper=float(root.value_upper), root_count=root.count,
                              env_steps=len(agent.planner.observations), rng_before=st0,
                              rng_after=mg.rng_state(agent.planner.np_random)))
        mg.put(store, p + "/tree", keyed_tree(root, OPD_FIELDS))
        onames.append(name)
    store["env_side/opd/names"] = np.asarray(onames)


def golden_state_aware_masked(store):
    """StateAwarePlannerAgent multi-plan episodes on environments that restrict their actions: table envs listing in
    ascending order (MaskedFiniteMDPEnv) and the highway-like env listing IDLE first (children in listing order)."""
    from rl_agents_amd.envs import HighwayLikeEnv
    large1 = mg.load_env_config("large/env_1.json")
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    grid = generators.gridworld()
    garnet = generators.random_deterministic(60, 4, seed=31, terminal_rate=0.05)
    cases = [
        # name, env factory, mdp cfg, availability (None: derived from the highway-like env), start, agent cfg, seed, plans
        ("grid_walls_b500", grid, generators.random_available(100, 4, seed=3, rate=0.3), 0, dict(budget=500, gamma=0.8), 0, 4),
        ("grid_accuracy", grid, generators.random_available(100, 4, seed=4, rate=0.4), 12, dict(budget=300, gamma=0.8, accuracy=0.05), 1, 3),
        ("large1_b500", large1, generators.random_available(100, 5, seed=1, rate=0.35), 0, dict(budget=500, gamma=0.8), 0, 3),
        ("large1_no_aggregation", large1, generators.random_available(100, 5, seed=1, rate=0.35), 7,
         dict(budget=300, gamma=0.8, backup_aggregated_nodes=False), 2, 3),
        ("highway_table", hw, generators.highway_available(hw), 0, dict(budget=300, gamma=0.8), 0, 4),
        ("garnet_half_tr05", garnet, generators.random_available(60, 4, seed=2, rate=0.5), 5,
         dict(budget=240, gamma=0.85, terminal_reward=0.5), 2, 3),
        ("highway_like_env", hw, None, 12, dict(budget=300, gamma=0.8), 0, 4),          # restriction on the env, IDLE first
        ("highway_like_env_corner", hw, None, 119 - 9, dict(budget=200, gamma=0.9), 3, 3),
    ]
    names = []
    for name, cfg, avail, s_start, agent_cfg, seed, n_plans in cases:
        if avail is None:
            env = HighwayLikeEnv(table=cfg, state=s_start)
            avail_store = generators.highway_available(cfg)

            def current(e=env):
                return e.state_index
        else:
            env = make_masked_env(cfg, avail, state=s_start)
            avail_store = np.asarray(avail, bool)

            def current(e=env):
                return e.mdp.state
        agent = agent_factory(env, dict(agent_cfg, __class__=mg.SAOPD))
        agent.seed(seed)
        p = "sa_masked/" + name
        mg.put_mdp(store, p + "/mdp", cfg)
        pc = agent.planner.config
        n_states = np.asarray(cfg["reward"]).shape[0]
        store[p + "/rng_before"] = mg.rng_state(agent.planner.np_random)
        store[p + "/available"] = avail_store
        store[p + "/listing_idle_first"] = np.asarray(avail is None)
        states = []
        for step in range(n_plans):
            states.append(current())
            try:
                plan = agent.plan(current())
            except ValueError as e:
                assert "empty" in str(e)
                store[p + "/raises_at_step"] = np.asarray(step)
                break
            planner = agent.planner
            leaves = set(id(n) for n in planner.leaves)
            tree = keyed_tree_sa(planner.root, leaves)
            sv = np.array([planner.state_values[str(s)] if str(s) in planner.state_values else np.nan for s in range(n_states)])
            q = "{}/step{}".format(p, step)
            mg.put(store, q, dict(plan=np.asarray(plan, np.int32), state_values=sv, n_leaves=len(planner.leaves),
                                  env_steps=len(planner.observations), rng_after=mg.rng_state(planner.np_random)))
            mg.put(store, q + "/tree", tree)
            _, _, term, trunc, _ = env.step(plan[0])
            if term or trunc:
                break
        mg.put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), seed=seed, budget=pc["budget"],
                              gamma=pc["gamma"], terminal_reward=agent.config["terminal_reward"], accuracy=pc["accuracy"],
                              backup_aggregated_nodes=pc["backup_aggregated_nodes"],
                              prune_suboptimal_leaves=pc["prune_suboptimal_leaves"]))
        names.append(name)
    store["sa_masked/names"] = np.asarray(names)


def keyed_tree_sa(root, leaves):
    from make_golden_variants import keyed_tree
    return keyed_tree(root, [("count", lambda n: n.count, np.int64), ("lower", lambda n: float(n.value_lower), np.float64),
                             ("reward", lambda n: float(n.reward), np.float64), ("done", lambda n: bool(n.done), bool),
                             ("depth", lambda n: n.depth, np.int32), ("obs", lambda n: int(n.observation), np.int64),
                             ("is_leaf", lambda n: id(n) in leaves, bool)])


def golden_uct_stochastic(store):
    """MCTSAgent on STOCHASTIC finite MDPs (dense `stochastic` and `sparse` modes), open and closed loop: the planner
    steps deep copies of the env, each copy carrying a copy of the env's own generator (common/factory.py:119-134), so
    every episode of a plan replays the same noise; closed loop keys the tree by the observed next states
    (mcts.py:147,267-273)."""
    from make_golden_variants import UCT_FIELDS, keyed_tree
    dense = generators.random_stochastic(30, 3, seed=5, terminal_rate=0.1)
    dense_b = generators.random_stochastic(50, 5, seed=6, concentration=0.05)       # few likely next states per (s, a)
    sparse = generators.random_sparse(60, 3, 2, seed=7, terminal_rate=0.1)
    sparse_b = generators.random_sparse(200, 5, 4, seed=8)
    sparse_ms = dict(sparse_b, max_steps=9)
    pref = {"type": "preference", "action": 1, "ratio": 3}
    cases = [
        ("dense_open", dense, 0, 0, dict(budget=300), [0, 1]),
        ("dense_closed", dense, 0, 0, dict(budget=300, closed_loop=True), [0, 1]),
        ("dense_b_closed_h30", dense_b, 7, 0, dict(budget=1000, horizon=30, episodes=33, closed_loop=True), [2]),
        ("dense_b_open_pref", dense_b, 11, 0, dict(budget=400, prior_policy=pref, rollout_policy=pref), [3]),
        ("sparse_open", sparse, 5, 0, dict(budget=400, gamma=0.9), [0]),
        ("sparse_closed", sparse, 5, 0, dict(budget=400, gamma=0.9, closed_loop=True), [0, 4]),
        ("sparse_b_closed", sparse_b, 17, 0, dict(budget=1000, horizon=30, episodes=33, closed_loop=True), [1]),
        ("sparse_maxsteps_closed", sparse_ms, 3, 4, dict(budget=300, closed_loop=True, temperature=3.0), [6]),
    ]
    names = []
    for name, cfg, s0, steps0, acfg, seeds in cases:
        for seed in seeds:
            env = mg.make_env(cfg, state=s0, steps=steps0)
            env.seed(1000 + seed)                                   # the env's own generator (copied with every clone)
            env_rng = mg.rng_state(env.np_random)
            agent = agent_factory(env, dict(acfg, __class__=mg.UCT))
            agent.seed(seed)
            st0 = mg.rng_state(agent.planner.np_random)
            plan = agent.plan(s0)
            root = agent.planner.root
            pc = agent.planner.config
            p = "uct_stoch/{}_seed{}".format(name, seed)
            mg.put_mdp(store, p + "/mdp", cfg)
            prior_a, prior_p = agent.planner.prior_policy(env, None)
            roll_a, roll_p = agent.planner.rollout_policy(env, None)
            mg.put(store, p, dict(s0=s0, steps0=steps0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], episodes=pc["episodes"],
                                  horizon=pc["horizon"], temperature=pc["temperature"], closed_loop=bool(pc["closed_loop"]),
                                  plan=np.asarray([int(x) for x in plan], np.int32),
                                  plan_is_obs=np.asarray([isinstance(x, str) for x in plan], bool),
                                  root_count=root.count, root_value=float(root.value), env_steps=len(agent.planner.observations),
                                  rng_before=st0, rng_after=mg.rng_state(agent.planner.np_random), env_rng=env_rng,
                                  env_rng_after=mg.rng_state(env.np_random),
                                  prior_p=np.asarray(prior_p, np.float64), rollout_p=np.asarray(roll_p, np.float64)))
            mg.put(store, p + "/tree", keyed_tree(root, UCT_FIELDS))
            assert env.mdp.state == s0 and np.array_equal(mg.rng_state(env.np_random), env_rng)   # the live env is untouched
            names.append("{}_seed{}".format(name, seed))
    store["uct_stoch/names"] = np.asarray(names)


def main():
    store = {}
    golden_rvi_v(store)
    golden_robust_masked(store)
    for extra in ("golden_env_side", "golden_state_aware_masked", "golden_uct_stochastic"):
        fn = globals().get(extra)
        if fn is not None:
            fn(store)
    path = os.path.join(mg.REPO, "tests", "golden", "round3.npz")
    np.savez_compressed(path, **store)
    print("{}: {} arrays, {:.1f} KB".format(path, len(store), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
