#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference planners.

Run in the build container (the only place /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden.py

The reference (pure Python) is imported from /root/reference through the two test-only stub
packages in tests/golden/gen/stubs (``gymnasium`` names + ``numba.jit`` pass-through, SURVEY.md
Appendix C) and driven on this repo's FiniteMDPEnv (rl_agents_amd/envs/finite_mdp.py), which
restates the absent third-party ``finite_mdp`` package.  Nothing from the reference is copied:
the .npz files hold only inputs (MDP tables, seeds, configs) and the outputs the reference
computed for them.  The fixtures travel to the GPU box; this script and the reference do not
need to.

Files written (all small):
  vi.npz    value iteration / robust value iteration Q tables, sweep counts, greedy actions
  opd.npz   optimistic deterministic planner plans, root bounds and full trees
  uct.npz   MCTS/UCT plans, trees, env-step counts and PCG64 states before/after plan()
  uct_prior.npz  MCTSWithPriorPolicyAgent (per-state prior/rollout policies from a prior agent) plans and trees
  state_aware.npz  StateAwarePlannerAgent multi-plan episodes: plans, trees, leaves, state values
  misc.npz  OLOP.allocation table, numpy Generator draw sequences used to pin the PCG64 port
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, REPO, HERE]

import numpy as np  # noqa: E402

from rl_agents.agents.common.factory import agent_factory  # noqa: E402
from rl_agents.agents.tree_search.olop import OLOP  # noqa: E402
from rl_agents_amd.envs import CartPoleEnv, FiniteMDPEnv, generators  # noqa: E402

CFG = os.path.join(REF, "scripts", "configs", "FiniteMDPEnv")
VI = "<class 'rl_agents.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"
RVI = "<class 'rl_agents.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent'>"
OPD = "<class 'rl_agents.agents.tree_search.deterministic.DeterministicPlannerAgent'>"
UCT = "<class 'rl_agents.agents.tree_search.mcts.MCTSAgent'>"
UCTP = "<class 'rl_agents.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent'>"
PRIOR = "<class 'prior_agents.BoltzmannQAgent'>"
SAOPD = "<class 'rl_agents.agents.tree_search.state_aware.StateAwarePlannerAgent'>"


def load_env_config(rel):
    with open(os.path.join(CFG, rel)) as f:
        cfg = json.load(f)
    return {k: cfg[k] for k in ("mode", "transition", "reward", "terminal", "max_steps") if k in cfg}


def as_arrays(cfg):
    """Canonical arrays of a finite-MDP config (what the fixtures store as inputs)."""
    out = dict(mode=cfg["mode"], reward=np.asarray(cfg["reward"], dtype=np.float64))
    s = out["reward"].shape[0]
    if cfg["mode"] == "deterministic":
        out["transition"] = np.asarray(cfg["transition"], dtype=np.int64)
    else:
        out["transition"] = np.asarray(cfg["transition"], dtype=np.float64)
    if cfg["mode"] == "sparse":
        out["next"] = np.asarray(cfg["next"], dtype=np.int64)
    term = cfg.get("terminal")
    out["terminal"] = (np.zeros(s, bool) if term is None else np.asarray(term).reshape(s).astype(bool))
    out["max_steps"] = int(cfg.get("max_steps", 0) or 0)
    return out


def make_env(cfg, state=0, steps=0):
    c = {k: v for k, v in cfg.items() if k != "original_shape"}
    c["state"] = int(state)
    env = FiniteMDPEnv(c)
    env.reset()
    env.steps = int(steps)
    return env


def put(store, prefix, arrays):
    for k, v in arrays.items():
        store["{}/{}".format(prefix, k)] = np.asarray(v)


def put_mdp(store, prefix, cfg):
    a = as_arrays(cfg)
    store[prefix + "/mode"] = np.asarray(a.pop("mode"))
    put(store, prefix, a)


# ----------------------------------------------------------------------------- VI / RVI
def golden_vi():
    store, names = {}, []
    cases = [
        ("large1_g09", load_env_config("large/env_1.json"), dict(gamma=0.9, iterations=200)),
        ("large1_default", load_env_config("large/env_1.json"), dict()),
        ("large2_g095", load_env_config("large/env_2.json"), dict(gamma=0.95, iterations=500)),
        ("trap1", load_env_config("trap/env_1.json"), dict(gamma=0.9)),
        ("trap2", load_env_config("trap/env_2.json"), dict(gamma=0.9)),
        ("doors1", load_env_config("doors/env_1.json"), dict(gamma=0.9)),
        ("antivi1", load_env_config("anti_vi/env_1.json"), dict(gamma=1.0, iterations=10)),
        ("loop", load_env_config("env_loop.json"), dict(gamma=0.7, iterations=60)),
        ("grid_c1", generators.gridworld(), dict(gamma=0.8, iterations=100)),
        ("highway_small", generators.highway_shaped(3, 4, 10, seed=3), dict(gamma=0.95, iterations=200)),
        ("highway_mid", generators.highway_shaped(5, 5, 20, seed=4), dict(gamma=1.0, iterations=100)),
        ("dense_s40", generators.random_stochastic(40, 3, seed=5, terminal_rate=0.1), dict(gamma=0.9, iterations=300)),
        ("dense_s130", generators.random_stochastic(130, 5, seed=6), dict(gamma=0.95, iterations=40)),
        ("sparse_s60", generators.random_sparse(60, 3, 2, seed=7, terminal_rate=0.1), dict(gamma=0.9, iterations=300)),
        ("sparse_s300_b4", generators.random_sparse(300, 5, 4, seed=8), dict(gamma=0.99, iterations=50)),
    ]
    for name, cfg, agent_cfg in cases:
        env = make_env(cfg)
        agent = agent_factory(env, dict(agent_cfg, __class__=VI))
        q = np.array(agent.get_state_action_value(), dtype=np.float64)
        # sweep count actually run (value_iteration.py:65-73), recomputed with the same operator
        sweeps, value = 0, np.zeros(q.shape)
        for it in range(agent.config["iterations"]):
            nxt = agent.bellman_expectation(agent.best_action_value(value))
            sweeps += 1
            if np.allclose(value, nxt):
                break
            value = nxt
        assert np.array_equal(value, q)
        actions = np.array([np.argmax(q[s, :]) for s in range(q.shape[0])], dtype=np.int64)
        v = np.array(agent.get_state_value(), dtype=np.float64)
        p = "vi/" + name
        put_mdp(store, p + "/mdp", cfg)
        put(store, p, dict(gamma=agent.config["gamma"], iterations=agent.config["iterations"],
                           Q=q, V=v, sweeps=sweeps, actions=actions))
        names.append(name)
    store["vi/names"] = np.asarray(names)

    rnames = []
    with open(os.path.join(CFG, "large/agents/robust_value_iteration.json")) as f:
        large_models = json.load(f)["models"]
    with open(os.path.join(CFG, "trap/agents/robust_value_iteration.json")) as f:
        trap_models = json.load(f)["models"]
    with open(os.path.join(CFG, "doors/agents/robust_value_iteration.json")) as f:
        doors_models = json.load(f)["models"]
    with open(os.path.join(CFG, "anti_vi/agents/robust_value_iteration.json")) as f:
        antivi_models = json.load(f)["models"]
    hw = generators.highway_shaped(4, 4, 12, seed=9)
    hw2 = generators.rewire(hw, 0.1, seed=10)
    d1 = generators.random_stochastic(30, 3, seed=11)
    d2 = generators.random_stochastic(30, 3, seed=12)
    d2["reward"] = d1["reward"] * 0.9

    def listify(m):
        return {k: (np.asarray(v).tolist() if not isinstance(v, str) else v) for k, v in m.items()
                if k in ("mode", "transition", "reward", "terminal")}
    rcases = [
        ("large_pair_g09", large_models, dict(gamma=0.9, iterations=200)),
        ("large_pair_cfg", large_models, dict(gamma=1.0, iterations=2)),
        ("trap_pair", trap_models, dict(gamma=0.9)),
        ("doors_pair", doors_models, dict(gamma=0.9)),
        ("antivi_pair", antivi_models, dict(gamma=1.0, iterations=10)),
        ("highway_pair", [listify(hw), listify(hw2)], dict(gamma=0.95, iterations=150)),
        ("dense_pair", [listify(d1), listify(d2)], dict(gamma=0.9, iterations=100)),
    ]
    for name, models, agent_cfg in rcases:
        env = make_env(dict(mode="deterministic", transition=[[0]], reward=[[0.0]]))
        agent = agent_factory(env, dict(agent_cfg, __class__=RVI, models=models))
        q = np.array(agent.get_state_action_value(), dtype=np.float64)
        sweeps, value = 0, np.zeros(q.shape)
        for it in range(agent.config["iterations"]):
            nxt = agent.worst_case(agent.bellman_expectation(agent.best_action_value(value)))
            sweeps += 1
            if np.allclose(value, nxt):
                break
            value = nxt
        assert np.array_equal(value, q)
        actions = np.array([agent.act(s) for s in range(min(q.shape[0], 50))], dtype=np.int64)
        p = "rvi/" + name
        store[p + "/mode"] = np.asarray(models[0]["mode"])
        dt = np.int64 if models[0]["mode"] == "deterministic" else np.float64
        put(store, p, dict(gamma=agent.config["gamma"], iterations=agent.config["iterations"],
                           transitions=np.array([m["transition"] for m in models], dtype=dt),
                           rewards=np.array([m["reward"] for m in models], dtype=np.float64),
                           Q=q, sweeps=sweeps, actions=actions))
        rnames.append(name)
    store["rvi/names"] = np.asarray(rnames)
    return store


# ----------------------------------------------------------------------------- trees
def bfs_tree(root, fields):
    """Canonical BFS listing (children in dict = creation order) of a reference tree."""
    nodes, parents, actions = [root], [-1], [-1]
    i = 0
    while i < len(nodes):
        for a, c in nodes[i].children.items():
            nodes.append(c)
            parents.append(i)
            actions.append(int(a))
        i += 1
    out = dict(parent=np.asarray(parents, np.int32), action=np.asarray(actions, np.int32))
    for name, fn, dt in fields:
        out[name] = np.asarray([fn(n) for n in nodes], dtype=dt)
    return out


def rng_state(gen):
    st = gen.bit_generator.state
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.asarray([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)


def golden_opd():
    store, names = {}, []
    large1 = load_env_config("large/env_1.json")
    large2 = load_env_config("large/env_2.json")
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    grid = generators.gridworld()
    loop = load_env_config("env_loop.json")
    cases = [
        ("large1_s0_b100", large1, 0, dict(budget=100, gamma=0.8), 0),
        ("large1_s7_b100", large1, 7, dict(budget=100, gamma=0.8), 0),
        ("large1_s0_b500", large1, 0, dict(budget=500, gamma=0.8), 0),
        ("large1_s7_b500", large1, 7, dict(budget=500, gamma=0.8), 0),
        ("large2_s42_b1000_g095", large2, 42, dict(budget=1000, gamma=0.95), 1),
        ("large1_s3_b37", large1, 3, dict(budget=37, gamma=0.5), 2),
        ("grid_c1", grid, 0, dict(budget=100, gamma=0.8), 0),
        ("grid_corner_b400", grid, 99, dict(budget=400, gamma=0.9), 5),
        ("highway_small_tr0", hw, 0, dict(budget=300, gamma=0.8), 0),
        ("highway_small_tr05", hw, 41, dict(budget=300, gamma=0.9, terminal_reward=0.5), 3),
        ("loop_b60", loop, 0, dict(budget=60, gamma=0.7), 4),
    ]
    for name, cfg, s0, agent_cfg, seed in cases:
        env = make_env(cfg, state=s0)
        agent = agent_factory(env, dict(agent_cfg, __class__=OPD))
        agent.seed(seed)
        st0 = rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        root = agent.planner.root
        tree = bfs_tree(root, [("count", lambda n: n.count, np.int64),
                               ("lower", lambda n: float(n.value_lower), np.float64),
                               ("upper", lambda n: float(n.value_upper), np.float64),
                               ("reward", lambda n: float(n.reward), np.float64),
                               ("done", lambda n: bool(n.done), bool),
                               ("depth", lambda n: n.depth, np.int32),
                               ("obs", lambda n: -1 if n.observation is None else int(n.observation), np.int64)])
        p = "opd/" + name
        put_mdp(store, p + "/mdp", cfg)
        put(store, p, dict(s0=s0, seed=seed, budget=agent.config["budget"], gamma=agent.config["gamma"],
                           terminal_reward=agent.config["terminal_reward"],
                           plan=np.asarray(plan, np.int32), root_lower=float(root.value_lower),
                           root_upper=float(root.value_upper), root_count=root.count,
                           env_steps=len(agent.planner.observations),
                           rng_before=st0, rng_after=rng_state(agent.planner.np_random)))
        put(store, p + "/tree", tree)
        names.append(name)
    store["opd/names"] = np.asarray(names)
    # error behaviour: rewards outside [0, 1] -> ValueError (deterministic.py:46-47)
    env = make_env(load_env_config("trap/env_1.json"))
    agent = agent_factory(env, dict(__class__=OPD, budget=20, gamma=0.8))
    try:
        agent.plan(0)
        raised = False
    except ValueError:
        raised = True
    store["opd/trap_raises_valueerror"] = np.asarray(raised)
    return store


def golden_uct():
    store, names = {}, []
    large1 = load_env_config("large/env_1.json")           # max_steps = 2 in the config
    large1_nolimit = {k: v for k, v in large1.items() if k != "max_steps"}
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    hw_mid = generators.highway_shaped(5, 5, 20, seed=4)
    trap = load_env_config("trap/env_1.json")
    doors = load_env_config("doors/env_1.json")
    antivi = load_env_config("anti_vi/env_1.json")         # max_steps = 10
    pref = {"type": "preference", "action": 1, "ratio": 3}
    cases = [
        # name, mdp cfg, s0, env.steps at plan time, agent cfg, seeds
        ("large1_b100", large1_nolimit, 0, 0, dict(budget=100), [0, 1, 2]),
        ("large1_b1000", large1_nolimit, 0, 0, dict(budget=1000), [0, 1]),
        ("large1_h30e33", large1_nolimit, 7, 0, dict(budget=1000, horizon=30, episodes=33), [0, 5]),
        ("large1_g095_b400", large1_nolimit, 3, 0, dict(budget=400, gamma=0.95), [3]),
        ("large1_temp200", large1_nolimit, 0, 0, dict(budget=400, temperature=200), [7]),
        ("large1_maxsteps2", large1, 0, 0, dict(budget=200), [0, 9]),
        ("large1_pref", large1_nolimit, 11, 0, dict(budget=300, prior_policy=pref, rollout_policy=pref), [4]),
        ("large1_random_policy", large1_nolimit, 11, 0,
         dict(budget=300, prior_policy={"type": "random"}, rollout_policy={"type": "random"}), [4]),
        ("highway_small", hw, 0, 0, dict(budget=1000, horizon=30, episodes=33), [0, 1, 2]),
        ("highway_small_default", hw, 13, 0, dict(budget=1000), [0, 6]),
        ("highway_mid", hw_mid, 22, 0, dict(budget=1000, horizon=30, episodes=33), [0, 1]),
        ("trap", trap, 0, 0, dict(budget=200, temperature=3000), [0, 1, 2, 3]),
        ("doors", doors, 0, 0, dict(budget=400, temperature=3000), [0, 1]),
        ("antivi_steps4", antivi, 0, 4, dict(budget=300), [0, 1]),
    ]
    for name, cfg, s0, steps0, agent_cfg, seeds in cases:
        for seed in seeds:
            env = make_env(cfg, state=s0, steps=steps0)
            agent = agent_factory(env, dict(agent_cfg, __class__=UCT))
            agent.seed(seed)
            st0 = rng_state(agent.planner.np_random)
            plan = agent.plan(s0)
            root = agent.planner.root
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64),
                                   ("value", lambda n: float(n.value), np.float64),
                                   ("prior", lambda n: float(n.prior), np.float64)])
            pc = agent.planner.config
            prior_a, prior_p = agent.planner.prior_policy(env, None)
            roll_a, roll_p = agent.planner.rollout_policy(env, None)
            p = "uct/{}_seed{}".format(name, seed)
            put_mdp(store, p + "/mdp", cfg)
            put(store, p, dict(s0=s0, steps0=steps0, seed=seed, budget=pc["budget"], gamma=pc["gamma"],
                               episodes=pc["episodes"], horizon=pc["horizon"], temperature=pc["temperature"],
                               prior_actions=np.asarray(prior_a, np.int32), prior_p=np.asarray(prior_p, np.float64),
                               rollout_actions=np.asarray(roll_a, np.int32), rollout_p=np.asarray(roll_p, np.float64),
                               plan=np.asarray(plan, np.int32), root_count=root.count, root_value=float(root.value),
                               env_steps=len(agent.planner.observations),
                               rng_before=st0, rng_after=rng_state(agent.planner.np_random)))
            put(store, p + "/tree", tree)
            names.append("{}_seed{}".format(name, seed))
    store["uct/names"] = np.asarray(names)

    # two consecutive plan() calls on one agent (RNG stream continues, tree is reset): large1, seed 0
    env = make_env(large1_nolimit, state=0)
    agent = agent_factory(env, dict(__class__=UCT, budget=200))
    agent.seed(0)
    plans = []
    for _ in range(3):
        plans.append(list(agent.plan(env.mdp.state))[:4] + [-1] * 4)
        env.step(plans[-1][0])
    store["uct/sequence_large1_b200_seed0/first_actions"] = np.asarray([p[0] for p in plans], np.int32)
    store["uct/sequence_large1_b200_seed0/states"] = np.asarray([0], np.int32)

    # step_strategy "subtree" (abstract.py:172-206): the tree is re-rooted at the executed action between plans
    for tag, cfg, s_start, acfg in (("subtree_large1", large1_nolimit, 0, dict(budget=150, step_strategy="subtree")),
                                    ("subtree_highway", hw, 5, dict(budget=300, horizon=12, episodes=25,
                                                                    step_strategy="subtree"))):
        env = make_env(cfg, state=s_start)
        agent = agent_factory(env, dict(acfg, __class__=UCT))
        agent.seed(11)
        p = "uct/" + tag
        put_mdp(store, p + "/mdp", cfg)
        pc = agent.planner.config
        store[p + "/rng_before"] = rng_state(agent.planner.np_random)
        states = []
        for step in range(5):
            states.append(env.mdp.state)
            plan = agent.plan(env.mdp.state)
            root = agent.planner.root
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64), ("value", lambda n: float(n.value), np.float64)])
            put(store, "{}/step{}".format(p, step), dict(plan=np.asarray(plan, np.int32), root_count=root.count,
                                                         root_value=float(root.value),
                                                         rng_after=rng_state(agent.planner.np_random)))
            put(store, "{}/step{}/tree".format(p, step), tree)
            _, _, term, trunc, _ = env.step(plan[0])
            if term or trunc:
                break
        put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), gamma=pc["gamma"],
                           episodes=pc["episodes"], horizon=pc["horizon"], temperature=pc["temperature"]))
    return store


def golden_uct_prior():
    """MCTSWithPriorPolicyAgent (mcts_with_prior.py): prior and rollout policies that depend on the state."""
    store, names = {}, []
    large1 = {k: v for k, v in load_env_config("large/env_1.json").items() if k != "max_steps"}
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    hw_mid = generators.highway_shaped(5, 5, 20, seed=4)
    doors = load_env_config("doors/env_1.json")
    cases = [
        # name, mdp cfg, s0, agent cfg, prior-agent cfg, separate rollout temperature (None = same policy), seeds
        ("large1_b200", large1, 0, dict(budget=200), dict(gamma=0.9, temperature=0.5), None, [0, 1, 2]),
        ("large1_b1000_t02", large1, 7, dict(budget=1000), dict(gamma=0.9, temperature=0.2), None, [0, 3]),
        ("large1_h30e33", large1, 11, dict(budget=1000, horizon=30, episodes=33), dict(gamma=0.8, temperature=1.0), None, [5]),
        ("large1_masked", large1, 4, dict(budget=400), dict(gamma=0.9, temperature=0.5, mask=7), None, [0, 1]),
        ("large1_two_policies", large1, 9, dict(budget=400, temperature=40), dict(gamma=0.9, temperature=0.3), 2.0, [2, 8]),
        ("highway_small", hw, 0, dict(budget=1000, horizon=30, episodes=33), dict(gamma=0.95, temperature=0.3), None, [0, 1]),
        ("highway_mid", hw_mid, 22, dict(budget=1000, horizon=30, episodes=33), dict(gamma=0.95, temperature=0.1, mask=3), 1.0, [0]),
        ("doors", doors, 0, dict(budget=400, temperature=3000), dict(gamma=0.9, temperature=1.0), None, [0, 1]),
    ]
    import prior_agents
    for name, cfg, s0, agent_cfg, prior_cfg, roll_temp, seeds in cases:
        for seed in seeds:
            env = make_env(cfg, state=s0)
            agent = agent_factory(env, dict(agent_cfg, __class__=UCTP, prior_agent=dict(prior_cfg, __class__=PRIOR)))
            prior_table = np.array(agent.prior_agent.table)
            rollout_table = prior_table
            if roll_temp is not None:
                # a second policy for the rollouts, installed the way the reference installs its own
                # (mcts_with_prior.py:31-32 assigns planner.rollout_policy)
                rollout_table = prior_agents.boltzmann_table(agent.prior_agent.q, roll_temp)
                agent.planner.rollout_policy = \
                    lambda state, observation, t=rollout_table: (list(range(t.shape[1])), list(t[observation]))
            agent.seed(seed)
            st0 = rng_state(agent.planner.np_random)
            plan = agent.plan(s0)
            root = agent.planner.root
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64),
                                   ("value", lambda n: float(n.value), np.float64),
                                   ("prior", lambda n: float(n.prior), np.float64)])
            pc = agent.planner.config
            p = "uct_prior/{}_seed{}".format(name, seed)
            put_mdp(store, p + "/mdp", cfg)
            put(store, p, dict(s0=s0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], episodes=pc["episodes"],
                               horizon=pc["horizon"], temperature=pc["temperature"], prior_table=prior_table,
                               rollout_table=rollout_table, q=np.array(agent.prior_agent.q),
                               prior_gamma=prior_cfg["gamma"], prior_temperature=prior_cfg["temperature"],
                               plan=np.asarray(plan, np.int32), root_count=root.count, root_value=float(root.value),
                               env_steps=len(agent.planner.observations),
                               rng_before=st0, rng_after=rng_state(agent.planner.np_random)))
            put(store, p + "/tree", tree)
            names.append("{}_seed{}".format(name, seed))
    store["uct_prior/names"] = np.asarray(names)

    # step_strategy "subtree" with a state-dependent prior (the reference's vi_prior.json uses it)
    env = make_env(hw, state=5)
    agent = agent_factory(env, dict(__class__=UCTP, budget=300, horizon=12, episodes=25, step_strategy="subtree",
                                    prior_agent=dict(__class__=PRIOR, gamma=0.95, temperature=0.3)))
    agent.seed(11)
    p = "uct_prior/subtree_highway"
    put_mdp(store, p + "/mdp", hw)
    pc = agent.planner.config
    store[p + "/rng_before"] = rng_state(agent.planner.np_random)
    store[p + "/prior_table"] = np.array(agent.prior_agent.table)
    states = []
    for step in range(5):
        states.append(env.mdp.state)
        plan = agent.plan(env.mdp.state)
        root = agent.planner.root
        tree = bfs_tree(root, [("count", lambda n: n.count, np.int64), ("value", lambda n: float(n.value), np.float64),
                               ("prior", lambda n: float(n.prior), np.float64)])
        put(store, "{}/step{}".format(p, step), dict(plan=np.asarray(plan, np.int32), root_count=root.count,
                                                     root_value=float(root.value),
                                                     rng_after=rng_state(agent.planner.np_random)))
        put(store, "{}/step{}/tree".format(p, step), tree)
        _, _, term, trunc, _ = env.step(plan[0])
        if term or trunc:
            break
    put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), gamma=pc["gamma"],
                       episodes=pc["episodes"], horizon=pc["horizon"], temperature=pc["temperature"]))
    return store


def golden_state_aware():
    """StateAwarePlannerAgent (tree_search/state_aware.py): episodes of consecutive plan() calls on ONE agent -- the
    planner's state_nodes / state_values dictionaries (and the nodes of earlier trees in them) persist across plans."""
    store, names = {}, []
    large1 = load_env_config("large/env_1.json")
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    grid = generators.gridworld()
    loop = load_env_config("env_loop.json")
    cases = [
        # name, mdp cfg, start state, agent cfg, seed, number of plan() calls
        ("grid_b500", grid, 0, dict(budget=500, gamma=0.8), 0, 4),          # the reference's state-aware.json config
        ("grid_b100_g09", grid, 55, dict(budget=100, gamma=0.9), 3, 3),
        ("grid_accuracy", grid, 12, dict(budget=300, gamma=0.8, accuracy=0.05), 1, 3),
        ("grid_no_aggregation", grid, 0, dict(budget=300, gamma=0.8, backup_aggregated_nodes=False), 2, 3),
        ("grid_no_pruning", grid, 0, dict(budget=200, gamma=0.8, prune_suboptimal_leaves=False), 2, 2),
        ("large1_b500", large1, 0, dict(budget=500, gamma=0.8), 0, 3),
        ("large1_b37_g05", large1, 3, dict(budget=37, gamma=0.5), 2, 2),
        ("highway_small", hw, 0, dict(budget=300, gamma=0.8), 0, 4),
        ("highway_small_tr05", hw, 41, dict(budget=300, gamma=0.9, terminal_reward=0.5), 3, 3),
        ("loop_b60", loop, 0, dict(budget=60, gamma=0.7), 4, 3),
    ]
    for name, cfg, s_start, agent_cfg, seed, n_plans in cases:
        env = make_env(cfg, state=s_start)
        agent = agent_factory(env, dict(agent_cfg, __class__=SAOPD))
        agent.seed(seed)
        p = "sa/" + name
        put_mdp(store, p + "/mdp", cfg)
        pc = agent.planner.config
        n_states = np.asarray(cfg["reward"]).shape[0]
        store[p + "/rng_before"] = rng_state(agent.planner.np_random)
        states = []
        for step in range(n_plans):
            states.append(env.mdp.state)
            try:
                plan = agent.plan(env.mdp.state)
            except ValueError as e:     # every leaf pruned: max() of an empty leaves list (state_aware.py:95)
                assert "empty" in str(e)
                store[p + "/raises_at_step"] = np.asarray(step)
                break
            planner = agent.planner
            leaves = set(id(n) for n in planner.leaves)
            tree = bfs_tree(planner.root, [("count", lambda n: n.count, np.int64),
                                           ("lower", lambda n: float(n.value_lower), np.float64),
                                           ("reward", lambda n: float(n.reward), np.float64),
                                           ("done", lambda n: bool(n.done), bool),
                                           ("depth", lambda n: n.depth, np.int32),
                                           ("obs", lambda n: int(n.observation), np.int64),
                                           ("is_leaf", lambda n: id(n) in leaves, bool)])
            sv = np.array([planner.state_values[str(s)] if str(s) in planner.state_values else np.nan
                           for s in range(n_states)])
            q = "{}/step{}".format(p, step)
            put(store, q, dict(plan=np.asarray(plan, np.int32), state_values=sv, n_leaves=len(planner.leaves),
                               n_state_nodes=sum(len(v) for v in planner.state_nodes.values()),
                               env_steps=len(planner.observations), rng_after=rng_state(planner.np_random)))
            put(store, q + "/tree", tree)
            _, _, term, trunc, _ = env.step(plan[0])
            if term or trunc:
                break
        put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), seed=seed, budget=pc["budget"],
                           gamma=pc["gamma"], terminal_reward=agent.config["terminal_reward"], accuracy=pc["accuracy"],
                           backup_aggregated_nodes=pc["backup_aggregated_nodes"],
                           prune_suboptimal_leaves=pc["prune_suboptimal_leaves"]))
        names.append(name)
    store["sa/names"] = np.asarray(names)
    return store


def golden_uct_cartpole():
    """MCTS on the restated CartPole (BASELINE config C3): single-root plans with full trees, and the
    reference's own functional test (tests/agents/tree_search/test_mcts.py:5-19) as a survival count."""
    store, names = {}, []
    cases = [
        # name, reset seed, warm-up steps (random actions) before planning, agent cfg, planner seeds
        ("b400_t200", 0, 0, dict(budget=400, temperature=200), [0, 1]),
        ("b1000_h50", 1, 0, dict(budget=1000, horizon=50, episodes=20), [0, 3]),
        ("b1000_h50_late", 2, 170, dict(budget=1000, horizon=50, episodes=20), [5]),       # TimeLimit inside the horizon
        ("b300_g095", 3, 12, dict(budget=300, gamma=0.95), [2]),
        ("b200_pref", 4, 5, dict(budget=200, prior_policy={"type": "preference", "action": 0, "ratio": 2},
                                  rollout_policy={"type": "preference", "action": 1, "ratio": 4}), [7]),
    ]
    for name, reset_seed, warm, agent_cfg, seeds in cases:
        for seed in seeds:
            env = CartPoleEnv()
            env.seed(reset_seed)
            env.reset()
            warm_agent = agent_factory(env, dict(__class__=UCT, budget=400, temperature=200))
            warm_agent.seed(reset_seed + 100)
            for _ in range(warm):            # the reference planner itself drives the env to a later state
                _, _, term, trunc, _ = env.step(warm_agent.act(None))
                assert not term
            agent = agent_factory(env, dict(agent_cfg, __class__=UCT))
            agent.seed(seed)
            st0 = rng_state(agent.planner.np_random)
            state0, steps0 = np.asarray(env.state, dtype=np.float64), env.steps
            plan = agent.plan(None)
            assert np.array_equal(np.asarray(env.state), state0) and env.steps == steps0   # root env untouched
            root = agent.planner.root
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64),
                                   ("value", lambda n: float(n.value), np.float64)])
            pc = agent.planner.config
            _, prior_p = agent.planner.prior_policy(env, None)
            _, roll_p = agent.planner.rollout_policy(env, None)
            p = "cartpole/{}_seed{}".format(name, seed)
            put(store, p, dict(state0=state0, steps0=steps0, seed=seed, budget=pc["budget"], gamma=pc["gamma"],
                               episodes=pc["episodes"], horizon=pc["horizon"], temperature=pc["temperature"],
                               prior_p=np.asarray(prior_p, np.float64), rollout_p=np.asarray(roll_p, np.float64),
                               plan=np.asarray(plan, np.int32), root_count=root.count, root_value=float(root.value),
                               env_steps=len(agent.planner.observations),
                               rng_before=st0, rng_after=rng_state(agent.planner.np_random)))
            put(store, p + "/tree", tree)
            names.append("{}_seed{}".format(name, seed))
    store["cartpole/names"] = np.asarray(names)
    store["cartpole/params"] = np.asarray([CartPoleEnv().cartpole_params()[k] for k in
                                           ("gravity", "masscart", "masspole", "length", "force_mag", "tau",
                                            "theta_threshold", "x_threshold", "max_steps", "euler")], dtype=np.float64)
    # the reference's functional test on this env: budget 400, temperature 200, one full episode
    env = CartPoleEnv()
    env.seed(0)
    obs, _ = env.reset()
    agent = agent_factory(env, dict(__class__=UCT, budget=400, temperature=200))
    agent.seed(0)
    steps, done, actions = 0, False, []
    while not done:
        a = agent.act(obs)
        actions.append(a)
        obs, _, term, trunc, _ = env.step(a)
        steps += 1
        done = term or trunc
    store["cartpole/episode_steps"] = np.asarray(steps)
    store["cartpole/episode_actions"] = np.asarray(actions, np.int32)
    return store


# ----------------------------------------------------------------------------- misc pins
def golden_misc():
    store = {}
    alloc_in = [(100, 0.8), (400, 0.8), (1000, 0.8), (1000, 0.95), (1000, 0.99), (5000, 0.8),
                (37, 0.5), (500, 0.9), (6, 0.8), (1000, 0.5)]
    store["alloc/in"] = np.asarray(alloc_in, dtype=np.float64)
    store["alloc/out"] = np.asarray([OLOP.allocation(int(b), g) for b, g in alloc_in], dtype=np.int64)

    # numpy Generator(PCG64) draw sequences that pin the PCG64 / bounded-integer port:
    # the planners consume the stream only through choice(indices) (Node.random_argmax,
    # tree_search/abstract.py:304-311) and choice(actions, 1, p=p) (mcts.py:172).
    seeds = [0, 1, 12345, 2 ** 40 + 17]
    for seed in seeds:
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        st0 = rng_state(g)
        # op codes: k > 0 -> choice(arange(k)) ; k == 0 -> random()
        ops = np.random.Generator(np.random.PCG64(99 + seed % 1000)).integers(0, 7, size=400)
        outs = np.zeros(400, dtype=np.float64)
        for i, k in enumerate(ops):
            if k == 0:
                outs[i] = g.random()
            else:
                outs[i] = g.choice(np.arange(k))
        p = "pcg/seed{}".format(seed)
        put(store, p, dict(state0=st0, ops=ops.astype(np.int32), outs=outs, state1=rng_state(g)))
    # choice with probabilities: index sequence for a few distributions
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    store["pchoice/state0"] = rng_state(g)
    ps = [np.ones(5) / 5, np.ones(2) / 2, np.ones(4) / (4 - 1 + 3.0) * np.array([1, 3, 1, 1]), np.ones(3) / 3]
    for j, p in enumerate(ps):
        store["pchoice/p{}".format(j)] = p
        store["pchoice/out{}".format(j)] = np.asarray(
            [g.choice(np.arange(len(p)), 1, p=np.array(p))[0] for _ in range(200)], dtype=np.int32)
    store["pchoice/state1"] = rng_state(g)
    return store


def main():
    out = os.path.join(REPO, "tests", "golden")
    only = sys.argv[1:]
    for name, fn in (("vi", golden_vi), ("opd", golden_opd), ("uct", golden_uct), ("uct_cartpole", golden_uct_cartpole),
                     ("uct_prior", golden_uct_prior), ("state_aware", golden_state_aware), ("misc", golden_misc)):
        if only and name not in only:
            continue
        store = fn()
        path = os.path.join(out, name + ".npz")
        np.savez_compressed(path, **store)
        print("{}: {} arrays, {:.1f} KB".format(path, len(store), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
