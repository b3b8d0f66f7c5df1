#!/usr/bin/env python3
"""Golden vectors: the UNMODIFIED reference MCTSAgent / MCTSWithPriorPolicyAgent on STOCHASTIC finite MDPs with per-state
policies -- environments that restrict their available actions (policies over ``state.get_available_actions()``,
mcts.py:59-97) and prior agents (``agent_policy_available``, mcts_with_prior.py:47-62) -- open and closed loop.  On a
stochastic env a node is expanded with the actions / priors of the state the env happens to be in AT THAT MOMENT
(mcts.py:151-154,237-246) and keeps them, whatever state later episodes reach it in (open loop).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_stoch_policies.py      (build container only)
-> tests/golden/stoch_policies.npz
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import agent_factory, generators, np  # noqa: E402
from make_golden_variants import UCT_FIELDS, keyed_tree, make_masked_env  # noqa: E402
import prior_agents  # noqa: E402,F401

OUT = os.path.abspath(os.path.join(HERE, "..", "stoch_policies.npz"))


def build(cases, store, names):
    """cases: (name, mdp config, (availability seed, rate) or None, root state, agent class, agent config, seeds)."""
    for name, cfg, av, s0, klass, acfg, seeds in cases:
        r = np.asarray(cfg["reward"])
        avail = None if av is None else generators.random_available(r.shape[0], r.shape[1], seed=av[0], rate=av[1])
        order = av[2] if av is not None and len(av) > 2 else None    # the env LISTS its actions in this (non-ascending) order
        for seed in seeds:
            if order is not None:
                from rl_agents_amd.envs import OrderedMaskedFiniteMDPEnv
                c = {k: v for k, v in cfg.items() if k != "original_shape"}
                c.update(state=int(s0), available=np.asarray(avail), listing_order=list(order))
                env = OrderedMaskedFiniteMDPEnv(c)
                env.reset()
            else:
                env = mg.make_env(cfg, state=s0) if avail is None else make_masked_env(cfg, avail, state=s0)
            env.seed(1000 + seed)
            env_rng = mg.rng_state(env.np_random)
            agent = agent_factory(env, dict(acfg, __class__=klass))
            agent.seed(seed)
            st0 = mg.rng_state(agent.planner.np_random)
            plan = agent.plan(s0)
            root = agent.planner.root
            pc = agent.planner.config
            p = "stoch_policies/{}_seed{}".format(name, seed)
            mg.put_mdp(store, p + "/mdp", cfg)
            extra = {}
            if avail is not None:
                extra["available"] = np.asarray(avail, bool)
            if order is not None:
                extra["listing_order"] = np.asarray(order, np.int32)
            if klass == mg.UCTP:
                extra.update(prior_table=np.array(agent.prior_agent.table), prior_gamma=acfg["prior_agent"]["gamma"],
                             prior_temperature=acfg["prior_agent"]["temperature"], prior_mask=acfg["prior_agent"].get("mask", 0))
            mg.put(store, p, dict(s0=s0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], episodes=pc["episodes"],
                                  horizon=pc["horizon"], temperature=pc["temperature"], closed_loop=bool(pc["closed_loop"]),
                                  with_prior_agent=klass == mg.UCTP,
                                  plan=np.asarray([int(x) for x in plan], np.int32),
                                  plan_is_obs=np.asarray([isinstance(x, str) for x in plan], bool),
                                  root_count=root.count, root_value=float(root.value), env_steps=len(agent.planner.observations),
                                  rng_before=st0, rng_after=mg.rng_state(agent.planner.np_random), env_rng=env_rng, **extra))
            store[p + "/prior_policy_json"] = np.asarray(json.dumps(agent.config["prior_policy"]))
            store[p + "/rollout_policy_json"] = np.asarray(json.dumps(agent.config["rollout_policy"]))
            mg.put(store, p + "/tree", keyed_tree(root, UCT_FIELDS))
            assert env.mdp.state == s0 and np.array_equal(mg.rng_state(env.np_random), env_rng)
            names.append("{}_seed{}".format(name, seed))


def main():
    store, names = {}, []
    dense = generators.random_stochastic(30, 3, seed=5, terminal_rate=0.1)
    dense_b = generators.random_stochastic(50, 5, seed=6, concentration=0.05)
    sparse = generators.random_sparse(60, 3, 2, seed=7, terminal_rate=0.1)
    sparse_b = generators.random_sparse(200, 5, 4, seed=8)
    pref = {"type": "preference", "action": 1, "ratio": 3}
    rnd = {"type": "random"}
    prior_cfg = dict(__class__=mg.PRIOR, gamma=0.9, temperature=0.5)
    prior_masked = dict(__class__=mg.PRIOR, gamma=0.9, temperature=0.5, mask=3)
    cases = [
        # name, cfg, available (seed, rate) or None, s0, agent class, agent cfg, seeds
        ("masked_sparse_open", sparse, (1, 0.4), 5, mg.UCT, dict(budget=400, gamma=0.9), [0, 1]),
        ("masked_sparse_closed", sparse, (1, 0.4), 5, mg.UCT, dict(budget=400, gamma=0.9, closed_loop=True), [0, 4]),
        ("masked_dense_open_pref", dense, (2, 0.3), 0, mg.UCT, dict(budget=300, prior_policy=pref, rollout_policy=pref), [2]),
        ("masked_dense_closed", dense, (2, 0.3), 0, mg.UCT, dict(budget=300, closed_loop=True), [1]),
        ("masked_sparse_b_h30_closed", sparse_b, (3, 0.35), 17, mg.UCT,
         dict(budget=1000, horizon=30, episodes=33, closed_loop=True), [1]),
        ("masked_sparse_b_rollout_random", sparse_b, (3, 0.35), 17, mg.UCT, dict(budget=400, rollout_policy=rnd), [5]),
        ("prior_sparse_open", sparse, None, 5, mg.UCTP, dict(budget=400, gamma=0.9, prior_agent=prior_cfg), [0]),
        ("prior_sparse_closed", sparse, None, 5, mg.UCTP, dict(budget=400, gamma=0.9, closed_loop=True, prior_agent=prior_cfg), [3]),
        ("prior_dense_b_open", dense_b, None, 7, mg.UCTP, dict(budget=1000, horizon=30, episodes=33, prior_agent=prior_cfg), [2]),
        ("prior_dense_b_closed_masked_table", dense_b, None, 11, mg.UCTP,
         dict(budget=400, closed_loop=True, prior_agent=prior_masked), [6]),
        ("prior_masked_env_sparse_b_open", sparse_b, (4, 0.3), 3, mg.UCTP, dict(budget=400, prior_agent=prior_cfg), [7]),
        ("prior_masked_env_sparse_closed", sparse, (1, 0.4), 9, mg.UCTP, dict(budget=300, closed_loop=True, prior_agent=prior_cfg), [8]),
    ]
    build(cases, store, names)
    store["stoch_policies/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays,", len(names), "cases")


if __name__ == "__main__":
    main()
