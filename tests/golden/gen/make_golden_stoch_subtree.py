#!/usr/bin/env python3
"""Golden vectors: the UNMODIFIED reference MCTSAgent with ``step_strategy="subtree"`` on STOCHASTIC finite MDPs, open loop
(AbstractPlanner.step_by_subtree, tree_search/abstract.py:195-206, keeps the subtree of the executed action whatever the
environment; closed-loop trees cannot be re-used in the reference either).  Multi-step episodes: the live env is stepped
with the first planned action (its own generator advances), the planner re-roots and plans again.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_stoch_subtree.py      (build container only)
-> tests/golden/stoch_subtree.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import agent_factory, generators, np  # noqa: E402
from make_golden_variants import UCT_FIELDS, keyed_tree  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "stoch_subtree.npz"))


def main():
    store, names = {}, []
    dense = generators.random_stochastic(30, 3, seed=5, terminal_rate=0.05)
    sparse = generators.random_sparse(60, 3, 2, seed=7, terminal_rate=0.05)
    sparse_b = generators.random_sparse(200, 5, 4, seed=8)
    pref = {"type": "preference", "action": 1, "ratio": 3}
    cases = [
        ("dense", dense, 0, dict(budget=300, horizon=10, episodes=30), 0, 6),
        ("sparse", sparse, 5, dict(budget=400, gamma=0.9, horizon=12, episodes=33), 1, 6),
        ("sparse_b_pref_rh2", sparse_b, 17, dict(budget=300, horizon=10, episodes=30, prior_policy=pref, rollout_policy=pref,
                                                receding_horizon=2), 2, 7),
    ]
    for name, cfg, s0, acfg, seed, n_steps in cases:
        env = mg.make_env(cfg, state=s0)
        env.seed(1000 + seed)
        agent = agent_factory(env, dict(acfg, __class__=mg.UCT, step_strategy="subtree"))
        agent.seed(seed)
        p = "stoch_subtree/" + name
        mg.put_mdp(store, p + "/mdp", cfg)
        pc = agent.planner.config
        mg.put(store, p, dict(s0=s0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], episodes=pc["episodes"],
                              horizon=pc["horizon"], temperature=pc["temperature"], rng_before=mg.rng_state(agent.planner.np_random),
                              receding_horizon=agent.config["receding_horizon"]))
        states, done = [], False
        for step in range(n_steps):
            states.append(env.mdp.state)
            q = "{}/step{}".format(p, step)
            store[q + "/env_rng"] = mg.rng_state(env.np_random)
            plan = agent.plan(env.mdp.state)
            root = agent.planner.root
            mg.put(store, q, dict(plan=np.asarray(plan, np.int32), root_count=root.count, root_value=float(root.value),
                                  rng_after=mg.rng_state(agent.planner.np_random), env_steps=len(agent.planner.observations)))
            mg.put(store, q + "/tree", keyed_tree(root, UCT_FIELDS))
            _, _, term, trunc, _ = env.step(plan[0])
            if term or trunc:
                done = True
                break
        mg.put(store, p, dict(states=np.asarray(states, np.int32), n_steps=len(states), ended=done))
        names.append(name)
    store["stoch_subtree/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays;", {n: int(store["stoch_subtree/%s/n_steps" % n]) for n in names})


if __name__ == "__main__":
    main()
