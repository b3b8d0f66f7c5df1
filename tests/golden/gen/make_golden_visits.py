#!/usr/bin/env python3
"""Golden vectors: AbstractPlanner.get_visits (abstract.py:163-167) of the UNMODIFIED reference MCTSAgent -- how often the
env steps of its plans (descents AND rollouts, every plan since the planner was made: planner.observations is never
cleared) observed each state -- after one plan and after several act() calls (receding horizon, step_strategy reset and
subtree), on deterministic, sparse and restricted environments, open and closed loop.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_visits.py      (build container only)
-> tests/golden/visits.npz
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import agent_factory, generators, np  # noqa: E402
from make_golden_variants import make_masked_env  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "visits.npz"))


def main():
    store, names = {}, []
    det = generators.random_deterministic(60, 4, seed=51, terminal_rate=0.05)
    det12 = generators.random_deterministic(80, 12, seed=52, terminal_rate=0.05)
    sparse = generators.random_sparse(50, 3, 2, seed=53, terminal_rate=0.1)
    cases = [
        # name, cfg, availability (seed, rate) or None, root state, agent config, seed, act() calls
        ("det_one_plan", det, None, 3, dict(budget=300, gamma=0.9), 0, 1),
        ("det_three_acts_reset", det, None, 3, dict(budget=200, gamma=0.9), 1, 3),
        ("det_three_acts_subtree", det, None, 3, dict(budget=200, gamma=0.9, step_strategy="subtree"), 2, 3),
        ("det_closed_loop", det, None, 7, dict(budget=200, gamma=0.9, closed_loop=True), 3, 2),
        ("masked_det_two_acts", det, (4, 0.4), 5, dict(budget=200, gamma=0.9), 4, 2),
        ("masked_det12_subtree", det12, (5, 0.4), 3, dict(budget=300, gamma=0.9, step_strategy="subtree"), 5, 2),
        ("sparse_closed_two_acts", sparse, None, 5, dict(budget=200, gamma=0.9, closed_loop=True), 6, 2),
        ("sparse_subtree_three_acts", sparse, None, 5, dict(budget=200, gamma=0.9, step_strategy="subtree"), 7, 3),
    ]
    for name, cfg, av, s0, acfg, seed, n_acts in cases:
        r = np.asarray(cfg["reward"])
        avail = None if av is None else generators.random_available(r.shape[0], r.shape[1], seed=av[0], rate=av[1])
        env = mg.make_env(cfg, state=s0) if avail is None else make_masked_env(cfg, avail, state=s0)
        env.seed(1000 + seed)
        agent = agent_factory(env, dict(acfg, __class__=mg.UCT))
        agent.seed(seed)
        obs, actions = s0, []
        for _ in range(n_acts):
            a = agent.act(obs)
            actions.append(int(a))
            out = env.step(a)
            obs = out[0]
            if out[2] or (len(out) > 4 and out[3]):
                break
        visits = agent.planner.get_visits()
        states = sorted(int(k) for k in visits)
        p = "visits/" + name
        mg.put_mdp(store, p + "/mdp", cfg)
        extra = {} if avail is None else {"available": np.asarray(avail, bool)}
        mg.put(store, p, dict(s0=s0, seed=seed, n_acts=n_acts, actions=np.asarray(actions, np.int32),
                              visit_states=np.asarray(states, np.int64),
                              visit_counts=np.asarray([visits[str(k)] for k in states], np.int64),
                              total=len(agent.planner.observations), **extra))
        store[p + "/agent_json"] = np.asarray(json.dumps(acfg))
        names.append(name)
    store["visits/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays,", len(names), "cases")


if __name__ == "__main__":
    main()
