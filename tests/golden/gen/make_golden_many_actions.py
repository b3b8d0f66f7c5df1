#!/usr/bin/env python3
"""Golden vectors: the UNMODIFIED reference MCTSAgent / MCTSWithPriorPolicyAgent with per-state policies over MORE THAN 8
ACTIONS -- environments that restrict their available actions (mcts.py:59-97) and prior agents (mcts_with_prior.py:47-62)
on deterministic, sparse and dense finite MDPs with 9 .. 40 actions, open and closed loop.  The device kernels keep a
node's policy rows in registers for 2..8 actions; beyond that the loop forms of uct_stoch.hip plan (round 4), on
deterministic tables too.  Same schema as stoch_policies.npz (make_golden_stoch_policies.py builds both).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_many_actions.py      (build container only)
-> tests/golden/many_actions.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import generators, np  # noqa: E402
from make_golden_stoch_policies import build  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "many_actions.npz"))


def main():
    store, names = {}, []
    det12 = generators.random_deterministic(80, 12, seed=11, terminal_rate=0.05)
    det40 = generators.random_deterministic(60, 40, seed=12, terminal_rate=0.05)       # more than 32: a byte per listed action
    sparse10 = generators.random_sparse(70, 10, 2, seed=13, terminal_rate=0.1)
    dense9 = generators.random_stochastic(40, 9, seed=14, concentration=0.05)
    pref = {"type": "preference", "action": 1, "ratio": 3}
    rnd = {"type": "random"}
    prior_cfg = dict(__class__=mg.PRIOR, gamma=0.9, temperature=0.5)
    cases = [
        ("masked_det12_open", det12, (5, 0.4), 3, mg.UCT, dict(budget=600, gamma=0.9), [0]),
        ("masked_det12_closed", det12, (5, 0.4), 3, mg.UCT, dict(budget=600, gamma=0.9, closed_loop=True), [1]),
        ("masked_det40_open_pref", det40, (6, 0.5), 7, mg.UCT, dict(budget=800, prior_policy=pref, rollout_policy=pref), [2]),
        ("masked_det40_closed", det40, (6, 0.5), 7, mg.UCT, dict(budget=800, closed_loop=True), [9]),
        ("prior_det12_open", det12, None, 9, mg.UCTP, dict(budget=600, gamma=0.9, prior_agent=prior_cfg), [0]),
        ("prior_det12_closed_masked_env", det12, (5, 0.4), 9, mg.UCTP,
         dict(budget=600, gamma=0.9, closed_loop=True, prior_agent=prior_cfg), [3]),
        ("masked_sparse10_closed", sparse10, (7, 0.4), 5, mg.UCT, dict(budget=500, gamma=0.9, closed_loop=True), [4]),
        ("masked_sparse10_rollout_random", sparse10, (7, 0.4), 5, mg.UCT, dict(budget=500, rollout_policy=rnd), [6]),
        ("prior_dense9_open", dense9, None, 2, mg.UCTP, dict(budget=1000, horizon=30, episodes=33, prior_agent=prior_cfg), [5]),
    ]
    build(cases, store, names)
    store["stoch_policies/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays,", len(names), "cases")


if __name__ == "__main__":
    main()
