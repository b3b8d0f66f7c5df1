#!/usr/bin/env python3
"""Golden vectors: the UNMODIFIED reference MCTSAgent / MCTSWithPriorPolicyAgent on environments that list their available
actions in a NON-ASCENDING order (OrderedMaskedFiniteMDPEnv: the restriction and the order live on the env object, as in
highway-env, which lists IDLE first) -- on STOCHASTIC / SPARSE finite MDPs and, with more than 8 actions, on a deterministic
table.  The reference creates a node's children in listing order and its tie-breaks index that order (mcts.py:237-246,
abstract.py:296-311); the device plans in the permuted action space and maps labels back.  Same schema as stoch_policies.npz.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_listing_order.py      (build container only)
-> tests/golden/listing_order.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from make_golden import generators, np  # noqa: E402
from make_golden_stoch_policies import build  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "listing_order.npz"))


def main():
    store, names = {}, []
    sparse = generators.random_sparse(60, 5, 2, seed=31, terminal_rate=0.1)
    dense = generators.random_stochastic(30, 4, seed=32, terminal_rate=0.1)
    det12 = generators.random_deterministic(80, 12, seed=33, terminal_rate=0.05)
    idle_first = [1, 0, 2, 3, 4]
    rev4 = [3, 2, 1, 0]
    mix12 = [5, 0, 11, 3, 2, 7, 1, 9, 4, 10, 6, 8]
    rnd = {"type": "random"}
    pref = {"type": "preference", "action": 2, "ratio": 3}
    prior_cfg = dict(__class__=mg.PRIOR, gamma=0.9, temperature=0.5)
    cases = [
        ("ordered_sparse_open", sparse, (1, 0.4, idle_first), 5, mg.UCT, dict(budget=400, gamma=0.9), [0, 1]),
        ("ordered_sparse_closed", sparse, (1, 0.4, idle_first), 5, mg.UCT, dict(budget=400, gamma=0.9, closed_loop=True), [2]),
        ("ordered_sparse_random_policies", sparse, (1, 0.4, idle_first), 5, mg.UCT,
         dict(budget=400, prior_policy=rnd, rollout_policy=rnd), [3]),           # type random lists np.arange(n) whatever the env lists
        ("ordered_sparse_rollout_random", sparse, (1, 0.4, idle_first), 7, mg.UCT, dict(budget=400, rollout_policy=rnd), [4]),
        ("ordered_dense_closed_pref", dense, (2, 0.3, rev4), 0, mg.UCT,
         dict(budget=300, closed_loop=True, prior_policy=pref, rollout_policy=pref), [5]),
        ("ordered_sparse_prior_agent", sparse, (1, 0.4, idle_first), 9, mg.UCTP, dict(budget=400, gamma=0.9, prior_agent=prior_cfg), [6]),
        ("ordered_det12_open", det12, (5, 0.4, mix12), 3, mg.UCT, dict(budget=600, gamma=0.9), [7]),
        ("ordered_det12_closed", det12, (5, 0.4, mix12), 3, mg.UCT, dict(budget=600, gamma=0.9, closed_loop=True), [8]),
    ]
    build(cases, store, names)
    store["stoch_policies/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays,", len(names), "cases")


if __name__ == "__main__":
    main()
