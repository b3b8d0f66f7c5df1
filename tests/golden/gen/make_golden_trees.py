#!/usr/bin/env python3
"""Golden vectors for the consumers of an exported tree: what the UNMODIFIED reference returns from
``Node.get_obs_visits`` / ``get_trajectories`` / ``breadth_first_search`` (tree_search/abstract.py:246-265,319-358) and
``AbstractPlanner.get_visits`` (:163-167) on the trees of its own plans.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/make_golden_trees.py      (build container only)

-> tests/golden/tree_tools.npz: per case the MDP, the plan inputs, the tree (BFS listing as in make_golden.py) and the
outputs of those functions.  Nothing of the reference is copied: inputs and its outputs only.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import OPD, UCT, agent_factory, bfs_tree, generators, load_env_config, make_env, np, put, put_mdp, rng_state  # noqa: E402,E501

from rl_agents.agents.tree_search.abstract import Node  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, "..", "tree_tools.npz"))


class OldGymEnv(object):
    """4-tuple ``step`` around a 5-tuple env: ``Node.get_obs_visits``'s replay branch still unpacks the old gym API
    (tree_search/abstract.py:356), so it only runs on such an env; deep-copyable like any env."""

    def __init__(self, env):
        self.env = env

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return obs, reward, terminated or truncated, info


def ragged(lists):
    flat = [int(x) for lst in lists for x in lst]
    offs = np.cumsum([0] + [len(lst) for lst in lists])
    return np.asarray(flat, np.int64), np.asarray(offs, np.int64)


def main():
    store, names = {}, []
    large1 = {k: v for k, v in load_env_config("large/env_1.json").items() if k != "max_steps"}
    hw = generators.highway_shaped(3, 4, 10, seed=3)
    grid = generators.gridworld()
    cases = [
        ("uct_highway_small", UCT, hw, 0, dict(budget=1000, horizon=30, episodes=33), 0),
        ("uct_large1_b100", UCT, large1, 0, dict(budget=100), 1),
        ("opd_grid_c1", OPD, grid, 0, dict(budget=100, gamma=0.8), 0),
        ("opd_highway_small", OPD, hw, 0, dict(budget=300, gamma=0.8), 0),
    ]
    for name, klass, cfg, s0, agent_cfg, seed in cases:
        env = make_env(cfg, state=s0)
        agent = agent_factory(env, dict(agent_cfg, __class__=klass))
        agent.seed(seed)
        st0 = rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        root = agent.planner.root
        if klass == UCT:
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64), ("value", lambda n: float(n.value), np.float64)])
        else:
            tree = bfs_tree(root, [("count", lambda n: n.count, np.int64), ("lower", lambda n: float(n.value_lower), np.float64),
                                   ("upper", lambda n: float(n.value_upper), np.float64)])
        p = "trees/" + name
        put_mdp(store, p + "/mdp", cfg)
        pc = agent.planner.config
        put(store, p, dict(s0=s0, seed=seed, budget=pc["budget"], gamma=pc["gamma"], plan=np.asarray(plan, np.int32),
                           is_uct=klass == UCT, episodes=pc.get("episodes", 0), horizon=pc.get("horizon", 0) or 0,
                           temperature=pc.get("temperature", 0.0), rng_before=st0))
        put(store, p + "/tree", tree)
        # Node.get_obs_visits (replay branch: these nodes hold no `observation` ... the OPD ones do)
        visits, updates = root.get_obs_visits(state=OldGymEnv(env))
        keys = sorted(visits)
        put(store, p, dict(visit_keys=np.asarray(keys),
                           visit_counts=np.asarray([visits[k] for k in keys], np.int64),
                           n_updates=len(updates)))
        # Node.get_trajectories
        full = root.get_trajectories(full_trajectories=True, include_leaves=True)
        flat = root.get_trajectories(full_trajectories=False, include_leaves=False)
        flat_leaves = root.get_trajectories(full_trajectories=False, include_leaves=True)
        put(store, p, dict(n_full=len(full), full_lengths=np.asarray([len(t) for t in full], np.int64),
                           n_flat=len(flat), n_flat_with_leaves=len(flat_leaves),
                           flat_counts=np.asarray([n.count for n in flat], np.int64)))
        # Node.breadth_first_search: every node, and the leaves only (blocking condition)
        paths = [list(path) for _, path in Node.breadth_first_search(root)]
        flat_p, offs = ragged(paths)
        put(store, p, dict(bfs_paths=flat_p, bfs_offsets=offs))
        leaf_paths = [list(path) for _, path in Node.breadth_first_search(root, condition=lambda n: n.is_leaf())]
        flat_p, offs = ragged(leaf_paths)
        put(store, p, dict(bfs_leaf_paths=flat_p, bfs_leaf_offsets=offs))
        counts = list(Node.breadth_first_search(root, operator=lambda n, path: n.count))
        put(store, p, dict(bfs_counts=np.asarray(counts, np.int64)))
        # AbstractPlanner.get_visits: every observation the planner stepped through
        pv = agent.planner.get_visits()
        keys = sorted(pv)
        put(store, p, dict(planner_visit_keys=np.asarray(keys),
                           planner_visit_counts=np.asarray([pv[k] for k in keys], np.int64)))
        # str(node) of the root's first child and of the node the plan ends in
        node = root
        for a in plan:
            node = node.children[a]
        put(store, p, dict(str_plan_end=np.asarray(str(node)), str_root=np.asarray(str(root))))
        names.append(name)
    store["trees/names"] = np.asarray(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, len(store), "arrays")


if __name__ == "__main__":
    main()
