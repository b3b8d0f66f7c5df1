"""Shared helpers for the parity tests."""
import numpy as np


def mdp_from_golden(z, prefix):
    """Rebuild the finite-MDP config stored by make_golden.put_mdp."""
    cfg = dict(mode=str(z[prefix + "/mode"]), transition=z[prefix + "/transition"], reward=z[prefix + "/reward"],
               terminal=z[prefix + "/terminal"], max_steps=int(z[prefix + "/max_steps"]))
    if cfg["mode"] == "sparse":
        cfg["next"] = z[prefix + "/next"]
    return cfg


def bfs_order(parent, first_child, n_actions):
    """Creation-order tree arrays -> canonical BFS permutation (same order make_golden.bfs_tree uses).

    Returns (order, bfs_parent, bfs_action): order[i] = creation index of the i-th BFS node.
    """
    order, bpar, bact = [0], [-1], [-1]
    i = 0
    while i < len(order):
        fc = int(first_child[order[i]])
        if fc >= 0:
            for a in range(n_actions):
                order.append(fc + a)
                bpar.append(i)
                bact.append(a)
        i += 1
    return np.asarray(order), np.asarray(bpar, np.int32), np.asarray(bact, np.int32)


def assert_tree_equal(z, prefix, tree, n_actions, fields):
    """Compare a creation-order tree (dict of arrays) with a golden BFS tree, bit for bit."""
    order, bpar, bact = bfs_order(tree["parent"], tree["first_child"], n_actions)
    assert len(order) == len(z[prefix + "/parent"]), (len(order), len(z[prefix + "/parent"]))
    np.testing.assert_array_equal(bpar, z[prefix + "/parent"])
    np.testing.assert_array_equal(bact, z[prefix + "/action"])
    for gold_name, mine in fields.items():
        got = np.asarray(tree[mine])[order]
        want = z[prefix + "/" + gold_name]
        assert np.array_equal(got.astype(want.dtype), want), "tree field {} differs".format(gold_name)


def replay_state_aware_episode(z, name, plan_fn):
    """Replay one golden StateAwarePlannerAgent episode (consecutive plan() calls on one planner) and compare every
    plan, tree, leaves set, state-value table, env-step count and generator state with the reference's.

    plan_fn(cfg, s0, params, rng, planner_state) -> dict(plan, env_steps, rng_after, tree, state_values, planner);
    raises ValueError where the reference does.  tree: creation-order arrays re-based at the root."""
    import pytest
    p = "sa/" + name
    cfg = mdp_from_golden(z, p + "/mdp")
    a = cfg["reward"].shape[1]
    params = dict(budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                  terminal_reward=float(z[p + "/terminal_reward"]), accuracy=float(z[p + "/accuracy"]),
                  backup_aggregated_nodes=bool(z[p + "/backup_aggregated_nodes"]),
                  prune_suboptimal_leaves=bool(z[p + "/prune_suboptimal_leaves"]))
    rng = np.array(z[p + "/rng_before"], dtype=np.uint64)
    raises_at = int(z[p + "/raises_at_step"]) if p + "/raises_at_step" in z.files else -1
    planner, env_steps = None, 0
    for step in range(int(z[p + "/n_steps"])):
        s0 = int(z[p + "/states"][step])
        if step == raises_at:
            with pytest.raises(ValueError):
                plan_fn(cfg, s0, params, rng, planner)
            break
        out = plan_fn(cfg, s0, params, rng, planner)
        q = "{}/step{}".format(p, step)
        np.testing.assert_array_equal(out["plan"], z[q + "/plan"], err_msg=q)
        np.testing.assert_array_equal(out["rng_after"], z[q + "/rng_after"], err_msg=q)
        env_steps += int(out["env_steps"])
        assert env_steps == int(z[q + "/env_steps"]), q
        tree = out["tree"]
        assert int(tree["alive"].sum()) == int(z[q + "/n_leaves"]), q
        assert_tree_equal(z, q + "/tree", tree, a, dict(count="count", lower="lower", reward="reward", done="done",
                                                        depth="depth", obs="state", is_leaf="alive"))
        want = z[q + "/state_values"]
        seen = ~np.isnan(want)              # states the reference's defaultdict holds; the others are at the default
        assert np.array_equal(out["state_values"][seen], want[seen]), q
        assert np.all(out["state_values"][~seen] == 1 / (1 - params["gamma"])), q
        planner, rng = out["planner"], out["rng_after"]


def bfs_children(first_child, n_children):
    """Creation-order trees whose nodes have a variable number of (contiguous) children -> BFS permutation and the BFS
    parent index of every node (the order make_golden_variants.keyed_tree lists a reference tree in)."""
    order, bpar = [0], [-1]
    i = 0
    while i < len(order):
        n = order[i]
        for j in range(int(n_children[n])):
            order.append(int(first_child[n]) + j)
            bpar.append(i)
        i += 1
    return np.asarray(order), np.asarray(bpar, np.int32)


def assert_keyed_tree_equal(z, prefix, tree, fields):
    """Compare a creation-order tree with per-node `action` keys and `n_children` with a golden keyed BFS tree."""
    order, bpar = bfs_children(tree["first_child"], tree["n_children"])
    assert len(order) == len(z[prefix + "/parent"]), (len(order), len(z[prefix + "/parent"]))
    np.testing.assert_array_equal(bpar, z[prefix + "/parent"])
    np.testing.assert_array_equal(np.asarray(tree["action"])[order], z[prefix + "/action"])
    for gold_name, mine in fields.items():
        got = np.asarray(tree[mine])[order]
        want = z[prefix + "/" + gold_name]
        assert np.array_equal(got.astype(want.dtype), want), "tree field {} differs".format(gold_name)


def reference_policy_lists(policy_config, available, order=None):
    """What the reference's policy functions return, state by state, on an environment whose
    get_available_actions() lists flatnonzero(available[s]) (mcts.py:46-97) -- or, with ``order`` (a permutation of the
    action ids), lists the available actions in that order: dict(actions=[...], p=[...]).
    Restated here for the tests only (the oracle consumes the lists; the product builds [S, A] tables of its own)."""
    available = np.asarray(available).astype(bool)
    n_states, n_actions = available.shape
    actions, probs = [], []
    for s in range(n_states):
        av = np.flatnonzero(available[s]) if order is None else np.asarray([a for a in order if available[s, a]])
        kind = policy_config["type"]
        if kind == "random":                                  # mcts.py:46-57: ignores availability
            a, p = np.arange(n_actions), np.ones(n_actions) / n_actions
        elif kind == "random_available":                      # mcts.py:59-73
            a, p = av, np.ones(len(av)) / len(av)
        elif kind == "preference":                            # mcts.py:75-97
            a, p = av, np.ones(len(av)) / len(av)
            for i in range(len(av)):
                if av[i] == policy_config["action"]:
                    p = np.ones(len(av)) / (len(av) - 1 + policy_config["ratio"])
                    p[i] *= policy_config["ratio"]
                    break
        else:
            raise ValueError("Unknown policy type")
        actions.append([int(x) for x in a])
        probs.append(np.asarray(p, dtype=np.float64))
    return dict(actions=actions, p=probs)


def restricted_agent_policy_lists(table, available, order=None):
    """MCTSWithPriorPolicyAgent.agent_policy_available (mcts_with_prior.py:56-62): the prior agent's distribution
    restricted to the available actions (in the env's listing order) and renormalised with numpy's sum."""
    available = np.asarray(available).astype(bool)
    actions, probs = [], []
    for s in range(available.shape[0]):
        av = np.flatnonzero(available[s]) if order is None else np.asarray([a for a in order if available[s, a]])
        p = np.array([table[s, a] for a in av])
        p /= np.sum(p)
        actions.append([int(x) for x in av])
        probs.append(p)
    return dict(actions=actions, p=probs)


def bfs_by_parent(parent):
    """Creation-order parent array -> BFS permutation in which a node's children come in ascending id = the order in
    which the reference inserted them into its `children` dict (expansion order for action nodes, first-visit order for
    observation nodes).  -> (order, bfs_parent)."""
    kids = [[] for _ in parent]
    for i, p in enumerate(parent):
        if p >= 0:
            kids[int(p)].append(i)
    order, bpar = [0], [-1]
    i = 0
    while i < len(order):
        for c in kids[order[i]]:
            order.append(c)
            bpar.append(i)
        i += 1
    return np.asarray(order), np.asarray(bpar, np.int32)


def assert_parent_tree_equal(z, prefix, tree, fields):
    """Compare a creation-order tree given by `parent` / `action` (= key) arrays with a golden keyed BFS tree."""
    order, bpar = bfs_by_parent(tree["parent"])
    assert len(order) == len(z[prefix + "/parent"]), (len(order), len(z[prefix + "/parent"]))
    np.testing.assert_array_equal(bpar, z[prefix + "/parent"])
    np.testing.assert_array_equal(np.asarray(tree["action"])[order], z[prefix + "/action"])
    for gold_name, mine in fields.items():
        got = np.asarray(tree[mine])[order]
        want = z[prefix + "/" + gold_name]
        assert np.array_equal(got.astype(want.dtype), want), "tree field {} differs".format(gold_name)


def replay_state_aware_masked_episode(z, name, plan_fn):
    """Replay one golden StateAwarePlannerAgent episode on an environment that restricts its actions
    (tests/golden/round3.npz, sa_masked/*): every plan, keyed tree, leaves set, state-value table, env-step count and
    generator state.  plan_fn(cfg, available, order, s0, params, rng, planner_state) -> dict(plan, env_steps, rng_after,
    tree, state_values, planner) in the ENVIRONMENT's action ids; `order` = the env's listing order (None = ascending);
    tree: creation-order arrays re-based at the root with `action`, `n_children`, `alive`.  Raises where the reference does."""
    import pytest
    p = "sa_masked/" + name
    cfg = mdp_from_golden(z, p + "/mdp")
    params = dict(budget=int(z[p + "/budget"]), gamma=float(z[p + "/gamma"]),
                  terminal_reward=float(z[p + "/terminal_reward"]), accuracy=float(z[p + "/accuracy"]),
                  backup_aggregated_nodes=bool(z[p + "/backup_aggregated_nodes"]),
                  prune_suboptimal_leaves=bool(z[p + "/prune_suboptimal_leaves"]))
    order = [1, 0, 2, 3, 4] if bool(z[p + "/listing_idle_first"]) else None
    rng = np.array(z[p + "/rng_before"], dtype=np.uint64)
    raises_at = int(z[p + "/raises_at_step"]) if p + "/raises_at_step" in z.files else -1
    planner, env_steps = None, 0
    n_steps = int(z[p + "/n_steps"])
    for step in range(n_steps):
        s0 = int(z[p + "/states"][step])
        if step == raises_at:
            with pytest.raises(ValueError):
                plan_fn(cfg, z[p + "/available"], order, s0, params, rng, planner)
            break
        out = plan_fn(cfg, z[p + "/available"], order, s0, params, rng, planner)
        q = "{}/step{}".format(p, step)
        np.testing.assert_array_equal(out["plan"], z[q + "/plan"], err_msg=q)
        np.testing.assert_array_equal(out["rng_after"], z[q + "/rng_after"], err_msg=q)
        env_steps += int(out["env_steps"])
        assert env_steps == int(z[q + "/env_steps"]), q
        tree = out["tree"]
        assert int(np.asarray(tree["alive"]).sum()) == int(z[q + "/n_leaves"]), q
        assert_keyed_tree_equal(z, q + "/tree", tree, dict(count="count", lower="lower", reward="reward", done="done",
                                                          depth="depth", obs="state", is_leaf="alive"))
        want = z[q + "/state_values"]
        seen = ~np.isnan(want)
        assert np.array_equal(out["state_values"][seen], want[seen]), q
        assert np.all(out["state_values"][~seen] == 1 / (1 - params["gamma"])), q
        planner, rng = out["planner"], out["rng_after"]
    assert raises_at >= 0 or n_steps > 0


def first_result(queue, procs, seconds):
    """The first item the ranks put on `queue` -- or a test failure, not a hang, when a rank dies or nothing arrives within
    `seconds` (a crashed rank leaves the others blocked in a collective forever; every process is killed then)."""
    import time
    import pytest
    deadline = time.time() + seconds
    while time.time() < deadline:
        if not queue.empty():
            return queue.get()
        if any(p.exitcode not in (None, 0) for p in procs):
            break
        time.sleep(0.02)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.kill()
    pytest.fail("the process group produced no result (exit codes {}; still running ones were killed)".format(codes))
