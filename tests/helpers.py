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
