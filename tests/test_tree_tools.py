"""Consumers of an exported tree, CPU only: the Node API of this package -- breadth_first_search,
get_trajectories, get_obs_visits, path / sequence, TreePlot -- on trees rebuilt from the reference's own plans, against
what the UNMODIFIED reference returned for those trees (tests/golden/tree_tools.npz, tests/golden/gen/make_golden_trees.py),
and, in the build container, the reference's TreePlot drawing this package's Node tree."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.fixture(scope="module")
def trees():
    return np.load(os.path.join(REPO, "tests", "golden", "tree_tools.npz"))


def rebuild(z, name):
    """This package's Node tree from a golden BFS listing, as export_tree builds it from device arrays."""
    from rl_agents_amd.agents.tree_search.abstract import build_tree
    from rl_agents_amd.agents.tree_search.deterministic import with_observations
    p = "trees/" + name
    arrays = {k: z["{}/tree/{}".format(p, k)] for k in ("parent", "action", "count")}
    uct = bool(z[p + "/is_uct"])
    arrays["v"] = z[p + "/tree/value"] if uct else z[p + "/tree/lower"]
    tree = build_tree(arrays, "v", transition=z[p + "/mdp/transition"], root_state=int(z[p + "/s0"]))
    return tree if uct else with_observations(tree)


def unragged(flat, offs):
    return [[int(x) for x in flat[offs[i]:offs[i + 1]]] for i in range(len(offs) - 1)]


def names(z):
    return [str(n) for n in z["trees/names"]]


def test_obs_visits_equal_the_reference(trees):
    for name in names(trees):
        p = "trees/" + name
        visits, updates = rebuild(trees, name).get_obs_visits(state=None)
        ref = dict(zip([str(k) for k in trees[p + "/visit_keys"]], [int(c) for c in trees[p + "/visit_counts"]]))
        assert dict(visits) == ref, name
        assert len(updates) == int(trees[p + "/n_updates"])


def test_trajectories_equal_the_reference(trees):
    for name in names(trees):
        p = "trees/" + name
        root = rebuild(trees, name)
        full = root.get_trajectories(full_trajectories=True, include_leaves=True)
        assert len(full) == int(trees[p + "/n_full"])
        assert [len(t) for t in full] == [int(x) for x in trees[p + "/full_lengths"]]
        assert all(t[0] is root for t in full)
        flat = root.get_trajectories(full_trajectories=False, include_leaves=False)
        assert [n.count for n in flat] == [int(x) for x in trees[p + "/flat_counts"]]
        assert flat[-1] is root                                        # children first, the node itself last
        assert len(root.get_trajectories(full_trajectories=False, include_leaves=True)) == int(trees[p + "/n_flat_with_leaves"])
        assert root.get_trajectories(full_trajectories=True, include_leaves=False) == []


def test_breadth_first_search_equals_the_reference(trees):
    from rl_agents_amd.agents.tree_search.abstract import Node
    for name in names(trees):
        p = "trees/" + name
        root = rebuild(trees, name)
        paths = [path for _, path in Node.breadth_first_search(root)]
        assert paths == unragged(trees[p + "/bfs_paths"], trees[p + "/bfs_offsets"])
        leaves = [path for _, path in Node.breadth_first_search(root, condition=lambda n: n.is_leaf())]
        assert leaves == unragged(trees[p + "/bfs_leaf_paths"], trees[p + "/bfs_leaf_offsets"])
        counts = list(Node.breadth_first_search(root, operator=lambda n, path: n.count))
        assert counts == [int(c) for c in trees[p + "/bfs_counts"]]
        for node, path in Node.breadth_first_search(root):              # path() / sequence() agree with the traversal
            assert node.path() == path and len(node.sequence()) == len(path) + 1 and node.sequence()[0] is root


def test_planner_visits_of_the_optimistic_planner(trees):
    """OptimisticDeterministicPlanner.get_visits from the tree = the reference's log of stepped observations."""
    from rl_agents_amd.agents.tree_search.deterministic import OptimisticDeterministicPlanner
    for name in [n for n in names(trees) if n.startswith("opd")]:
        p = "trees/" + name
        planner = OptimisticDeterministicPlanner.__new__(OptimisticDeterministicPlanner)
        planner._root, planner.last = rebuild(trees, name), {}
        ref = dict(zip([str(k) for k in trees[p + "/planner_visit_keys"]], [int(c) for c in trees[p + "/planner_visit_counts"]]))
        assert dict(planner.get_visits()) == ref


def test_tree_plot_segments(trees):
    """TreePlot: one segment per visited child within max_depth, widths within [0.5, 4], the root's children fanned out
    over [-0.5, 0.5]."""
    import types
    from rl_agents_amd.agents.tree_search.graphics import TreePlot
    root = rebuild(trees, "uct_highway_small")
    planner = types.SimpleNamespace(root=root, env=types.SimpleNamespace(action_space=types.SimpleNamespace(n=5)))
    plot = TreePlot(planner, max_depth=6)
    segs = plot.segments()
    visited = sum(1 for node, path in root.breadth_first_search(root)
                  if node.parent is not None and 0 < len(path) <= 7 and all(n.count for n in node.sequence()[1:]))
    assert len(segs) == visited and all(0.5 <= s[4] <= 4.0 for s in segs)
    level1 = sorted(s[2] for s in segs if s[0] == 0.0 and s[1] == 0.0)
    assert level1[0] >= -0.5 and level1[-1] <= 0.5
    writer = types.SimpleNamespace(images=[])
    writer.add_image = lambda title, image, epoch: writer.images.append((title, image.shape, epoch))
    import matplotlib
    matplotlib.use("Agg")
    image = plot.plot_to_writer(writer, epoch=3, show=True)
    assert writer.images == [("Expanded_tree", image.shape, 3)] and image.shape[0] == 3 and image.dtype == np.uint8


SCRIPT = r'''
import os, sys, types
sys.dont_write_bytecode = True
repo, ref = sys.argv[1], sys.argv[2]
sys.path[:0] = [os.path.join(repo, "tests", "golden", "gen", "stubs"), ref, repo]
import matplotlib
matplotlib.use("Agg")
import matplotlib.pyplot as plt
import numpy as np
from rl_agents.agents.tree_search.graphics import TreePlot as RefTreePlot          # the REFERENCE's plot
from rl_agents_amd.agents.tree_search.graphics import TreePlot
sys.path.insert(0, os.path.join(repo, "tests"))
from test_tree_tools import rebuild
z = np.load(os.path.join(repo, "tests", "golden", "tree_tools.npz"))
for name, n_actions in (("uct_highway_small", 5), ("opd_grid_c1", 4)):
    root = rebuild(z, name)
    planner = types.SimpleNamespace(root=root, env=types.SimpleNamespace(action_space=types.SimpleNamespace(n=n_actions)))
    fig, ax = plt.subplots()
    RefTreePlot(planner, max_depth=6).plot(filename=None, ax=ax)                    # reads planner.root / children / count
    theirs = sorted((round(float(l.get_xdata()[0]), 9), round(float(l.get_ydata()[0]), 9), round(float(l.get_xdata()[1]), 9),
                     round(float(l.get_ydata()[1]), 9), round(float(l.get_linewidth()), 9)) for l in ax.lines)
    ours = sorted(tuple(round(float(v), 9) for v in s) for s in TreePlot(planner, max_depth=6).segments())
    assert len(theirs) > 10 and theirs == ours, (name, len(theirs), len(ours))
print("ok")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference only exists in the build container")
def test_reference_tree_plot_draws_this_packages_tree():
    """The reference's TreePlot (tree_search/graphics.py:115-147) takes the exported Node tree as it is, and draws the very
    segments this package's TreePlot lists."""
    out = subprocess.run([sys.executable, "-c", SCRIPT, REPO, REF], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
