"""The CPU oracle against the round-2 golden vectors of the UNMODIFIED reference (tests/golden/variants.npz, made by
tests/golden/gen/make_golden_variants.py): closed-loop MCTS, planners on environments that restrict the available
actions, and the discrete robust planner.  Bit for bit: plans, trees, env-step counts, generator states."""
import json
import os

import numpy as np
import pytest

from tests.helpers import (assert_keyed_tree_equal, mdp_from_golden, reference_policy_lists,
                           restricted_agent_policy_lists)

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(REPO, "tests", "golden", "variants.npz"))


def names(z, group):
    return [str(n) for n in z[group + "/names"]]


UCT_FIELDS = dict(count="count", value="value", prior="prior", is_obs="is_obs")


def _uct_params(z, p):
    return dict(episodes=int(z[p + "/episodes"]), horizon=int(z[p + "/horizon"]), gamma=float(z[p + "/gamma"]),
                temperature=float(z[p + "/temperature"]))


def test_closed_loop_mcts(z):
    from oracle import oracle
    for name in names(z, "closed"):
        p = "closed/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        k = _uct_params(z, p)
        out = oracle.uct_plan(cfg["transition"], cfg["reward"], cfg["terminal"], int(z[p + "/s0"]), k["episodes"],
                              k["horizon"], k["gamma"], k["temperature"], z[p + "/prior_p"], z[p + "/rollout_p"],
                              z[p + "/rng_before"], steps0=int(z[p + "/steps0"]), max_steps=cfg["max_steps"],
                              max_plan_len=2 * k["horizon"] + 2, closed_loop=True)
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        assert_keyed_tree_equal(z, p + "/tree", out["tree"], UCT_FIELDS)
        # plan entries alternate action / observation key
        assert np.array_equal(z[p + "/plan_is_obs"], np.arange(len(out["plan"])) % 2 == 1), name


def _masked_uct(z, p, prior, rollout, closed_loop=False, init_tree=None, rng=None):
    from oracle import oracle
    cfg = mdp_from_golden(z, p.rsplit("/step", 1)[0] + "/mdp" if "/step" in p else p + "/mdp")
    return cfg


def test_mcts_with_restricted_actions(z):
    from oracle import oracle
    for name in names(z, "uct_masked"):
        p = "uct_masked/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        k = _uct_params(z, p)
        avail = z[p + "/available"]
        prior = reference_policy_lists(json.loads(str(z[p + "/prior_policy_json"])), avail)
        rollout = reference_policy_lists(json.loads(str(z[p + "/rollout_policy_json"])), avail)
        closed = bool(z[p + "/closed_loop"])
        out = oracle.uct_plan(cfg["transition"], cfg["reward"], cfg["terminal"], int(z[p + "/s0"]), k["episodes"],
                              k["horizon"], k["gamma"], k["temperature"], prior, rollout, z[p + "/rng_before"],
                              max_steps=cfg["max_steps"], max_plan_len=2 * k["horizon"] + 2, closed_loop=closed)
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        assert_keyed_tree_equal(z, p + "/tree", out["tree"], UCT_FIELDS)


def test_mcts_restricted_actions_subtree_episode(z):
    """step_strategy "subtree" on a restricted-action env: the kept tree carries per-node child lists."""
    from oracle import oracle
    p = "uct_masked/subtree_highway"
    cfg = mdp_from_golden(z, p + "/mdp")
    avail = z[p + "/available"]
    pol = reference_policy_lists({"type": "random_available"}, avail)
    k = _uct_params(z, p)
    rng, kept = z[p + "/rng_before"], None
    for step in range(int(z[p + "/n_steps"])):
        q = "{}/step{}".format(p, step)
        out = oracle.uct_plan(cfg["transition"], cfg["reward"], cfg["terminal"], int(z[p + "/states"][step]),
                              k["episodes"], k["horizon"], k["gamma"], k["temperature"], pol, pol, rng,
                              max_plan_len=k["horizon"], init_tree=kept)
        np.testing.assert_array_equal(out["plan"], z[q + "/plan"], err_msg=q)
        np.testing.assert_array_equal(out["rng_after"], z[q + "/rng_after"], err_msg=q)
        assert_keyed_tree_equal(z, q + "/tree", out["tree"], dict(count="count", value="value", prior="prior"))
        rng = out["rng_after"]
        kept = oracle.uct_reroot(out["tree"], int(out["plan"][0]), cfg["reward"].shape[1])


def test_mcts_with_prior_agent_and_restricted_actions(z):
    from oracle import oracle
    for name in names(z, "uct_prior_masked"):
        p = "uct_prior_masked/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        k = _uct_params(z, p)
        pol = restricted_agent_policy_lists(z[p + "/prior_table"], z[p + "/available"])
        out = oracle.uct_plan(cfg["transition"], cfg["reward"], cfg["terminal"], int(z[p + "/s0"]), k["episodes"],
                              k["horizon"], k["gamma"], k["temperature"], pol, pol, z[p + "/rng_before"],
                              max_steps=cfg["max_steps"], max_plan_len=k["horizon"])
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        assert_keyed_tree_equal(z, p + "/tree", out["tree"], dict(count="count", value="value", prior="prior"))


def test_opd_with_restricted_actions(z):
    from oracle import oracle
    for name in names(z, "opd_masked"):
        p = "opd_masked/" + name
        cfg = mdp_from_golden(z, p + "/mdp")
        out = oracle.opd_plan(cfg["transition"], cfg["reward"], cfg["terminal"], int(z[p + "/s0"]), int(z[p + "/budget"]),
                              float(z[p + "/gamma"]), float(z[p + "/terminal_reward"]), z[p + "/rng_before"],
                              available=z[p + "/available"])
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["root_lower"] == float(z[p + "/root_lower"]) and out["root_upper"] == float(z[p + "/root_upper"]), name
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        tree = out["tree"]
        tree["obs"] = np.where(np.arange(len(tree["state"])) == 0, -1, tree["state"])
        assert_keyed_tree_equal(z, p + "/tree", tree, dict(count="count", lower="lower", upper="upper", reward="reward",
                                                          done="done", depth="depth", obs="obs"))


def robust_models(z, p):
    m = int(z[p + "/n_models"])
    cfgs = [mdp_from_golden(z, "{}/mdp{}".format(p, i)) for i in range(m)]
    return (np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs]),
            np.stack([c["terminal"] for c in cfgs]))


ROBUST_FIELDS = dict(count="count", depth="depth", lower="lower", upper="upper", reward="reward", done="done")


def test_discrete_robust_planner(z):
    """DiscreteRobustPlanner / RobustNode (agents/robust/robust.py:28-50): vector bounds per leaf, min over models."""
    from oracle import oracle
    for name in names(z, "robust"):
        p = "robust/" + name
        t, r, term = robust_models(z, p)
        out = oracle.ropd_plan(t, r, term, int(z[p + "/s0"]), int(z[p + "/budget"]), float(z[p + "/gamma"]),
                               float(z[p + "/terminal_reward"]), z[p + "/rng_before"])
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["root_lower"] == float(z[p + "/root_lower"]) and out["root_upper"] == float(z[p + "/root_upper"]), name
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        tree = out["tree"]
        assert_keyed_tree_equal(z, p + "/tree", tree, ROBUST_FIELDS)
        from tests.helpers import bfs_children
        order, _ = bfs_children(tree["first_child"], tree["n_children"])
        assert np.array_equal(tree["state"][order][1:], z[p + "/tree/obs"][1:]), name
        assert np.array_equal(tree["lower"].min(axis=1)[order], z[p + "/tree/lower_min"])
    assert bool(z["robust/trap_raises_valueerror"])
    with pytest.raises(ValueError):
        trap_t = np.array([[[1, 2], [1, 1], [3, 4], [3, 3], [4, 4]]] * 2)
        trap_r = np.array([[[0, 0], [0, 0], [0, 0], [1, 1], [-1, -1]]] * 2, dtype=float)
        oracle.ropd_plan(trap_t, trap_r, np.array([[0, 1, 0, 1, 1]] * 2), 0, 20, 0.8)
