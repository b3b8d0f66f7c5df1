"""The oracle against the round-3 golden vectors (tests/golden/round3.npz, made by the unmodified reference through
tests/golden/gen/make_golden_round3.py): the checker is pinned before the device is compared with it."""
import os

import numpy as np
import pytest

from tests.helpers import assert_keyed_tree_equal, mdp_from_golden

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(REPO, "tests", "golden", "round3.npz"))


@pytest.fixture(scope="module")
def zvi():
    return np.load(os.path.join(REPO, "tests", "golden", "vi.npz"))


def names(z, group):
    return [str(n) for n in z[group + "/names"]]


def test_robust_state_value_goldens(z, zvi):
    """RobustValueIterationAgent.get_state_value (robust_value_iteration.py:32-37): bit-exact for deterministic models,
    1e-12 for dense ones (numpy's pairwise order is restated, the tolerance is the dense-mode contract)."""
    from oracle import oracle
    for name in names(z, "rvi_v"):
        p = "rvi/" + name
        mode = str(zvi[p + "/mode"])
        v = oracle.vi_solve(mode, zvi[p + "/transitions"], zvi[p + "/rewards"], None, gamma=float(zvi[p + "/gamma"]),
                            iterations=int(zvi[p + "/iterations"]), robust=True, state_value=True)
        assert np.array_equal(v, z["rvi_v/{}/V".format(name)]), name


def robust_models(z, p):
    m = int(z[p + "/n_models"])
    cfgs = [mdp_from_golden(z, "{}/mdp{}".format(p, i)) for i in range(m)]
    return (np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs]),
            np.stack([c["terminal"] for c in cfgs]))


def test_robust_planner_restricted_actions_goldens(z):
    from oracle import oracle
    for name in names(z, "robust_masked"):
        p = "robust_masked/" + name
        t, r, term = robust_models(z, p)
        m = t.shape[0]
        out = oracle.ropd_plan(t, r, term, [int(z[p + "/s0"])] * m, int(z[p + "/budget"]), float(z[p + "/gamma"]),
                               float(z[p + "/terminal_reward"]), rng_state=z[p + "/rng_before"], available=z[p + "/available"])
        np.testing.assert_array_equal(out["plan"], z[p + "/plan"], err_msg=name)
        assert out["root_lower"] == float(z[p + "/root_lower"]) and out["root_upper"] == float(z[p + "/root_upper"]), name
        assert out["env_steps"] == int(z[p + "/env_steps"]), name
        np.testing.assert_array_equal(out["rng_after"], z[p + "/rng_after"], err_msg=name)
        tree = dict(out["tree"])
        tree["obs"] = np.where(np.arange(len(tree["parent"]))[:, None] == 0, -1, tree["state"])
        tree["lower_min"], tree["upper_min"] = tree["lower"].min(axis=1), tree["upper"].min(axis=1)
        assert_keyed_tree_equal(z, p + "/tree", tree, dict(count="count", depth="depth", lower_min="lower_min",
                                                          upper_min="upper_min", reward="reward", done="done", obs="obs",
                                                          n_children="n_children"))
