"""Pin the oracle on a batch of episodes with their own, per-step-changing tables (tests/golden/per_episode.npz: one
ValueIterationAgent / MCTSAgent / DeterministicPlannerAgent of the UNMODIFIED reference per episode, the episode's table
replaced before every step as a to_finite_mdp() re-extraction would: value_iteration.py:29-35)."""
import numpy as np

from oracle import oracle

E, T_STEPS = 6, 3


def episode_tables(z, t):
    """The tables of all episodes at step t: transition [E,S,A], reward, terminal [E,S]."""
    return z["transition"][:, t], z["reward"][:, t], z["terminal"][:, t]


def test_vi_each_episode_each_step(golden):
    z = golden["per_episode"]
    for t in range(T_STEPS):
        tr, rw, tm = episode_tables(z, t)
        q, sweeps = oracle.vi_solve_each(tr, rw, tm, gamma=float(z["vi/gamma"]), iterations=int(z["vi/iterations"]))
        for e in range(E):
            p = "vi/e{}/t{}".format(e, t)
            assert np.array_equal(q[e], z[p + "/Q"]), p
            assert int(sweeps[e]) == int(z[p + "/sweeps"]), p
            s = int(z["vi/e{}/states".format(e)][t])
            assert int(np.argmax(q[e][s])) == int(z[p + "/action"]), p


def test_uct_each_episode_each_step(golden):
    z = golden["per_episode"]
    a = z["reward"].shape[-1]
    p_uniform = np.ones(a) / a
    rng = np.stack([z["uct/e{}/rng_before".format(e)] for e in range(E)]).astype(np.uint64)
    total = np.zeros(E, np.int64)
    for t in range(T_STEPS):
        tr, rw, tm = episode_tables(z, t)
        s0 = [int(z["uct/e{}/states".format(e)][t]) for e in range(E)]
        out = oracle.uct_plan_each(tr, rw, tm, np.arange(E), s0, int(z["uct/episodes"]), int(z["uct/horizon"]),
                                   float(z["uct/gamma"]), float(z["uct/temperature"]), p_uniform, p_uniform, rng,
                                   max_plan_len=int(z["uct/horizon"]))
        rng = out["rng_after"]
        total += out["env_steps"]
        for e in range(E):
            p = "uct/e{}/t{}".format(e, t)
            np.testing.assert_array_equal(out["plans"][e, :out["plan_len"][e]], z[p + "/plan"], err_msg=p)
            np.testing.assert_array_equal(rng[e], z[p + "/rng_after"], err_msg=p)
            assert out["root_value"][e] == float(z[p + "/root_value"]), p
            assert int(total[e]) == int(z[p + "/env_steps_total"]), p


def test_opd_each_episode_each_step(golden):
    z = golden["per_episode"]
    rng = np.stack([z["opd/e{}/rng_before".format(e)] for e in range(E)]).astype(np.uint64)
    total = np.zeros(E, np.int64)
    for t in range(T_STEPS):
        tr, rw, tm = episode_tables(z, t)
        s0 = [int(z["opd/e{}/states".format(e)][t]) for e in range(E)]
        out = oracle.opd_plan_each(tr, rw, tm, np.arange(E), s0, int(z["opd/budget"]), float(z["opd/gamma"]), 0.0, rng,
                                   max_plan_len=64)
        rng = out["rng_after"]
        total += out["env_steps"]
        for e in range(E):
            p = "opd/e{}/t{}".format(e, t)
            np.testing.assert_array_equal(out["plans"][e, :out["plan_len"][e]], z[p + "/plan"], err_msg=p)
            np.testing.assert_array_equal(rng[e], z[p + "/rng_after"], err_msg=p)
            assert out["root_lower"][e] == float(z[p + "/root_lower"]) and out["root_upper"][e] == float(z[p + "/root_upper"]), p
            assert int(total[e]) == int(z[p + "/env_steps_total"]), p
