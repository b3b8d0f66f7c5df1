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


# ---------------------------------------------------------------------------------------------- MCTS with a VI prior, per episode
def boltzmann(q, temperature):
    z = np.exp((q - q.max(axis=1, keepdims=True)) / temperature)
    return z / z.sum(axis=1, keepdims=True)


def test_uct_with_vi_prior_each_episode_each_step(golden):
    """tests/golden/per_episode_prior.npz: the unmodified MCTSWithPriorPolicyAgent whose prior agent (the reference's
    ValueIterationAgent + action_distribution) re-solves value iteration on the table of every step; plain environments and
    environments with restricted action sets (distribution restricted to the listed actions and renormalised,
    mcts_with_prior.py:56-62).  Oracle: vi_solve -> Boltzmann table (numpy) -> uct_plan_batch with per-state policies."""
    from tests.helpers import restricted_agent_policy_lists
    z = golden["per_episode_prior"]
    n_e, n_t = 4, 3
    for name in ("plain", "masked"):
        available = z[name + "/available"] if name == "masked" else None
        rng = {e: z["{}/e{}/rng_before".format(name, e)].astype(np.uint64).reshape(1, 6) for e in range(n_e)}
        total = np.zeros(n_e, np.int64)
        for e in range(n_e):
            for t in range(int(z["{}/e{}/n_steps".format(name, e)])):
                p = "{}/e{}/t{}".format(name, e, t)
                tr, rw, tm = z[name + "/transition"][e, t], z[name + "/reward"][e, t], z[name + "/terminal"][e, t]
                q, _ = oracle.vi_solve("deterministic", tr, rw, tm, gamma=float(z["prior/gamma"]), iterations=int(z["prior/iterations"]))
                assert np.array_equal(q, z[p + "/q"]), p
                table = boltzmann(q, float(z["prior/temperature"]))
                assert np.array_equal(table, z[p + "/prior_table"]), p
                pol = table if available is None else restricted_agent_policy_lists(table, available)
                s = int(z["{}/e{}/states".format(name, e)][t])
                out = oracle.uct_plan_batch(tr, rw, tm, [s], int(z[name + "/episodes"]), int(z[name + "/horizon"]),
                                            float(z[name + "/gamma"]), float(z[name + "/temperature"]), pol, pol, rng[e],
                                            max_plan_len=int(z[name + "/horizon"]))
                rng[e] = out["rng_after"]
                total[e] += int(out["env_steps"][0])
                np.testing.assert_array_equal(out["plans"][0, :out["plan_len"][0]], z[p + "/plan"], err_msg=p)
                np.testing.assert_array_equal(rng[e][0], z[p + "/rng_after"], err_msg=p)
                assert out["root_value"][0] == float(z[p + "/root_value"]), p
                assert int(total[e]) == int(z[p + "/env_steps_total"]), p
