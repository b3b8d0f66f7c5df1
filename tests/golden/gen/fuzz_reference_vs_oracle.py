#!/usr/bin/env python3
"""Randomised check of the CPU oracle against the UNMODIFIED reference planners (build container only).

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/fuzz_reference_vs_oracle.py [n_cases] [seed]

The committed goldens pin the oracle on ~90 hand-picked cases; this draws random finite MDPs and random agent
configurations, runs the reference agent (through the stubs, like make_golden.py) and the oracle on the same inputs and
compares plan, env-step count and generator state after plan() -- plus the root statistics -- bit for bit.  Nothing is
stored; the run is recorded in DESIGN.md.  The oracle-vs-device sweep (tools/fuzz_parity.py) covers the other half.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path[:0] = [os.path.join(HERE, "stubs"), "/root/reference", REPO, HERE]

import numpy as np  # noqa: E402

from rl_agents.agents.common.factory import agent_factory  # noqa: E402
from oracle import oracle  # noqa: E402
from rl_agents_amd.envs import FiniteMDPEnv  # noqa: E402
from make_golden import rng_state  # noqa: E402

UCT = "<class 'rl_agents.agents.tree_search.mcts.MCTSAgent'>"
OPD = "<class 'rl_agents.agents.tree_search.deterministic.DeterministicPlannerAgent'>"
SAOPD = "<class 'rl_agents.agents.tree_search.state_aware.StateAwarePlannerAgent'>"
VI = "<class 'rl_agents.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"
UCTP = "<class 'rl_agents.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent'>"
PRIOR = "<class 'prior_agents.BoltzmannQAgent'>"


def random_mdp(g):
    s = int(g.choice([2, 3, 7, 30, 120]))
    a = int(g.choice([2, 3, 4, 5, 7]))
    t = g.integers(0, s, size=(s, a), dtype=np.int64)
    kind = g.integers(0, 3)
    r = g.random((s, a)) if kind == 0 else (g.integers(0, 2, size=(s, a)).astype(float) if kind == 1 else np.round(g.random((s, a)), 1))
    term = g.random(s) < g.choice([0.0, 0.1, 0.4])
    return t, r, term


def env_of(t, r, term, state, max_steps, done_rule):
    cfg = dict(mode="deterministic", transition=t.tolist(), reward=r.tolist(), terminal=term.astype(int).tolist(), state=int(state),
               done_rule=done_rule)
    if max_steps:
        cfg["max_steps"] = max_steps
    env = FiniteMDPEnv(cfg)
    env.reset()
    return env


def one_case(g, case):
    t, r, term = random_mdp(g)
    s, a = r.shape
    s0 = int(g.integers(0, s))
    gamma = float(g.choice([0.3, 0.8, 0.95]))
    done_rule = "next" if g.random() < 0.3 else "source"
    seed = int(g.integers(0, 1000))
    kind = ["uct", "opd", "saopd", "vi", "uct_prior"][int(g.integers(0, 5))]
    desc = dict(case=case, kind=kind, S=s, A=a, s0=s0, gamma=gamma, done_rule=done_rule, seed=seed)
    if kind == "uct":
        max_steps = int(g.choice([0, 0, 5]))
        steps0 = int(g.integers(0, 3)) if max_steps else 0
        env = env_of(t, r, term, s0, max_steps, done_rule)
        env.steps = steps0
        cfg = dict(__class__=UCT, budget=int(g.choice([20, 100, 300])), gamma=gamma, temperature=float(g.choice([0.5, 10.0, 200.0])))
        if g.random() < 0.5:
            cfg.update(horizon=int(g.choice([3, 9])), episodes=int(g.choice([4, 15])))
        if g.random() < 0.4:
            pref = {"type": "preference", "action": int(g.integers(0, a)), "ratio": float(g.choice([2, 5]))}
            cfg.update(prior_policy=pref, rollout_policy=pref)
        desc.update(cfg=cfg, max_steps=max_steps, steps0=steps0)
        agent = agent_factory(env, cfg)
        agent.seed(seed)
        st0 = rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        pc = agent.planner.config
        _, prior_p = agent.planner.prior_policy(env, None)
        _, roll_p = agent.planner.rollout_policy(env, None)
        o = oracle.uct_plan(t, r, term, s0, pc["episodes"], pc["horizon"], pc["gamma"], pc["temperature"], prior_p, roll_p, st0,
                            steps0=steps0, max_steps=max_steps, done_rule=done_rule, max_plan_len=pc["horizon"] + 1)
        assert list(plan) == list(o["plan"]), (desc, plan, o["plan"])
        assert len(agent.planner.observations) == o["env_steps"], desc
        assert np.array_equal(rng_state(agent.planner.np_random), o["rng_after"]), desc
        assert agent.planner.root.count == o["tree"]["count"][0] and float(agent.planner.root.value) == o["tree"]["value"][0], desc
    elif kind == "uct_prior":
        env = env_of(t, r, term, s0, 0, done_rule)
        cfg = dict(__class__=UCTP, budget=int(g.choice([20, 100, 300])), gamma=gamma, temperature=float(g.choice([0.5, 10.0, 200.0])),
                   prior_agent=dict(__class__=PRIOR, gamma=float(g.choice([0.5, 0.9])), temperature=float(g.choice([0.1, 1.0])),
                                    mask=int(g.choice([0, 3]))))
        if g.random() < 0.5:
            cfg.update(horizon=int(g.choice([3, 9])), episodes=int(g.choice([4, 15])))
        desc.update(cfg=cfg)
        agent = agent_factory(env, cfg)
        agent.seed(seed)
        st0 = rng_state(agent.planner.np_random)
        table = np.array(agent.prior_agent.table)
        plan = agent.plan(s0)
        pc = agent.planner.config
        o = oracle.uct_plan(t, r, term, s0, pc["episodes"], pc["horizon"], pc["gamma"], pc["temperature"], table, table, st0,
                            done_rule=done_rule, max_plan_len=pc["horizon"] + 1)
        assert list(plan) == list(o["plan"]), (desc, plan, o["plan"])
        assert len(agent.planner.observations) == o["env_steps"], desc
        assert np.array_equal(rng_state(agent.planner.np_random), o["rng_after"]), desc
        assert agent.planner.root.count == o["tree"]["count"][0] and float(agent.planner.root.value) == o["tree"]["value"][0], desc
    elif kind == "opd":
        env = env_of(t, r, term, s0, 0, done_rule)
        cfg = dict(__class__=OPD, budget=int(g.choice([a - 1, 3 * a, 60, 250])), gamma=gamma, terminal_reward=float(g.choice([0.0, 0.5])))
        desc.update(cfg=cfg)
        agent = agent_factory(env, cfg)
        agent.seed(seed)
        st0 = rng_state(agent.planner.np_random)
        plan = agent.plan(s0)
        o = oracle.opd_plan(t, r, term, s0, cfg["budget"], gamma, cfg["terminal_reward"], st0, done_rule=done_rule,
                            max_plan_len=cfg["budget"] + 1)
        assert list(plan) == list(o["plan"]), (desc, plan, o["plan"])
        assert np.array_equal(rng_state(agent.planner.np_random), o["rng_after"]), desc
        root = agent.planner.root
        assert float(root.value_lower) == o["tree"]["lower"][0] and float(root.value_upper) == o["tree"]["upper"][0], desc
    elif kind == "saopd":
        env = env_of(t, r, term, s0, 0, done_rule)
        cfg = dict(__class__=SAOPD, budget=int(g.choice([3 * a, 60, 150])), gamma=gamma, terminal_reward=float(g.choice([0.0, 0.5])),
                   accuracy=float(g.choice([0.0, 0.0, 0.1])), backup_aggregated_nodes=bool(g.random() < 0.8),
                   prune_suboptimal_leaves=bool(g.random() < 0.8))
        desc.update(cfg=cfg)
        agent = agent_factory(env, cfg)
        agent.seed(seed)
        rng, planner = rng_state(agent.planner.np_random), None
        kw = {k: cfg[k] for k in ("accuracy", "backup_aggregated_nodes", "prune_suboptimal_leaves")}
        for step in range(3):
            state = env.mdp.state
            try:
                plan = agent.plan(state)
            except ValueError as e:
                assert "empty" in str(e), desc
                try:
                    oracle.saopd_plan(t, r, term, state, cfg["budget"], gamma, cfg["terminal_reward"], rng, planner, done_rule=done_rule,
                                      max_plan_len=cfg["budget"] + 1, **kw)
                except ValueError:
                    break
                raise AssertionError(("the reference raises, the oracle does not", desc))
            o = oracle.saopd_plan(t, r, term, state, cfg["budget"], gamma, cfg["terminal_reward"], rng, planner, done_rule=done_rule,
                                  max_plan_len=cfg["budget"] + 1, **kw)
            assert list(plan) == list(o["plan"]), (desc, step, plan, o["plan"])
            assert np.array_equal(rng_state(agent.planner.np_random), o["rng_after"]), (desc, step)
            assert len(agent.planner.leaves) == int(o["tree"]["alive"].sum()), (desc, step)
            for key, val in agent.planner.state_values.items():
                assert float(val) == o["state_values"][int(key)], (desc, step, key)
            rng, planner = o["rng_after"], o["planner"]
            _, _, term_, trunc_, _ = env.step(plan[0])
            if term_ or trunc_ or not plan:
                break
    else:
        rewards = r * float(g.choice([1.0, -1.0, 10.0]))
        env = env_of(t, rewards, term, s0, 0, "source")
        cfg = dict(__class__=VI, gamma=float(g.choice([gamma, 1.0])), iterations=int(g.choice([1, 10, 100])))
        desc.update(cfg=cfg)
        agent = agent_factory(env, cfg)
        q, _ = oracle.vi_solve("deterministic", t, rewards, term, gamma=cfg["gamma"], iterations=cfg["iterations"])
        assert np.array_equal(np.asarray(agent.state_action_value), q), desc
        assert np.array_equal(agent.get_state_value(), oracle.vi_solve("deterministic", t, rewards, term, gamma=cfg["gamma"],
                                                                         iterations=cfg["iterations"], state_value=True)), desc
    return kind


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    g = np.random.Generator(np.random.PCG64(seed))
    kinds = {}
    for case in range(n):
        k = one_case(g, case)
        kinds[k] = kinds.get(k, 0) + 1
    print("ok", kinds)


if __name__ == "__main__":
    main()
