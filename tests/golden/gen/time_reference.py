#!/usr/bin/env python3
"""Time the UNMODIFIED Python reference on the bench.py tables -> profiles/reference_cpu.json.

`cpu_baseline` in bench.py is the C port of the reference (oracle/planning_oracle.c) timed live on the GPU box's
host; the tier also asks for the reference's own CPU path beside the GPU number.  /root/reference does not travel
to the GPU box, so this script -- run in the BUILD CONTAINER, the only place the reference exists -- times the real
thing on the same tables, seeds and planner parameters and commits the record; bench.py prints it as
`cpu_baseline.reference_python` with the host it was measured on.

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/gen/time_reference.py [workload ...]

Reference entry points timed: MCTSAgent.plan (tree_search/mcts.py:179-184), DeterministicPlannerAgent
.plan (deterministic.py:116-122), StateAwarePlannerAgent.plan (state_aware.py:117-127), ValueIterationAgent /
RobustValueIterationAgent.get_state_action_value (value_iteration.py:42-45, robust_value_iteration.py:39-44).
One core = one Python process (the reference is single-threaded); "all cores" = the reference's own fan-out style,
one process per experiment (scripts/experiments.py:102-106), here multiprocessing.Pool over roots.
"""
import json
import multiprocessing
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
REF = "/root/reference"
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, REPO, HERE]

import numpy as np  # noqa: E402

UCT = "<class 'rl_agents.agents.tree_search.mcts.MCTSAgent'>"
UCTP = "<class 'rl_agents.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent'>"
PRIOR = "<class 'prior_agents.BoltzmannQAgent'>"
OPD = "<class 'rl_agents.agents.tree_search.deterministic.DeterministicPlannerAgent'>"
SAOPD = "<class 'rl_agents.agents.tree_search.state_aware.StateAwarePlannerAgent'>"
VI = "<class 'rl_agents.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"
RVI = "<class 'rl_agents.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent'>"


def make_env(cfg, state=0):
    from rl_agents_amd.envs import FiniteMDPEnv
    c = {k: v for k, v in cfg.items() if k != "original_shape"}
    c["state"] = int(state)
    env = FiniteMDPEnv(c)
    env.reset()
    return env


def bench_roots(term, n, seed=12345):
    non_term = np.flatnonzero(~np.asarray(term))
    return np.random.Generator(np.random.PCG64(seed)).choice(non_term, size=n).astype(np.int32)


def headline():
    from rl_agents_amd.envs import generators
    return generators.highway_shaped(10, 10, 100, seed=0)


def _plan_roots(args):
    """Worker: plan `roots` one after the other with a fresh agent each (what N independent episodes would do)."""
    kind, roots, seed0 = args
    from rl_agents.agents.common.factory import agent_factory
    steps, spent = 0, 0.0           # only plan() is timed: env / agent construction is not the path

    def timed_plan(agent, obs):
        t = time.perf_counter()
        agent.plan(obs)
        return time.perf_counter() - t
    if kind in ("uct", "uct_prior"):
        cfg = headline()
        for i, s0 in enumerate(roots):
            env = make_env(cfg, state=s0)
            acfg = dict(__class__=UCT, budget=1000, horizon=30, episodes=33)
            if kind == "uct_prior":
                acfg = dict(acfg, __class__=UCTP, prior_agent=dict(__class__=PRIOR, gamma=0.95, iterations=200, temperature=0.3))
            agent = agent_factory(env, acfg)
            agent.seed(seed0 + i)
            spent += timed_plan(agent, int(s0))
            steps += len(agent.planner.observations)
    elif kind == "uct_cartpole":
        from rl_agents_amd.envs import CartPoleEnv
        for i, s0 in enumerate(roots):
            env = CartPoleEnv()
            env.seed(int(s0))
            env.reset()
            agent = agent_factory(env, dict(__class__=UCT, budget=1000, horizon=50, episodes=20))
            agent.seed(seed0 + i)
            spent += timed_plan(agent, None)
            steps += len(agent.planner.observations)
    elif kind == "opd":
        cfg = headline()
        for i, s0 in enumerate(roots):
            agent = agent_factory(make_env(cfg, state=s0), dict(__class__=OPD, budget=5000, gamma=0.8))
            agent.seed(seed0 + i)
            spent += timed_plan(agent, int(s0))
            steps += len(agent.planner.observations)
    elif kind == "saopd":
        from rl_agents_amd.envs import generators
        cfg = generators.gridworld()
        for i, s0 in enumerate(roots):
            agent = agent_factory(make_env(cfg, state=s0), dict(__class__=SAOPD, budget=500, gamma=0.8))
            agent.seed(seed0 + i)
            spent += timed_plan(agent, int(s0))
            steps += len(agent.planner.observations)
    return steps, spent


def time_planner(kind, roots_1core, roots_per_proc, cores):
    roots_all = bench_roots(headline()["terminal"], 4096) if kind in ("uct", "uct_prior", "opd") else \
        np.random.Generator(np.random.PCG64(12345)).integers(0, 100, size=4096).astype(np.int32)
    steps1, dt1 = _plan_roots((kind, roots_all[:roots_1core], 0))
    out = dict(value_1core=steps1 / dt1, unit="env-steps/s", plan_ms_per_root_1core=1e3 * dt1 / roots_1core,
               sample_1core="{} roots, one process".format(roots_1core))
    if cores > 1 and roots_per_proc > 0:
        jobs = [(kind, roots_all[64 + p * roots_per_proc: 64 + (p + 1) * roots_per_proc], 1000 * (p + 1)) for p in range(cores)]
        t0 = time.perf_counter()
        with multiprocessing.Pool(processes=cores) as pool:
            res = pool.map(_plan_roots, jobs)
        wall = time.perf_counter() - t0
        dt = max(r[1] for r in res)          # the slowest worker's plan() time: construction and pool start-up excluded
        out.update(value=sum(r[0] for r in res) / dt, cores=cores, pool_wall_s=wall,
                   sample="{} roots over a multiprocessing.Pool of {} processes, time = slowest worker's sum of plan() "
                          "calls".format(cores * roots_per_proc, cores))
    else:
        out.update(value=out["value_1core"], cores=1, sample=out["sample_1core"])
    return out


def time_vi(kind):
    from rl_agents.agents.common.factory import agent_factory
    from rl_agents_amd.envs import generators
    if kind == "vi":
        cfg = headline()
        agent = agent_factory(make_env(cfg), dict(__class__=VI, gamma=0.95, iterations=1))
        agent.config["iterations"] = 200
        sweeps, note = 200, "highway-shaped S=10000 A=5 deterministic, 200 sweeps requested (allclose exit as in the reference)"
    elif kind == "rvi":
        cfg = generators.highway_shaped(10, 50, 100, seed=2)
        cfg2 = generators.rewire(cfg, 0.1, seed=3)
        models = [dict(mode="deterministic", transition=cfg["transition"].tolist(), reward=cfg["reward"].tolist()),
                  dict(mode="deterministic", transition=cfg2["transition"].tolist(), reward=(cfg2["reward"] * 0.97).tolist())]
        env = make_env(dict(mode="deterministic", transition=[[0]], reward=[[0.0]]))
        agent = agent_factory(env, dict(__class__=RVI, gamma=0.95, iterations=200, models=models))
        sweeps, note = 200, "intersection-shaped S=50000 A=5 M=2 deterministic, 200 sweeps requested"
    else:  # vi_dense: the reference materialises (S, A, S) temporaries -- time S = 2000 and scale by (S / 10000)^2
        s_cpu = 2000
        g = np.random.Generator(np.random.PCG64(0))
        t = g.random((s_cpu, 5, s_cpu))
        t /= t.sum(-1, keepdims=True)
        cfg = dict(mode="stochastic", transition=t, reward=g.random((s_cpu, 5)))
        agent = agent_factory(make_env(cfg), dict(__class__=VI, gamma=0.95, iterations=1))
        agent.config["iterations"] = 5
        sweeps, note = 5, "dense S=2000 A=5 (160 MB), 5 sweeps, scaled by (2000 / 10000)^2 to S = 10000"
    # count the sweeps really run (the reference stops on np.allclose)
    t0 = time.perf_counter()
    q = agent.get_state_action_value()
    dt = time.perf_counter() - t0
    run, value = 0, np.zeros(np.shape(q))
    for _ in range(agent.config["iterations"]):
        nxt = (agent.worst_case(agent.bellman_expectation(agent.best_action_value(value))) if kind == "rvi"
               else agent.bellman_expectation(agent.best_action_value(value)))
        run += 1
        if np.allclose(value, nxt):
            break
        value = nxt
    rate = run / dt
    if kind == "vi_dense":
        rate *= (2000 / 10000.0) ** 2
    return dict(value=rate, value_1core=rate, unit="sweeps/s", cores=1, sweeps_run=run, seconds=dt, sample=note)


def main():
    only = sys.argv[1:]
    cores = len(os.sched_getaffinity(0))
    cpu = "unknown"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            cpu = line.split(":", 1)[1].strip()
            break
    path = os.path.join(REPO, "profiles", "reference_cpu.json")
    rec = dict(workloads={})
    if os.path.exists(path):
        rec = json.load(open(path))
    rec["host"] = dict(where="build container (no GPU)", cpu=cpu, cores=cores, python=platform.python_version(),
                       numpy=np.__version__)
    plan = [("uct", lambda: time_planner("uct", 16, 16, cores)),
            ("uct_prior", lambda: time_planner("uct_prior", 8, 0, 1)),
            ("uct_cartpole", lambda: time_planner("uct_cartpole", 16, 16, cores)),
            ("saopd", lambda: time_planner("saopd", 16, 16, cores)),
            ("opd", lambda: time_planner("opd", 1, 1, cores)),
            ("vi", lambda: time_vi("vi")), ("rvi", lambda: time_vi("rvi")), ("vi_dense", lambda: time_vi("vi_dense"))]
    for name, fn in plan:
        if only and name not in only:
            continue
        t0 = time.perf_counter()
        rec["workloads"][name] = fn()
        print("{}: {} ({:.1f} s)".format(name, json.dumps(rec["workloads"][name]), time.perf_counter() - t0), flush=True)
        with open(path, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
