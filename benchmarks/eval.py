"""End-to-end slice of the PER-EPISODE evaluation loop (trainer/per_episode_evaluation.py: SURVEY 8 f-1 + f-2 together): N
environments that each own a finite MDP which changes at every step -- highway-v0's surface -- advanced in lock-step with one
batched plan per step.  The reference runs this loop one (environment, agent) pair per process (trainer/evaluation.py:139-194)."""
import time

import numpy as np

from .common import *      # noqa: F401,F403


def bench_per_episode_eval(args, rank, world, local):
    """N = 4096 ChangingHighwayEnv episodes (a (3, 4, 10) time-to-collision grid re-drawn after every step, restricted action
    sets listed IDLE first), MCTSAgent budget 1000 (33 x 30); a step of the slice = every live episode advances by one
    environment step.  `value` = episode-steps per second of the whole loop; the split says where a step's time goes: the
    environments' own to_finite_mdp() + table extraction (host), comparing / uploading the changed tables, the batched plan
    (launches + results), env.step (host).  Beside it: the same loop with value iteration and with MCTSWithPriorPolicyAgent
    (value iteration re-solved per episode per step as the prior)."""
    import torch
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.agents.tree_search.mcts_with_prior import MCTSWithPriorPolicyAgent
    from rl_agents_amd.envs import ChangingHighwayEnv
    from rl_agents_amd.trainer.per_episode_evaluation import PerEpisodeEvaluation
    n = args.roots or 4096
    steps = max(int(args.steps), 1)
    vi_cls = "<class 'rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent'>"

    def envs():
        return [ChangingHighwayEnv(3, 4, 10, table_seed=100000 * rank + 20 * i, state=((i % 3) * 4 + (i % 4)) * 10,
                                   collision_rate=0.03 + 0.02 * (i % 4)) for i in range(n)]
    kinds = [("mcts", MCTSAgent, dict(budget=1000, gamma=0.8, horizon=30, episodes=33)),
             ("vi", ValueIterationAgent, dict(gamma=0.95, iterations=200)),
             ("mcts_vi_prior", MCTSWithPriorPolicyAgent,
              dict(budget=1000, gamma=0.8, horizon=30, episodes=33,
                   prior_agent={"__class__": vi_cls, "gamma": 0.95, "iterations": 200, "temperature": 0.3}))]
    rows = {}
    for name, cls, cfg in kinds:
        es = envs()
        ev = PerEpisodeEvaluation(es, cls(es[0], dict(cfg)), sim_seed=1000 * rank, max_steps=steps)
        t0 = time.perf_counter()
        out = ev.run()
        wall = time.perf_counter() - t0
        torch.cuda.synchronize()
        done = int(out["lengths"].sum())
        sec = out["seconds"]
        rows[name] = dict(episode_steps=done, wall_s=wall, episode_steps_per_s=done / wall,
                          ms_per_lockstep=1e3 * wall / max(int(out["lengths"].max()), 1),
                          split_ms_per_lockstep={k: 1e3 * v / max(int(out["lengths"].max()), 1) for k, v in sec.items()},
                          planner_env_steps=int(out["planner_env_steps"]), uploads=int(out["uploads"]),
                          kernel_variant=ev.ctx.last_kernel_variant() if name != "vi" else "vi_det_batch",
                          mean_return=float(out["returns"].mean()))
        ev.close()
        del ev, es
    head = rows["mcts"]
    total = sum_over_ranks(float(head["episode_steps"]), world)
    dt = max_over_ranks(head["wall_s"], world)
    # the per-step plan of the headline kind: what the device does of a lock-step
    plan_ms = head["split_ms_per_lockstep"]["plan"]
    res = dict(
        metric="episode-steps/sec (per-episode evaluation loop, one finite MDP per episode re-extracted every step)", unit="episode-steps/s",
        value=total / dt, ms_per_step=1e3 * dt / steps, dtype="f64",
        config=dict(workload="per_episode_eval_changing_highway_S120_A5_mcts_budget1000_e33xh30_episodes{}_per_gpu".format(n),
                    episodes=33, horizon=30, actions=5, n_roots_per_gpu=n, lock_steps=steps, kinds=rows,
                    parallelism="episodes sharded over {} GPU(s), no collective".format(world)),
        # the loop is bound by the HOST (one Python environment object per episode): the roofline of its device part is the plan
        roofline=dict(bound="hbm", kernel="the batched plan of one lock-step ({})".format(head["kernel_variant"]), kernel_ms=plan_ms,
                      achieved=0.0, peak=HBM_PEAK_GBS, unit="GB/s", frac=0.0, traffic=None, traffic_frac=None,
                      note="host-bound loop: extraction + env.step of {} Python environment objects per lock-step; `kernel_ms` is "
                           "the plan's share (upload of changed tables excluded), see config.kinds[*].split_ms_per_lockstep".format(n)),
    )
    alg = 35.0 * head["planner_env_steps"] / max(steps, 1)        # SURVEY 8(d)'s headline figure per planner env step
    res["roofline"]["algorithmic_bytes_per_launch"] = alg
    res["roofline"]["achieved"] = alg / (plan_ms * 1e-3) / 1e9 if plan_ms > 0 else 0.0
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
    if not args.no_parity_sample and rank == 0:
        # 16 of the episodes again as sequential (environment, agent) loops of the single agents -- what the reference runs
        k = min(16, n)
        es = envs()[:k]
        ev = PerEpisodeEvaluation(es, MCTSAgent(es[0], dict(kinds[0][2])), sim_seed=1000 * rank, max_steps=steps)
        got = ev.run()["actions"]
        ev.close()
        ok = True
        for i, env in enumerate(envs()[:k]):
            obs, _ = env.reset()
            agent = MCTSAgent(env, dict(kinds[0][2]))
            agent.seed(1000 * rank + i)
            for t in range(steps):
                a = int(agent.act(obs))
                ok = ok and a == int(got[i, t])
                obs, _, term, trunc, _ = env.step(a)
                if term or trunc:
                    break
        res["parity_sample"] = parity_record(ok, "{} episodes x {} steps: the batch's actions == sequential single-agent loops "
                                             "(each bit-exact vs the reference on tests/golden/per_episode*.npz)".format(k, steps))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the same loop on the host's cores: the C port planning one episode's step at a time (the reference's structure)
        from oracle import oracle
        from rl_agents_amd.envs import generators
        t1, done = time.perf_counter(), 0
        p = np.ones(5) / 5
        rng = seed_states(np.arange(64))
        while time.perf_counter() - t1 < args.cpu_seconds:
            cfg = generators.highway_shaped(3, 4, 10, seed=done)
            s0 = np.resize(np.arange(0, 120, 10), 64)
            oracle.uct_plan_batch(cfg["transition"], cfg["reward"], cfg["terminal"], s0, 33, 30, 0.8, 10.0, p, p, rng, n_threads=host_cores())
            done += 64
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="episode-steps/s", cores=host_cores(), kind="port",
                                   sample="oracle uct_plan_batch 33x30 on one (3,4,10) table per 64 episode-steps, {:.1f} s; plan only "
                                          "(no extraction / env.step)".format(cdt))
    return res
