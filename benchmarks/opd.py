"""OPD workloads: deterministic, discrete robust, state-aware."""
import json
import os
import sys
import time

import numpy as np

from .common import *      # noqa: F401,F403  (peaks, rank helpers, parity sampling)
from .common import _episode_tables


def bench_opd(args, rank, world, local):
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    n_roots = args.roots or 1024
    budget, gamma = 5000, 0.8
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_table(t, r, term)
    roots_rng = np.random.Generator(np.random.PCG64(12345))
    non_term = np.flatnonzero(~np.asarray(term))
    all_roots = roots_rng.choice(non_term, size=world * n_roots).astype(np.int32)
    s0 = all_roots[rank * n_roots:(rank + 1) * n_roots]
    dev = torch.device("cuda", local)
    d_s0 = torch.from_numpy(s0).to(dev)
    rng0 = seed_states(np.arange(rank * n_roots, (rank + 1) * n_roots))
    d_rng = torch.from_numpy(rng0.view(np.int64)).to(dev)
    mpl = 32
    d_plans = torch.empty((n_roots, mpl), dtype=torch.int32, device=dev)
    d_len = torch.empty(n_roots, dtype=torch.int32, device=dev)
    d_lo = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_up = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_steps = torch.empty(n_roots, dtype=torch.int64, device=dev)
    d_status = torch.empty(n_roots, dtype=torch.int32, device=dev)
    sp, cross = None, None
    if world > 1:
        # N > 1 (BASELINE config C4: 8192 roots over 8 GPUs): the PRODUCT's sharded path, as the headline -- a
        # DeterministicPlannerAgent from agent_factory, ShardedDevicePlan (roots by global index, asynchronous launch,
        # mp_pack_rows -> ONE all_gather_into_tensor -> mp_unpack_rows on a side stream), cross-checked inside the run
        from rl_agents_amd import runtime
        from rl_agents_amd.agents.common.factory import agent_factory
        from rl_agents_amd.distributed import ShardedDevicePlan
        from rl_agents_amd.envs import FiniteMDPEnv
        model.close()
        ctx.close()
        ctx = runtime.get_context(local)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        env = FiniteMDPEnv(dict(mode="deterministic", transition=t, reward=r, terminal=np.asarray(term).astype(int)))
        env.reset()
        agent = agent_factory(env, {"__class__": "<class 'rl_agents_amd.agents.tree_search.deterministic.DeterministicPlannerAgent'>",
                                    "budget": budget, "gamma": gamma})
        agent.seed(0)
        sp = ShardedDevicePlan(agent, world * n_roots, max_plan_len=mpl, time_exchange=True)
        assert (sp.lo, sp.hi) == (rank * n_roots, (rank + 1) * n_roots)
        model = sp.model
        rng0 = agent.planner.batch_rng_states(n_roots, first_root=sp.lo)
        first = sp.wait(sp.plan(d_s0))
        if rank == 0:
            other, k = 1 % world, min(64, n_roots)
            lo_o = other * n_roots + (n_roots - k) // 2
            got = {key: first[key][lo_o:lo_o + k].cpu().numpy() for key in ("plans", "value", "env_steps", "status")}
            chk = ctx.opd_plan(model, all_roots[lo_o:lo_o + k], budget, gamma, 0.0, agent.planner.batch_rng_states(k, first_root=lo_o),
                               max_plan_len=mpl)
            same = (np.array_equal(got["plans"][:, 0], chk["plans"][:, 0]) and np.array_equal(got["value"], chk["root_lower"])
                    and np.array_equal(got["env_steps"], chk["env_steps"]) and not got["status"].any())
            cross = dict(cross_check="ok" if same else "MISMATCH", cross_check_roots=int(k), cross_check_of_rank=int(other))
            if not same:
                print("bench.py: gathered OPD results of rank {} differ from rank 0's re-plan".format(other), file=sys.stderr)
                os._exit(3)

    def step():
        if sp is not None:
            sp.plan(d_s0)
            return
        ctx.opd_plan_device(model, n_roots, d_s0, budget, gamma, 0.0, d_rng, mpl, plans=d_plans, plan_len=d_len,
                            root_lower=d_lo, root_upper=d_up, env_steps=d_steps, status=d_status)

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    step()
    k_ms = ctx.last_kernel_ms()[0]
    exchange_ms = None if sp is None else sp.last_exchange_ms()
    if sp is not None:
        loc = sp.local[(sp.turn - 1) % len(sp.local)]
        d_steps, d_status = loc["env_steps"], loc["status"]
    env_steps = int(d_steps.sum().item())
    assert int(d_status.abs().sum().item()) == 0
    total = sum_over_ranks(float(env_steps), world)
    k = budget // a_
    # Algorithmic bytes of THIS launch (task 2): the terms of SURVEY.md 8(d) the kernel really executes, with
    # the quantities measured on the trees the timed launch left.  Per expansion: |A| model records (13 B: T 4 + R 8 +
    # term 1) and |A| node records written (48 B).  The reference's backup_to_root after EVERY expansion (8(d)'s
    # 16 |A| d + 16 d) is NOT executed -- no decision reads an internal node's bounds, so the bounds are the bottom-up
    # fixed point computed ONCE: every expanded node reads its |A| children's (L, U) and writes its own,
    # i.e. the 8(d) backup term with d = 1.  Expansions and depth come from exported trees, not from assumptions.
    sample = np.unique(np.linspace(0, n_roots - 1, 33).astype(np.int64))
    n_exp = depth_sum = 0
    for root in sample:
        tr = ctx.opd_tree(int(root), 1 + k * a_)
        expanded = tr["first_child"] >= 0
        n_exp += int(expanded.sum())
        depth_sum += int(tr["depth"][expanded].sum())
    exp_per_root = n_exp / float(len(sample))
    d_avg = depth_sum / float(max(n_exp, 1))            # mean depth of an expanded leaf = length of the walk NOT replayed
    bytes_per_exp = a_ * (13 + 48) + 16 * a_ + 16
    alg = bytes_per_exp * exp_per_root * n_roots
    alg_survey = (a_ * (13 + 48) + 16 * a_ * d_avg + 16 * d_avg) * exp_per_root * n_roots
    res = dict(
        metric="rollout env-steps/sec (OPD plan(), budget=5000)", unit="env-steps/s",
        value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="opd_highway_shaped_S{}_A{}_budget{}_roots{}_per_gpu".format(s_, a_, budget, n_roots),
                    n_roots_per_gpu=n_roots, n_roots_total=n_roots * world, budget=budget, gamma=gamma,
                    plan_ms_per_root=1e3 * dt / args.steps / n_roots,
                    parallelism="roots sharded over {} GPU(s)".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel="opd_kernel<EXPG> (bounds in LDS) or opd_wide_kernel (bounds in HBM), chosen by the host per batch size",
                      kernel_ms=k_ms, algorithmic_bytes_per_launch=alg, bytes_per_expansion=bytes_per_exp,
                      measured_expansions_per_root=exp_per_root, measured_mean_expanded_depth=d_avg,
                      survey_formula_bytes_with_per_expansion_backup=alg_survey,
                      note="algorithmic bytes = the SURVEY 8(d) terms this kernel executes (model gathers, node records, ONE "
                           "deferred bottom-up backup); the reference's per-expansion backup walk of measured depth d is "
                           "reported separately and not charged"),
    )
    add_traffic(res["roofline"], "opd", "opd_", n_roots * 64)
    if sp is not None:
        res["exchange"] = dict(payload=sp.payload, row_bytes=int(sp.row_bytes), exchange_ms=exchange_ms, kernel_ms=k_ms,
                               step_ms=1e3 * dt / args.steps, on_side_stream=bool(sp.overlapped),
                               backend="rccl" if sp.on_device else "gloo via host")
        res["config"]["parallelism"] = ("{} roots sharded over {} GPU(s) (BASELINE C4: 8192 over 8), product path "
                                        "rl_agents_amd.distributed.ShardedDevicePlan, ONE all_gather_into_tensor of {} B rows per step"
                                        .format(world * n_roots, world, sp.row_bytes))
    if cross is not None:
        res["_cross"] = cross
    if not args.no_parity_sample and world == 1:
        from oracle import oracle
        d_rng.copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        idx = sample_rows(n_roots)
        ti = torch.from_numpy(idx).to(dev)
        ref = oracle.opd_plan_batch(t, r, term, s0[idx], budget, gamma, 0.0, rng0[idx], max_plan_len=mpl, n_threads=host_cores())
        ok = (np.array_equal(d_plans[ti].cpu().numpy(), ref["plans"]) and np.array_equal(d_len[ti].cpu().numpy(), ref["plan_len"])
              and np.array_equal(d_lo[ti].cpu().numpy(), ref["root_lower"]) and np.array_equal(d_up[ti].cpu().numpy(), ref["root_upper"])
              and np.array_equal(d_steps[ti].cpu().numpy(), ref["env_steps"]))
        res["parity_sample"] = parity_record(ok, "{} roots of a {}-root launch vs oracle.opd_plan_batch: plans, plan_len, root "
                                             "bounds, env_steps bit for bit".format(len(idx), n_roots))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the host baseline is an N = 1 figure
        from oracle import oracle
        cores = host_cores()
        n_cpu = 4 * cores
        t1 = time.perf_counter()
        o = oracle.opd_plan_batch(t, r, term, np.resize(all_roots, n_cpu), budget, gamma, 0.0, seed_states(np.arange(n_cpu)),
                                  n_threads=cores)
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=float(o["env_steps"].sum()) / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c opd_plan_batch, {} roots, OpenMP".format(n_cpu))
    return res


def bench_ropd(args, rank, world, local):
    """Discrete robust OPD (agents/robust/robust.py:28-50) at C4's shape with M = 2 models: highway-shaped S = 10 000,
    A = 5 and the same table with 10 % of the transitions rewired, budget 5000 (1000 expansions), 1024 roots per GPU."""
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    n_roots = args.roots or 1024
    budget, gamma, m_ = 5000, 0.8, 2
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    cfg2 = generators.rewire(cfg, 0.1, seed=1)
    t = np.stack([cfg["transition"], cfg2["transition"]])
    r = np.stack([cfg["reward"], cfg2["reward"]])
    term = np.stack([cfg["terminal"], cfg2["terminal"]])
    _, s_, a_ = r.shape
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_joint(t, r, term)
    non_term = np.flatnonzero(~np.asarray(cfg["terminal"]))
    all_roots = np.random.Generator(np.random.PCG64(12345)).choice(non_term, size=world * n_roots).astype(np.int32)
    s0 = np.repeat(all_roots[rank * n_roots:(rank + 1) * n_roots, None], m_, axis=1)
    dev = torch.device("cuda", local)
    d_s0 = torch.from_numpy(np.ascontiguousarray(s0)).to(dev)
    rng0 = seed_states(np.arange(rank * n_roots, (rank + 1) * n_roots))
    d_rng = torch.from_numpy(rng0.view(np.int64)).to(dev)
    mpl = 32
    d_plans = torch.empty((n_roots, mpl), dtype=torch.int32, device=dev)
    d_len = torch.empty(n_roots, dtype=torch.int32, device=dev)
    d_lo = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_up = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_steps = torch.empty(n_roots, dtype=torch.int64, device=dev)
    d_status = torch.empty(n_roots, dtype=torch.int32, device=dev)

    def step():
        ctx.ropd_plan_device(model, n_roots, d_s0, budget, gamma, 0.0, d_rng, mpl, plans=d_plans, plan_len=d_len,
                             root_lower=d_lo, root_upper=d_up, env_steps=d_steps, status=d_status)

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    step()
    k_ms = ctx.last_kernel_ms()[0]
    joint_steps = int(d_steps.sum().item())
    assert int(d_status.abs().sum().item()) == 0
    total = sum_over_ranks(float(joint_steps), world) * m_          # every joint step steps M model environments
    k = budget // a_
    # executed terms only, measured expansions / depth (see bench_opd): per expansion and child, per model one model record
    # (13 B) + {L, state, reward} (20 B), per child the minima + meta (24 B); ONE deferred bottom-up backup (16 |A| + 16).
    sample = np.unique(np.linspace(0, n_roots - 1, 33).astype(np.int64))
    n_exp = depth_sum = 0
    for root in sample:
        tr = ctx.ropd_tree(int(root), 1 + k * a_, m_)
        expanded = tr["first_child"] >= 0
        n_exp += int(expanded.sum())
        depth_sum += int(tr["depth"][expanded].sum())
    exp_per_root = n_exp / float(len(sample))
    d_avg = depth_sum / float(max(n_exp, 1))
    bytes_per_exp = a_ * m_ * (13 + 20) + a_ * 24 + 16 * a_ + 16
    alg = bytes_per_exp * exp_per_root * n_roots
    alg_survey = (a_ * m_ * (13 + 20) + a_ * 24 + 16 * a_ * d_avg + 16 * d_avg) * exp_per_root * n_roots
    res = dict(
        metric="rollout env-steps/sec (discrete robust OPD plan(), budget=5000, M=2 models)", unit="env-steps/s",
        value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="robust_opd_highway_shaped_S{}_A{}_M{}_budget{}_roots{}_per_gpu".format(s_, a_, m_, budget, n_roots),
                    n_roots_per_gpu=n_roots, n_roots_total=n_roots * world, budget=budget, gamma=gamma, models=m_,
                    plan_ms_per_root=1e3 * dt / args.steps / n_roots, parallelism="roots sharded over {} GPU(s)".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel="ropd_kernel<EXPG> or ropd_wide_kernel, chosen by the host per batch size",
                      kernel_ms=k_ms, algorithmic_bytes_per_launch=alg, bytes_per_expansion=bytes_per_exp,
                      measured_expansions_per_root=exp_per_root, measured_mean_expanded_depth=d_avg,
                      survey_formula_bytes_with_per_expansion_backup=alg_survey,
                      note="executed terms only (one deferred bottom-up backup), expansions / depth measured on exported trees"),
    )
    add_traffic(res["roofline"], "ropd", "ropd_", n_roots * 64)
    if not args.no_parity_sample and world == 1:
        from oracle import oracle
        d_rng.copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        idx = sample_rows(n_roots)
        ti = torch.from_numpy(idx).to(dev)
        ref = oracle.ropd_plan_batch(t, r, term, s0[idx], budget, gamma, 0.0, rng0[idx], max_plan_len=mpl, n_threads=host_cores())
        ok = (np.array_equal(d_plans[ti].cpu().numpy(), ref["plans"]) and np.array_equal(d_len[ti].cpu().numpy(), ref["plan_len"])
              and np.array_equal(d_lo[ti].cpu().numpy(), ref["root_lower"]) and np.array_equal(d_steps[ti].cpu().numpy(), ref["env_steps"]))
        res["parity_sample"] = parity_record(ok, "{} roots of a {}-root launch vs oracle.ropd_plan_batch: plans, plan_len, root "
                                             "lower bound, joint env_steps bit for bit".format(len(idx), n_roots))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = host_cores()
        n_cpu = 4 * cores
        t1 = time.perf_counter()
        o = oracle.ropd_plan_batch(t, r, term, np.resize(s0, (n_cpu, m_)), budget, gamma, 0.0, seed_states(np.arange(n_cpu)),
                                   n_threads=cores)
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=float(o["env_steps"].sum()) * m_ / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c orc_ropd_plan_batch, {} roots, OpenMP".format(n_cpu))
    return res


def bench_saopd(args, rank, world, local):
    """State-aware OPD (tree_search/state_aware.py) at the reference's own GridWorld configuration
    (scripts/configs/GridWorld/agents/state-aware.json: budget 500, gamma 0.8; 10x10 grid).  A step = the first plan()
    of a fresh batch of planners (the costly one: ~4 200 Bellman backups per planner on average, 1 700 .. 13 000 by root state),
    planner creation included."""
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    n_roots = args.roots or 16384
    budget, gamma = 500, 0.8
    cfg = generators.gridworld()
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_table(t, r, term)
    all_roots = np.random.Generator(np.random.PCG64(12345)).integers(0, s_, size=world * n_roots).astype(np.int32)
    s0 = all_roots[rank * n_roots:(rank + 1) * n_roots]
    rng0 = seed_states(np.arange(rank * n_roots, (rank + 1) * n_roots))
    last = {}

    def step():
        planners = native.StateAwarePlanners(ctx, model, n_roots)
        out = planners.plan(s0, budget, gamma, 0.0, rng0.copy(), max_plan_len=8)
        last.update(out=out, ms=ctx.last_kernel_ms()[0])
        planners.close()

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    out, k_ms = last["out"], last["ms"]
    assert (out["status"] == 0).all()
    env_steps = int(out["env_steps"].sum())
    total = sum_over_ranks(float(env_steps), world)
    k = budget // a_
    # algorithmic bytes of one plan: per expansion |A| model records (13 B) and node records (37 B); per iteration
    # the leaf argmax reads (lower, depth, state, state value) = 28 B of every leaf (~1/3 of the nodes are leaves),
    # per Bellman backup |A| children (28 B) + two state values; list walks of pruning / aggregation are not counted
    alg = float(n_roots) * (k * a_ * (13 + 37) + sum(28.0 * (1 + i * a_) / 3 for i in range(k))) + \
        float(out["updates"].sum()) * (28 * a_ + 16)
    res = dict(
        metric="rollout env-steps/sec (state-aware OPD plan(), budget=500)", unit="env-steps/s",
        value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="state_aware_opd_gridworld_S{}_A{}_budget{}_planners{}_per_gpu".format(s_, a_, budget, n_roots),
                    n_roots_per_gpu=n_roots, n_roots_total=n_roots * world, budget=budget, gamma=gamma,
                    plan_ms_per_root=1e3 * dt / args.steps / n_roots,
                    bellman_backups_per_planner=float(out["updates"].mean()),
                    dispatch="planners start longest first: the cost of a fresh planner's first plan by root state is learned with "
                             "the model from the warm-up batch on (saopd_order_kernel, inside the timed launch batch; "
                             "MP_SAOPD_ORDER=0 keeps the index order: +1.6 ms at 16 384 planners).  Results do not depend on it",
                    parallelism="planners sharded over {} GPU(s)".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel="saopd_wave_kernel", kernel_ms=k_ms, algorithmic_bytes_per_launch=alg),
    )
    add_traffic(res["roofline"], "saopd", "saopd_wave_kernel", n_roots * 64)
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        idx = sample_rows(n_roots)
        ref = oracle.saopd_plan_batch(t, r, term, s0[idx], budget, gamma, rng_states=rng0[idx], max_plan_len=8, n_threads=host_cores())
        ok = all(np.array_equal(out[k][idx], ref[k]) for k in ("plans", "plan_len", "env_steps", "updates", "status"))
        res["parity_sample"] = parity_record(ok, "{} planners of the timed {}-planner batch (first plan of fresh planners) vs "
                                             "oracle.saopd_plan_batch: plans, plan_len, env_steps, Bellman-backup counts, status "
                                             "bit for bit".format(len(idx), n_roots))
    if rank == 0:   # outside the timed region: what the FOLLOWING plans of the same planners cost (receding horizon)
        planners = native.StateAwarePlanners(ctx, model, n_roots)
        states, rng, follow = s0.copy(), rng0.copy(), []
        for _ in range(3):
            o = planners.plan(states, budget, gamma, 0.0, rng, max_plan_len=8)
            follow.append(round(ctx.last_kernel_ms()[0], 3))
            states = np.where(o["plan_len"] > 0, t[states, np.maximum(o["plans"][:, 0], 0)], states).astype(np.int32)
        planners.close()
        res["config"]["kernel_ms_first_and_following_plans"] = follow
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the host baseline is an N = 1 figure
        from oracle import oracle
        cores = host_cores()
        n_cpu = 64 * cores
        oracle.saopd_plan_batch(t, r, term, all_roots[:cores], budget, gamma, n_threads=cores)
        done, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds:
            o = oracle.saopd_plan_batch(t, r, term, np.resize(all_roots, n_cpu), budget, gamma,
                                        rng_states=np.resize(rng0, (n_cpu, 6)), n_threads=cores)
            done += int(o["env_steps"].sum())
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c orc_saopd_plan_batch, first plan of fresh planners, OpenMP, {} "
                                          "planners per batch for {:.1f} s".format(n_cpu, cdt))
    ctx.synchronize()
    return res
