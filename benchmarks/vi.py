"""Value-iteration workloads: deterministic / robust / dense, the row-sharded dense robust model, one MDP per agent."""
import json
import os
import sys
import time

import numpy as np

from .common import *      # noqa: F401,F403  (peaks, rank helpers, parity sampling)
from .common import _episode_tables


def bench_vi(args, rank, world, local, dense, robust=False, exact=False):
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    if dense:   # the contraction on the f64 matrix cores (tolerance parity) or in numpy's order of additions (bit-exact)
        exact = exact or args.dense_mode == "exact"
        ctx.vi_dense_mode("exact" if exact else "mfma")
    gamma, sweeps = 0.95, 200
    dev = torch.device("cuda", local)
    n_models = 1
    if robust:
        # BASELINE config C5, deterministic form: intersection-shaped table S = 50 000, A = 5, M = 2 models
        # (the second with 10 % of the transitions rewired), min over models in every backup
        cfg = generators.highway_shaped(10, 50, 100, seed=2)
        cfg2 = generators.rewire(cfg, 0.1, seed=3)
        t = np.stack([cfg["transition"], cfg2["transition"]])
        r = np.stack([cfg["reward"], cfg2["reward"] * 0.97])
        term = None
        n_models, (s_, a_) = 2, cfg["reward"].shape
        model = ctx.load_table(t, r)
        alg = 12.0 * n_models * s_ * a_ + 17.0 * s_
        flops = 0.0
        name = "vi_det_sweep (robust, M=2)"
    elif dense:
        s_, a_ = (args.states or 10000), 5
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        tt = torch.rand((s_, a_, s_), dtype=torch.float64, device=dev, generator=g)
        tt /= tt.sum(-1, keepdim=True)
        rr = torch.rand((s_, a_), dtype=torch.float64, device=dev, generator=g)
        model = ctx.load_dense(tt, rr, None)
        sweeps = 20
        alg = 8.0 * s_ * s_ * a_
        flops = 2.0 * s_ * s_ * a_
        name = "vi_dense_exact_q" if exact else "vi_dense_q"
    else:
        cfg = generators.highway_shaped(10, 10, 100, seed=0)
        t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
        s_, a_ = r.shape
        model = ctx.load_table(t, r, term)
        alg = 12.0 * s_ * a_ + 17.0 * s_
        flops = 0.0
        name = "vi_det_sweep"

    def step():
        ctx.vi_sweeps(model, gamma, sweeps, robust=robust)

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    step()
    k_ms, n_launch = ctx.last_kernel_ms()
    if not dense and n_launch == 1:
        name = "vi_det_persist (one launch, {} sweeps)".format(sweeps)
    per_sweep_ms = k_ms / sweeps
    res = dict(
        metric="value-iteration Bellman sweeps/sec", unit="sweeps/s", value=world * sweeps * args.steps / dt,
        ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="{}_S{}_A{}_{}sweeps".format("robust_vi_intersection_shaped_M2" if robust else
                                                       (("vi_dense_numpy_order" if exact else "vi_dense") if dense else "vi_highway_shaped"), s_, a_, sweeps),
                    states=s_, actions=a_, gamma=gamma, ms_per_sweep=1e3 * dt / args.steps / sweeps,
                    parallelism="replicas only ({} GPU(s))".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (per_sweep_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel=name, kernel_ms=per_sweep_ms, algorithmic_bytes_per_launch=alg),
    )
    if dense:
        add_traffic(res["roofline"], "vi_dense_exact" if exact else "vi_dense", name, None, pattern="stream")
    else:
        # committed PMC passes of this workload (profiles/*_pmc.json); the persistent kernel is ONE launch for every sweep
        traffic, raw = pmc_traffic("rvi" if robust else "vi", name.split("<")[0].split(" ")[0], None, pattern="stream")
        if traffic is not None and "persist" in name:
            traffic /= float(sweeps)
        res["roofline"].update(traffic=traffic, traffic_counters=raw, frac=res["roofline"]["achieved"] / HBM_PEAK_GBS,
                               traffic_frac=None if traffic is None else traffic / (per_sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
    if dense and not exact:
        res["roofline"]["mfma_tflops"] = flops / (per_sweep_ms * 1e-3) / 1e12
        res["roofline"]["mfma_frac_of_f64_peak"] = res["roofline"]["mfma_tflops"] / MFMA_F64_PEAK_TFLOPS
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        if dense:
            # three sweeps of the reference's iteration (value_iteration.py:65-73) on the device; a backup is independent
            # per source row, so the oracle replays a SAMPLE of rows of every sweep from the device's previous value vector
            idx = sample_rows(s_, PARITY_DENSE_ROWS)
            ti = torch.from_numpy(idx).to(dev)
            rows_t, rows_r = tt[ti].cpu().numpy(), rr[ti].cpu().numpy()
            v = torch.zeros(s_, dtype=torch.float64, device=dev)
            q = torch.empty((s_, a_), dtype=torch.float64, device=dev)
            worst, equal = 0.0, True
            for _ in range(3):
                ctx.vi_backup(model, gamma, v, q_out=q)
                ref = oracle.dense_backup_rows(rows_t, rows_r, None, v.cpu().numpy(), gamma)
                got = q[ti].cpu().numpy()
                worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))))
                equal = equal and bool(np.array_equal(got, ref))
                v = q.max(dim=-1).values
            if exact:
                res["parity_sample"] = parity_record(equal, "3 sweeps, {} sampled source rows per sweep vs oracle.dense_backup_rows "
                                                     "(numpy's add.reduce order): bit for bit".format(len(idx)), max_rel_err=worst)
            else:
                res["parity_sample"] = parity_record(worst <= 1e-12, "3 sweeps, {} sampled source rows per sweep vs "
                                                     "oracle.dense_backup_rows (numpy's pairwise order); tolerance 1e-12 relative "
                                                     "(matrix-core accumulation order)".format(len(idx)), max_rel_err=worst)
        else:
            q, sw = ctx.vi_solve(model, gamma, 3, robust=robust)
            q_ref, sw_ref = oracle.vi_solve("deterministic", t, r, term, gamma=gamma, iterations=3, robust=robust)
            res["parity_sample"] = parity_record(bool(sw == sw_ref and np.array_equal(q, q_ref)),
                                                 "3 sweeps vs oracle.vi_solve: Q [{} x {}] and the sweep count bit for bit".format(s_, a_))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and dense:
        # bounded sample: the oracle's dense sweep (numpy's pairwise add.reduce restated, one thread) costs
        # O(S^2 |A|); time it at S = 2000 (160 MB of transitions) and scale by (2000 / S)^2
        from oracle import oracle
        s_cpu = min(2000, s_)
        g_cpu = np.random.Generator(np.random.PCG64(0))
        t_cpu = g_cpu.random((s_cpu, a_, s_cpu))
        t_cpu /= t_cpu.sum(-1, keepdims=True)
        r_cpu = g_cpu.random((s_cpu, a_))
        t1, reps, n_sw = time.perf_counter(), 0, 5
        while time.perf_counter() - t1 < args.cpu_seconds:
            oracle.vi_solve("stochastic", t_cpu, r_cpu, None, gamma=gamma, iterations=n_sw, rtol=-1.0, atol=-1.0)
            reps += 1
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=reps * n_sw / cdt * (s_cpu / s_) ** 2, unit="sweeps/s", cores=1, kind="port",
                                   sample="oracle/planning_oracle.c orc_vi_solve (dense), {} x {} sweeps at S = {} in {:.1f} s, "
                                          "scaled by (S_sample / S)^2 to S = {}".format(reps, n_sw, s_cpu, cdt, s_))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dense:
        from oracle import oracle
        t1 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t1 < args.cpu_seconds:
            oracle.vi_solve("deterministic", t, r, term, gamma=gamma, iterations=sweeps, rtol=-1.0, atol=-1.0,
                            robust=robust)
            reps += 1
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=reps * sweeps / cdt, unit="sweeps/s", cores=1, kind="port",
                                   sample="oracle/planning_oracle.c orc_vi_solve, {} x {} sweeps".format(reps, sweeps))
    return res


def bench_rvi_dense_shard(args, rank, world, local):
    """BASELINE config C5 in its dense form -- robust VI, S = 50 000, A = 5, M = 2 models, 8*M*S^2*A = 200 GB of fp64
    transitions, row-sharded over the 8 GPUs of a node -- timed at the size it exists for: every rank owns
    6 250 source-state rows of both models (25 GB, generated on the device and borrowed by the library), a step is ONE
    sweep of the sharded solver's loop: mp_vi_backup on the rank's rows (min over models fused), max_a, the allclose
    test, and the exchange of V (all_gather_into_tensor over RCCL; at N = 1 a single-rank process group stands in for it,
    which measures the collective's software path but no wire time).  N < 8 ranks cover N * 6250 of the 50 000 source
    rows (weak scaling: the per-GPU work is the C5 rank's; the missing rows' values stay 0 -- a timing harness, the
    solver's results are covered by the tests at small sizes)."""
    import torch
    import torch.distributed as dist
    from rl_agents_amd import native
    s_, a_, m_ = (args.states or 50000), 5, 2
    rows = args.roots or s_ // 8
    gamma = 0.95
    dev = torch.device("cuda", local)
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    exact = (args.dense_mode or DENSE_SHARD_MODE) == "exact"
    ctx.vi_dense_mode("exact" if exact else "mfma")
    kname = "vi_dense_exact_q (robust, row block)" if exact else "vi_dense_q (robust, row block)"
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    tt = torch.empty((m_, rows, a_, s_), dtype=torch.float64, device=dev)
    for m in range(m_):                                   # row-stochastic blocks, normalised in place model by model
        tt[m].uniform_(generator=g)
        tt[m] /= tt[m].sum(-1, keepdim=True)
    rr = torch.rand((m_, rows, a_), dtype=torch.float64, device=dev, generator=g)
    model = ctx.load_dense_rows(tt, rr, None)
    lo = rank * rows
    group = world > 1
    standin = None
    if world == 1 and os.environ.get("BENCH_RCCL_STANDIN"):
        # (opt-in: RCCL prints its version banner on stdout, which would follow the JSON line)
        try:                                              # single-rank RCCL group: the collective's launch path, no wire
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
            group, standin = True, "single-rank RCCL process group (software path only)"
        except Exception as e:                            # pragma: no cover - depends on the box
            standin = "unavailable ({})".format(type(e).__name__)
    n_cover = max(world, 1) * rows
    v = torch.zeros(s_, dtype=torch.float64, device=dev)
    v_all = torch.zeros(n_cover, dtype=torch.float64, device=dev)
    q_local = torch.zeros((rows, a_), dtype=torch.float64, device=dev)
    q_next = torch.empty_like(q_local)
    done = torch.zeros(1, dtype=torch.int32, device=dev)
    t_gather = []

    def step(timed_gather=False):
        nonlocal q_local, q_next, done
        ctx.vi_backup(model, gamma, v, q_out=q_next, robust=True)
        close = torch.isclose(q_local, q_next, rtol=0.0, atol=0.0).all().to(torch.int32).reshape(1)   # (exact equality: never close here)
        if group:
            dist.all_reduce(close, op=dist.ReduceOp.MIN)
        done = torch.maximum(done, close)
        q_local, q_next = q_next, q_local
        v_loc = q_local.max(dim=-1).values
        if group:
            if timed_gather:
                torch.cuda.synchronize()
                t1 = time.perf_counter()
            dist.all_gather_into_tensor(v_all, v_loc)
            if timed_gather:
                torch.cuda.synchronize()
                t_gather.append(time.perf_counter() - t1)
            v[:n_cover] = v_all
        else:
            v[lo:lo + rows] = v_loc

    for _ in range(args.warmup):
        step()
    barrier(world if world > 1 else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world if world > 1 else 1)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    k_ms = []
    for _ in range(5):
        step(timed_gather=True)
        k_ms.append(ctx.last_kernel_ms()[0])
    k_ms = float(np.mean(k_ms))
    alg = 8.0 * m_ * rows * a_ * s_
    flops = 2.0 * m_ * rows * a_ * s_
    gather_ms = 1e3 * float(np.median(t_gather)) if t_gather else None
    ms_sweep = 1e3 * dt / args.steps
    res = dict(
        metric="value-iteration Bellman sweeps/sec (dense robust VI, one C5 rank's row block per GPU)", unit="sweeps/s",
        value=args.steps / dt, ms_per_step=ms_sweep, dtype="f64",
        config=dict(workload="robust_vi_dense_row_shard_S{}_A{}_M{}_rows{}_per_gpu".format(s_, a_, m_, rows), states=s_,
                    actions=a_, models=m_, rows_per_gpu=rows, block_bytes=alg, gamma=gamma, ms_per_sweep=ms_sweep,
                    kernel_ms_per_sweep=k_ms, all_gather_ms=gather_ms, all_gather_standin=standin,
                    projection_8_ranks=dict(
                        note="C5 = 8 such ranks: a sweep costs max over ranks of (backup + torch epilogue) + the V exchange; "
                             "the exchange moves 8*S = {} B and is latency-bound on the xGMI mesh".format(8 * s_),
                        ms_per_sweep=ms_sweep, sweeps_per_s=args.steps / dt,
                        full_model_bytes_per_sweep=8.0 * alg, aggregate_tb_per_s=8.0 * alg / (ms_sweep * 1e-3) / 1e12),
                    parallelism="rows sharded over {} GPU(s) ({} of 8 C5 ranks), all_gather of V + 4-byte all_reduce per sweep".format(world, world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", kernel=kname,
                      kernel_ms=k_ms, algorithmic_bytes_per_launch=alg),
    )
    if not exact:
        res["roofline"]["mfma_tflops"] = flops / (k_ms * 1e-3) / 1e12
        res["roofline"]["mfma_frac_of_f64_peak"] = res["roofline"]["mfma_tflops"] / MFMA_F64_PEAK_TFLOPS
    res["config"]["dense_mode"] = "exact" if exact else "mfma"
    add_traffic(res["roofline"], "rvi_dense_shard_exact" if exact else "rvi_dense_shard", "vi_dense_exact_q" if exact else "vi_dense_q", None,
                pattern="stream")
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        idx = sample_rows(rows, PARITY_DENSE_ROWS)
        ti = torch.from_numpy(idx).to(dev)
        rows_t, rows_r = tt[:, ti].cpu().numpy(), rr[:, ti].cpu().numpy()
        vv = torch.zeros(s_, dtype=torch.float64, device=dev)
        qq = torch.empty((rows, a_), dtype=torch.float64, device=dev)
        worst, equal = 0.0, True
        for _ in range(3):
            ctx.vi_backup(model, gamma, vv, q_out=qq, robust=True)
            ref = oracle.dense_backup_rows(rows_t, rows_r, None, vv.cpu().numpy(), gamma, robust=True)
            got = qq[ti].cpu().numpy()
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))))
            equal = equal and bool(np.array_equal(got, ref))
            vv[lo:lo + rows] = qq.max(dim=-1).values
        res["parity_sample"] = parity_record(equal if exact else worst <= 1e-12,
                                             "3 sweeps, {} sampled rows of this rank's block per sweep vs oracle.dense_backup_rows "
                                             "(robust, M = 2); {}".format(len(idx), "bit for bit" if exact else "tolerance 1e-12 relative"),
                                             max_rel_err=worst)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        s_cpu = 1000
        g_cpu = np.random.Generator(np.random.PCG64(0))
        t_cpu = g_cpu.random((m_, s_cpu, a_, s_cpu))
        t_cpu /= t_cpu.sum(-1, keepdims=True)
        r_cpu = g_cpu.random((m_, s_cpu, a_))
        t1, reps, n_sw = time.perf_counter(), 0, 5
        while time.perf_counter() - t1 < args.cpu_seconds:
            oracle.vi_solve("stochastic", t_cpu, r_cpu, None, gamma=gamma, iterations=n_sw, rtol=-1.0, atol=-1.0, robust=True)
            reps += 1
        cdt = time.perf_counter() - t1
        scale = (float(m_) * s_cpu * a_ * s_cpu) / (float(m_) * rows * a_ * s_)
        res["cpu_baseline"] = dict(value=reps * n_sw / cdt * scale, unit="sweeps/s", cores=1, kind="port",
                                   sample="oracle/planning_oracle.c orc_vi_solve (dense, robust M=2), {} x {} sweeps at S = {} in "
                                          "{:.1f} s, scaled by bytes to this rank's block".format(reps, n_sw, s_cpu, cdt))
    if world == 1 and group and dist.is_initialized():
        dist.destroy_process_group()
    return res


def bench_vi_batch(args, rank, world, local):
    """N value-iteration agents in ONE launch (round 5): a batch of episodes each owns its finite MDP (highway-v0's
    to_finite_mdp() table, re-extracted at every step: value_iteration.py:29-35) -- mp_vi_solve_batch solves all of them, each to
    its own allclose exit, as N ValueIterationAgent objects would (gamma 0.95, at most 200 sweeps).  --roots = MDPs per GPU
    (default 4096), --states 120 (grid 3 x 4 x 10, highway-env's default shape) or 10 000 (10 x 10 x 100, BASELINE C2's shape).
    A step = the solve of all MDPs of this rank (their tables resident on the device).  Independent MDPs shard over ranks with no
    collective."""
    import torch
    from rl_agents_amd import native
    s_req = args.states or 120
    shape = (3, 4, 10) if s_req <= 120 else (10, 10, 100)
    n = args.roots or (4096 if s_req <= 120 else 64)
    gamma, iters = 0.95, 200
    tr, rw, tm = _episode_tables(n, shape, seed0=1000 * rank, distinct=n if s_req <= 120 else 64)
    s_, a_ = tr.shape[1:]
    dev = torch.device("cuda", local)
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_table_batch(tr, rw, tm)
    d_q = torch.zeros((n * s_, a_), dtype=torch.float64, device=dev)
    d_sw = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def step():
        ctx.vi_solve_batch_device(model, gamma, iters, d_q, d_sw)

    for _ in range(args.warmup):
        step()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    k_ms = []
    for _ in range(5):
        step()
        k_ms.append(ctx.last_kernel_ms()[0])
    k_ms = float(np.mean(k_ms))
    variant = ctx.last_kernel_variant()
    sweeps = d_sw.cpu().numpy().astype(np.int64)
    total_sweeps = sum_over_ranks(float(sweeps.sum()), world)
    # the single-solve path on ONE of these MDPs (what an agent that owns one environment calls): sweeps per second
    single = ctx.load_table(tr[0], rw[0], tm[0])
    q1 = torch.zeros((s_, a_), dtype=torch.float64, device=dev)
    sw1 = torch.zeros(1, dtype=torch.int32, device=dev)
    ctx.vi_solve_device(single, gamma, iters, q1, sw1)
    one_ms = []
    for _ in range(5):
        ctx.vi_solve_device(single, gamma, iters, q1, sw1)
        one_ms.append(ctx.last_kernel_ms()[0])
    one_sweeps = int(sw1.cpu().numpy()[0])
    single_rate = one_sweeps / (float(np.mean(one_ms)) * 1e-3)
    single.close()
    per_sweep = 12.0 * s_ * a_ + 17.0 * s_                      # T 4 + R 8 per (s, a); V read + write + flag per state
    alg_survey = per_sweep * float(sweeps.sum())
    # what a launch must move BEYOND THE CU (the roofline's numerator): the register form reads an MDP's tables once per SOLVE
    # and writes its Q; the streaming form re-reads 10 B per (s, a) and writes 8 B per state every sweep, after one pass that
    # re-lays the tables out lane-major (12 S A + S read, 10 S A written)
    sa = float(s_ * a_)
    kc = int(variant[len("vi_batch_cluster"):]) if variant.startswith("vi_batch_cluster") else 0
    if kc:
        # the cluster form: tables once per solve (rows in registers), Q out, and per sweep each of the K workgroups of an MDP
        # publishes its slice of V (8 S in all) and reads the other K - 1 slices (8 S (K - 1) / K each): 8 S K bytes past the CU
        alg = float(n) * (12.0 * sa + s_ + 8.0 * sa + 4.0) + float(sweeps.sum()) * 8.0 * s_ * kc
    elif "reg" in variant:
        alg = float(n) * (12.0 * sa + s_ + 8.0 * sa + 4.0)
    elif "stream" in variant:
        alg = float(sweeps.sum()) * (10.0 * sa + 8.0 * s_) + float(n) * ((12.0 * sa + s_) + 10.0 * sa + 8.0 * sa + 4.0)
    else:
        alg = alg_survey + float(n) * 8.0 * sa
    rate = total_sweeps * args.steps / dt
    res = dict(
        metric="value-iteration Bellman sweeps/sec (N independent MDPs per launch, each to its own allclose exit)", unit="sweeps/s",
        value=rate, ms_per_step=1e3 * dt / args.steps, dtype="f64", variant=variant,
        speedup_vs_single_solve=dict(batch_sweeps_per_s=float(sweeps.sum()) / (k_ms * 1e-3), single_solve_sweeps_per_s=single_rate,
                                     ratio=float(sweeps.sum()) / (k_ms * 1e-3) / single_rate, single_solve_kernel_ms=float(np.mean(one_ms)),
                                     note="kernel time of ONE mp_vi_solve_batch launch over all MDPs against mp_vi_solve on one of them"),
        config=dict(workload="vi_batch_{}_mdps_highway_shaped_S{}_A{}_gamma{}_max{}sweeps".format(n, s_, a_, gamma, iters),
                    mdps_per_gpu=n, states=s_, actions=a_, gamma=gamma, iterations=iters, sweeps_run_mean=float(sweeps.mean()),
                    sweeps_run_min=int(sweeps.min()), sweeps_run_max=int(sweeps.max()), solves_per_s=n * args.steps / dt * world,
                    parallelism="independent MDPs sharded over {} GPU(s), no collective".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", kernel=variant, kernel_ms=k_ms,
                      algorithmic_bytes_per_launch=alg, survey_formula_bytes_per_launch=alg_survey,
                      survey_formula_rate_vs_hbm_peak=alg_survey / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      note="bytes = what the launch moves beyond the CU: the register form (S <= 4096) keeps an MDP's rows in "
                           "registers and V in LDS and touches memory once per SOLVE (tables in, Q out); the streaming form "
                           "(S = 10 000) re-reads 10 B per (s, a) and writes 8 B per state per sweep (L2 / infinity cache "
                           "resident) after one lane-major re-layout pass; the cluster form (few large MDPs: K workgroups each) is the register "
                           "form + 8 S K bytes of V exchange per sweep.  SURVEY 8(d)'s per-sweep formula (12 S A + 17 S) x the "
                           "sweeps really run is beside it (`survey_formula_*`): for the register form that rate exceeds the HBM peak "
                           "because those bytes never leave the CU -- it is not HBM traffic"),
    )
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
    block = 1024 if ("wg" in variant or kc) else int(variant.split(",")[-1].rstrip(">"))
    kernel_name = "vi_det_batch_cluster" if kc else "vi_det_batch_reg" if "reg" in variant else ("vi_det_batch_wgr" if "stream" in variant else "vi_det_batch_wg<")
    add_traffic(res["roofline"], "vi_batch", kernel_name, ((n + 7) // 8 * 8 * kc if kc else n) * block, pattern="stream")
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        idx = sample_rows(n, 512 if s_ <= 120 else 8)
        q_ref, sw_ref = oracle.vi_solve_each(tr[idx], rw[idx], tm[idx], gamma=gamma, iterations=iters)
        q = d_q.cpu().numpy().reshape(n, s_, a_)
        ok = np.array_equal(q[idx], q_ref) and np.array_equal(sweeps[idx], sw_ref)
        res["parity_sample"] = parity_record(ok, "{} MDPs of the timed {}-MDP launch vs {} sequential oracle solves: Q and sweep "
                                             "counts bit for bit".format(len(idx), n, len(idx)))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        t1, done, i = time.perf_counter(), 0, 0
        while time.perf_counter() - t1 < args.cpu_seconds:
            _, k = oracle.vi_solve("deterministic", tr[i % n], rw[i % n], tm[i % n], gamma=gamma, iterations=iters)
            done += k
            i += 1
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="sweeps/s", cores=1, kind="port",
                                   sample="oracle/planning_oracle.c orc_vi_solve on {} of these MDPs one after the other ({:.1f} s)".format(i, cdt))
    model.close()
    return res
