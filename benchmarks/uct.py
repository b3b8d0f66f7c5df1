"""UCT workloads: shared finite MDP (headline), CartPole, stochastic model, one MDP per root."""
import json
import os
import sys
import time

import numpy as np

from .common import *      # noqa: F401,F403  (peaks, rank helpers, parity sampling)
from .common import _episode_tables


def bench_uct(args, rank, world, local, with_prior=False):
    """Headline.  The N-GPU form times the PRODUCT's sharded path: an MCTSAgent built by agent_factory on a FiniteMDPEnv
    of the table, rl_agents_amd.distributed.ShardedDevicePlan (roots sharded by global index, the planner's asynchronous
    batched launch, mp_pack_rows -> one all_gather_into_tensor -> mp_unpack_rows on a side stream).
    with_prior: MCTSWithPriorPolicyAgent's path -- value iteration on the device, its Boltzmann
    distribution as per-state prior and rollout policy (tables built and uploaded outside the timed region)."""
    import torch
    from rl_agents_amd import native, runtime
    from rl_agents_amd.envs import generators
    n_roots = args.roots or 262144
    episodes, horizon, gamma = 33, 30, 0.8
    temperature = 2 / (1 - 0.8)                       # mcts.py:121-124 default
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    gids = np.arange(rank * n_roots, (rank + 1) * n_roots)
    roots_rng = np.random.Generator(np.random.PCG64(12345))
    non_term = np.flatnonzero(~np.asarray(term))
    all_roots = roots_rng.choice(non_term, size=world * n_roots).astype(np.int32)
    s0 = all_roots[gids]
    dev = torch.device("cuda", local)
    d_s0 = torch.from_numpy(s0).to(dev)
    mpl = 8
    p = np.ones(a_) / a_
    policy, tables, sp, cross = None, None, None, None
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    if with_prior:
        ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
        model = ctx.load_table(t, r, term)
        rng0 = seed_states(gids)
        d_rng = torch.from_numpy(rng0.view(np.int64)).to(dev)   # raw 64-bit words
        d_plans = torch.empty((n_roots, mpl), dtype=torch.int32, device=dev)
        d_len = torch.empty(n_roots, dtype=torch.int32, device=dev)
        d_val = torch.empty(n_roots, dtype=torch.float64, device=dev)
        d_steps = torch.empty(n_roots, dtype=torch.int64, device=dev)
        q, _ = ctx.vi_solve(model, 0.95, 200)
        z = np.exp((q - q.max(axis=1, keepdims=True)) / 0.3)
        tables = z / z.sum(axis=1, keepdims=True)
        policy = ctx.load_policy(model, tables, tables)
        p = tables                                           # the oracle takes the [S, A] tables in p's place

        def step():
            ctx.uct_plan_device(model, n_roots, d_s0, episodes, horizon, gamma, temperature, p, p, d_rng, mpl,
                                plans=d_plans, plan_len=d_len, root_value=d_val, env_steps=d_steps, policy=policy)
            d_total.add_(d_steps.sum())
    else:
        from rl_agents_amd.agents.common.factory import agent_factory
        from rl_agents_amd.distributed import ShardedDevicePlan
        from rl_agents_amd.envs import FiniteMDPEnv
        # the package's process-wide context enqueues on this bench's stream (torch ops, RCCL and kernels: one order)
        ctx = runtime.get_context(local)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        env = FiniteMDPEnv(dict(mode="deterministic", transition=t, reward=r, terminal=np.asarray(term).astype(int)))
        env.reset()
        agent = agent_factory(env, {"__class__": "<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>",
                                    "budget": 1000, "gamma": gamma, "horizon": horizon, "episodes": episodes})
        agent.seed(0)
        assert agent.planner.config["temperature"] == temperature
        sp = ShardedDevicePlan(agent, world * n_roots, max_plan_len=mpl, time_exchange=world > 1)
        assert (sp.lo, sp.hi) == (rank * n_roots, (rank + 1) * n_roots)
        model = sp.model
        rng0 = agent.planner.batch_rng_states(n_roots, first_root=sp.lo)
        d_rng = sp.d_rng
        # ---- cross-check inside the run: what the gather delivered for ANOTHER rank's roots == rank 0's own re-plan
        first = sp.wait(sp.plan(d_s0))
        if rank == 0:
            other = 1 % world
            k = min(256, n_roots)
            lo_o = other * n_roots + (n_roots - k) // 2            # a block from the middle of that rank's shard
            take = np.arange(lo_o, lo_o + k)
            got = {key: first[key][lo_o:lo_o + k].cpu().numpy() for key in ("plans", "plan_len", "value", "env_steps", "status")}
            chk = ctx.uct_plan(model, all_roots[take], episodes, horizon, gamma, temperature, p, p,
                               agent.planner.batch_rng_states(k, first_root=lo_o), max_plan_len=mpl)
            w = got["plans"].shape[1]                  # plan entries a row carries (1: the compact payload)
            same = (np.array_equal(got["plans"], chk["plans"][:, :w]) and np.array_equal(got["plan_len"], np.minimum(chk["plan_len"], w))
                    and np.array_equal(got["value"], chk["root_value"]) and np.array_equal(got["env_steps"], chk["env_steps"])
                    and not got["status"].any())
            cross = dict(cross_check="ok" if same else "MISMATCH", cross_check_roots=int(k), cross_check_of_rank=int(other),
                         cross_check_what="rank 0 re-planned global roots [{}, {}) through the host-array API and compared "
                                          "first action / root value / env_steps / status with the gathered rows".format(lo_o, lo_o + k))
            if not same:
                print("bench.py: gathered results of rank {} differ from rank 0's re-plan".format(other), file=sys.stderr)
                os._exit(3)
        loc = sp.local[0]
        d_plans, d_len, d_val, d_steps = loc["plans"], loc["plan_len"], loc["value"], loc["env_steps"]

        def step():
            sp.plan(d_s0)
            d_total.add_(sp.local[(sp.turn - 1) % len(sp.local)]["env_steps"].sum())

    for _ in range(args.warmup):
        step()
    barrier(world)
    env_steps = 0
    kernel_ms = []
    d_total.zero_()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = time.perf_counter() - t0
    timed_env_steps = int(d_total.item())
    # per-launch kernel time from HIP events on the kernel's stream (separate short pass so that the
    # event synchronisation does not sit inside the timed region above)
    def last_buffers():
        """The result buffers of the last step (the sharded path alternates between two sets)."""
        if sp is not None:
            return sp.local[(sp.turn - 1) % len(sp.local)]
        return dict(plans=d_plans, plan_len=d_len, value=d_val, env_steps=d_steps)

    exchange_ms = []
    for _ in range(min(args.steps, 10)):
        step()
        kernel_ms.append(ctx.last_kernel_ms()[0])
        if sp is not None and sp.last_exchange_ms() is not None:
            exchange_ms.append(sp.last_exchange_ms())
        d_steps = last_buffers()["env_steps"]
        env_steps = int(d_steps.sum().item())
    variant = ctx.last_kernel_variant()
    # the GENERAL-model kernel on the same batch: a model that does not fit LDS, has more than 256 distinct rewards
    # or 32 768+ states gathers 16-byte records from L2 / HBM instead (`uct_global`); printed beside the LDS-resident headline
    general = None
    if variant == "uct_ldsr" and not with_prior and not os.environ.get("MP_UCT_MODEL"):
        os.environ["MP_UCT_MODEL"] = "global"
        try:
            gk = []
            step()
            for _ in range(3):
                step()
                gk.append(ctx.last_kernel_ms()[0])
            g_steps = int(last_buffers()["env_steps"].sum().item())
            general = dict(kernel_variant=ctx.last_kernel_variant(), kernel_ms=float(np.mean(gk)),
                           value=sum_over_ranks(g_steps / (float(np.mean(gk)) * 1e-3), world), unit="env-steps/s",
                           note="same roots, same plans (bit-identical results), the record-gather kernel: what a model that "
                                "cannot live in LDS gets; rate = env steps / kernel time")
        finally:
            os.environ.pop("MP_UCT_MODEL", None)
    dt = max_over_ranks(dt, world)
    total_env_steps = sum_over_ranks(float(timed_env_steps), world) / args.steps   # per step, all ranks
    # Algorithmic bytes of THIS run, SURVEY.md 8(d): per env step 13 B of model (T 4 + R 8 + term 1); per selection
    # level |A| children x 16 B; per episode a backup read-modify-write of 24 B on each of its depth + 1 path nodes; per
    # expansion |A| node records of 24 B.  Depth and expansion counts are MEASURED on the trees this launch left
    # (sum of the visit counts of the non-root nodes = selection steps; nodes with children = expansions), not assumed.
    sample = np.unique(np.linspace(0, n_roots - 1, 257).astype(np.int64))
    sel_steps = expansions = sample_env = 0
    smp_steps = d_steps[torch.from_numpy(sample).to(dev)].cpu().numpy()
    for i, root in enumerate(sample):
        tr = ctx.uct_tree(int(root))
        sel_steps += int(tr["count"][1:].sum())
        expansions += int((tr["first_child"] >= 0).sum())
        sample_env += int(smp_steps[i])
    n_smp = len(sample)
    mean_depth = sel_steps / float(n_smp * episodes)
    # model term: 13 B per env step gathered from the 16-byte records -- or, when the kernel keeps the whole model in LDS
    # (uct_ldsr: the default from 65 536 roots), only what every workgroup stages once per launch: 3 B per (s, a) + tables
    staged, bytes_per_step_hbm = None, None
    tree_bytes = 16.0 * a_ * sel_steps + 24.0 * (sel_steps + n_smp * episodes) + 24.0 * a_ * expansions
    bytes_per_step = (13.0 * sample_env + tree_bytes) / sample_env          # the ALGORITHM's bytes, whatever serves them
    if variant == "uct_ldsr":
        # ... of which the 13 B per env step of the model are served from LDS by this kernel: what it must move through HBM is
        # the tree terms + what every workgroup stages once per launch (3 B per (s, a) + tables) -- reported beside `frac`
        cus = ctx.device_info()["n_cu"]
        waves = 1
        while waves < -(-(n_roots // 64) // cus) and waves < 16:
            waves *= 2
        n_wg = -(-n_roots // (64 * waves))
        staged = n_wg * (3.0 * s_ * a_ + 8.0 * len(np.unique(r)) + 8.0 * (horizon + 1 + 2 * a_ + (episodes + 1) + a_ * (episodes + 2)))
        bytes_per_step_hbm = (staged / float(env_steps) * sample_env + tree_bytes) / sample_env
    # metric half (ii) and the 8(d) definition: small batches and the host-inclusive call, rank 0's GPU
    latency = {}
    for nl in (1, 4096):
        if nl > n_roots:
            continue
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ctx.uct_plan_device(model, nl, d_s0, episodes, horizon, gamma, temperature, p, p, d_rng, mpl,
                                plans=d_plans, plan_len=d_len, root_value=d_val, env_steps=d_steps, policy=policy)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        latency["plan_wall_ms_batch_of_{}".format(nl)] = 1e3 * float(np.median(ts))
        latency["kernel_ms_batch_of_{}".format(nl)] = ctx.last_kernel_ms()[0]
        latency["env_steps_batch_of_{}".format(nl)] = int(d_steps[:nl].sum().item())

    pageable_ms, host_kernel_ms = {}, {}

    def host_inclusive(nr, reps):
        """SURVEY.md 8(d) as written: wall time of the batched plan() handing over HOST arrays (MP_MEM_HOST: root
        states and generator records uploaded, plans / values / counts / env-step counters downloaded, stream
        synchronised inside the call); the model upload is excluded, as there."""
        # round 3: the caller's arrays are PINNED host arrays (ctx.plan_buffers: root states in; plans, plan lengths,
        # root values and env-step counters out -- what agent.plan() and the metric read), the generator records stay
        # on the device between calls (ctx.device_rng), and batches above 32 768 roots are pipelined in chunks over
        # side streams inside mp_uct_plan.  Every call still starts from host root states and ends with host results.
        bufs = ctx.plan_buffers(nr, mpl, outputs=("plans", "plan_len", "root_value", "env_steps"))
        bufs["root_state"][:] = s0[:nr]
        rngd = ctx.device_rng(rng0[:nr])
        pp = None if with_prior else p
        ctx.uct_plan(model, bufs["root_state"], episodes, horizon, gamma, temperature, pp, pp, rngd, policy=policy, out=bufs)
        steps, w = 0, 0.0
        for _ in range(reps):
            t1 = time.perf_counter()
            o = ctx.uct_plan(model, bufs["root_state"], episodes, horizon, gamma, temperature, pp, pp, rngd, policy=policy, out=bufs)
            w += time.perf_counter() - t1
            steps += int(o["env_steps"].sum())        # (the metric's counter, read while the clock is stopped: not part of plan())
        host_kernel_ms[nr] = ctx.last_kernel_ms()[0]
        # the round-2 form of the same call for comparison: pageable numpy arrays, all six outputs, records in and out
        s0h, rngh = np.ascontiguousarray(s0[:nr]), rng0[:nr].copy()
        ctx.uct_plan(model, s0h, episodes, horizon, gamma, temperature, pp, pp, rngh, max_plan_len=mpl, policy=policy)
        t2 = time.perf_counter()
        for _ in range(max(reps // 2, 1)):
            ctx.uct_plan(model, s0h, episodes, horizon, gamma, temperature, pp, pp, rngh, max_plan_len=mpl, policy=policy)
        pageable_ms[nr] = 1e3 * (time.perf_counter() - t2) / max(reps // 2, 1)
        rngd.close()
        bufs.close()
        return steps / w, 1e3 * w / reps

    hi_val, hi_ms = host_inclusive(n_roots, 5)
    hi4_val, hi4_ms = host_inclusive(min(4096, n_roots), 10)
    hi1_val, hi1_ms = host_inclusive(1, 20)
    k_ms = float(np.mean(kernel_ms))
    nl4 = min(4096, n_roots)
    res = dict(
        metric="rollout env-steps/sec (UCT plan(), budget=1000)", unit="env-steps/s",
        value=total_env_steps * args.steps / dt, ms_per_step=1e3 * dt / args.steps,
        # the same metric on SURVEY.md 8(d)'s own terms: host arrays in / out (PCIe inclusive) and the 4096-root batch
        value_host_inclusive=sum_over_ranks(hi_val, world), host_inclusive_ms_per_step=hi_ms,
        value_roots4096=latency.get("env_steps_batch_of_4096", 0) / (latency.get("plan_wall_ms_batch_of_4096", float("inf")) * 1e-3),
        value_roots4096_host_inclusive=hi4_val,
        plan_wall_ms_per_root=dict(batch_262144_device=1e3 * dt / args.steps / n_roots,
                                   batch_4096_device=latency.get("plan_wall_ms_batch_of_4096", float("nan")) / nl4,
                                   batch_4096_host_inclusive=hi4_ms / nl4,
                                   single_root_device=latency.get("plan_wall_ms_batch_of_1"),
                                   single_root_host_inclusive=hi1_ms),
        host_inclusive_pageable_all_outputs_ms={str(k): v for k, v in pageable_ms.items()},
        host_inclusive_kernel_ms={str(k): v for k, v in host_kernel_ms.items()},
        dtype="f64",
        config=dict(workload="{}_highway_shaped_S{}_A{}_budget1000_e{}xh{}_roots{}_per_gpu".format(
            "uct_with_vi_boltzmann_prior" if with_prior else "uct", s_, a_, episodes, horizon, n_roots), n_roots_per_gpu=n_roots, n_roots_total=n_roots * world,
            states=s_, actions=a_, episodes=episodes, horizon=horizon, gamma=gamma,
            env_steps_per_step=total_env_steps, plan_ms_per_root=1e3 * dt / args.steps / n_roots, latency=latency,
            value_definition="`value` = device-resident: roots, generator records and results stay in HBM (this tier's "
                             "bench contract); SURVEY 8(d)'s host-inclusive form of the same metric (host arrays in and out, the "
                             "transfers inside the call) is `value_host_inclusive` in this line",
            measured_mean_selection_depth=mean_depth, measured_expansions_per_episode=expansions / float(n_smp * episodes),
            algorithmic_bytes_per_env_step=bytes_per_step,
            parallelism="roots sharded over {} GPU(s); per step ONE all_gather_into_tensor of the packed per-root rows "
                        "{{plan[0], root value, env_steps (status in its top byte)}} ({} B per root), product path "
                        "rl_agents_amd.distributed.ShardedDevicePlan".format(world, sp.row_bytes) if sp is not None else
                        "single GPU"),
        roofline=dict(bound="hbm", achieved=bytes_per_step * env_steps / (k_ms * 1e-3) / 1e9,
                      peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel={"uct_row_shared": "uct_row_kernel<5, SHARED> (four roots per wavefront, transitions + trees in LDS)",
                              "uct_lone": "uct_lone_kernel<5> (one root per workgroup)",
                              "uct_lone_mw": "uct_lone_kernel<5, MW> (2 / 4 / 8 roots per workgroup, a wavefront each, one copy of the transitions in LDS)",
                              "uct_quad": "uct_kernel<5, ENV_TABLE_LDSR, QD> (four lanes per root)"}.get(variant) or
                      "uct_kernel<5, {}>".format("ENV_TABLE, per-state policies" if with_prior else
                                                 ("ENV_TABLE_LDSR (model resident in LDS)" if variant == "uct_ldsr" else "ENV_TABLE")),
                      kernel_variant=variant, model_bytes_staged_per_launch=staged,
                      hbm_side_bytes_per_launch=None if bytes_per_step_hbm is None else bytes_per_step_hbm * env_steps,
                      frac_hbm_side=None if bytes_per_step_hbm is None else bytes_per_step_hbm * env_steps / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      kernel_ms=k_ms, algorithmic_bytes_per_launch=bytes_per_step * env_steps,
                      note="algorithmic bytes = SURVEY 8(d) terms with the depth / expansions measured on this launch's "
                           "trees" + ("; this kernel serves the model term (13 B per env step) from LDS -- staged once per workgroup -- so "
                                      "`frac` is the rate at which the ALGORITHM's bytes are consumed, not HBM traffic: "
                                      "`frac_hbm_side` prices what must cross HBM (tree terms + staging) and `traffic` is what "
                                      "the counters saw; the kernel is bound by vector-ALU issue" if variant == "uct_ldsr" else "") + ("; the per-state policy tables (L2-resident by construction, like the 800 KB model) "
                                      "are not charged" if with_prior else "")),
    )
    if general is not None:
        res["general_model_kernel"] = general
    if sp is not None and world > 1:
        # the price of the one exchange of the sharded path: HIP events from the end of the planner's kernel to the end
        # of the unpack (pack + all_gather_into_tensor + unpack), on the side stream the next launch overlaps
        ex = float(np.mean(exchange_ms)) if exchange_ms else None
        step_ms = 1e3 * dt / args.steps
        res["exchange"] = dict(
            payload=sp.payload, row_bytes=int(sp.row_bytes), bytes_sent_per_rank_per_step=int(sp.row_bytes) * int(sp.per),
            bytes_received_per_rank_per_step=int(sp.row_bytes) * int(sp.per) * world, exchange_ms=ex,
            on_side_stream=bool(sp.overlapped), backend="rccl" if sp.on_device else ("gloo via host" if sp.grouped else None),
            kernel_ms=k_ms, step_ms=step_ms,
            hidden_ms=None if ex is None else max(0.0, min(ex, k_ms + ex - step_ms)),
            note="exchange_ms = pack + all_gather_into_tensor + unpack (HIP events around them); hidden_ms = how much of it the "
                 "timed loop did not pay (kernel_ms + exchange_ms - step_ms, clamped to [0, exchange_ms]): the side stream runs "
                 "it under the next step's kernel" if world > 1 else "single rank: no process group, no exchange")
    # (the same run also launches the record-gather kernel on this grid -- `general_model_kernel` -- so the counters are looked
    # up by the full template name: ENV 3 = model resident in LDS, 0 = records gathered)
    kname = "uct_kernel<{}, {},".format(a_, 3 if variant in ("uct_ldsr", "uct_quad") else 0)
    if variant == "uct_row_shared":       # (257 .. 4096 roots: the row kernel on a shared model; 1024-thread workgroups)
        add_traffic(res["roofline"], "uct", "uct_row_kernel<{}, true>".format(a_), None)
    elif variant == "uct_lone":
        add_traffic(res["roofline"], "uct", "uct_lone_kernel<{}".format(a_), n_roots * 1024)
    elif variant == "uct_lone_mw":
        add_traffic(res["roofline"], "uct", "uct_lone_kernel<{}, false, true>".format(a_), None)
    else:
        add_traffic(res["roofline"], "uct_prior" if with_prior else "uct", "uct_kernel" if with_prior else kname, n_roots)
    if not with_prior and rank == 0 and world == 1 and not args.headline_only:
        # the default run measures the headline kernel's HBM traffic itself (it used to be read from a committed
        # summary); the committed figure stays beside it as `traffic_committed`
        live, raw = live_pmc_traffic(kname, n_roots, "scattered", ["--roots", str(n_roots)])
        roof = res["roofline"]
        roof["traffic_committed"] = roof["traffic"]
        if live is not None:
            roof["traffic"], roof["traffic_counters"] = live, raw
            roof["traffic_frac"] = live / (roof["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            roof["traffic_live"] = raw              # (why not: the committed summary is what `traffic` holds then)
    if cross is not None:
        res["_cross"] = cross
    if not args.no_parity_sample:
        # the timed launch itself, replayed: generator records back to their initial values, one more step at the
        # benchmarked geometry, a sample of its roots through the CPU oracle (every rank steps: the exchange is collective)
        d_rng.copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        if rank == 0:
            from oracle import oracle
            buf = last_buffers()
            idx = sample_rows(n_roots)
            ti = torch.from_numpy(idx).to(dev)
            got = {k: buf[k][ti].cpu().numpy() for k in ("plans", "plan_len", "value", "env_steps")}
            ref = oracle.uct_plan_batch(t, r, term, s0[idx], episodes, horizon, gamma, temperature, p, p, rng0[idx],
                                        max_plan_len=mpl, n_threads=host_cores())
            ok = (np.array_equal(got["plans"], ref["plans"]) and np.array_equal(got["plan_len"], ref["plan_len"])
                  and np.array_equal(got["value"], ref["root_value"]) and np.array_equal(got["env_steps"], ref["env_steps"]))
            res["parity_sample"] = parity_record(ok, "{} roots of a {}-root launch vs oracle.uct_plan_batch: plans, plan_len, root "
                                                 "value, env_steps bit for bit".format(len(idx), n_roots))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the host baseline is an N = 1 figure
        from oracle import oracle
        cores = host_cores()
        n_cpu = max(2048, 64 * cores)
        cpu_rng = seed_states(np.arange(n_cpu))
        oracle.uct_plan_batch(t, r, term, all_roots[:64], episodes, horizon, gamma, temperature, p, p, cpu_rng[:64],
                              n_threads=cores)
        done, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds:
            o = oracle.uct_plan_batch(t, r, term, np.resize(all_roots, n_cpu), episodes, horizon, gamma, temperature,
                                      p, p, cpu_rng, n_threads=cores)
            done += int(o["env_steps"].sum())
        cdt = time.perf_counter() - t1
        t2 = time.perf_counter()
        o1 = oracle.uct_plan_batch(t, r, term, np.resize(all_roots, 256), episodes, horizon, gamma, temperature, p, p,
                                   cpu_rng[:256], n_threads=1)
        one = int(o1["env_steps"].sum()) / (time.perf_counter() - t2)
        res["cpu_baseline"] = dict(value=done / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c uct_plan_batch, OpenMP over roots, {} roots per "
                                          "batch repeated for {:.1f} s, same tables/params".format(n_cpu, cdt),
                                   value_1core=one)
    ctx.synchronize()
    return res


def bench_uct_cartpole(args, rank, world, local):
    """BASELINE config C3: UCT on closed-form CartPole-v0, budget 1000 as 20 episodes x horizon 50, 4096 roots per GPU."""
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import CartPoleEnv
    n_roots = args.roots or 4096
    episodes, horizon, gamma, temperature = 20, 50, 0.8, 2 / (1 - 0.8)
    params = CartPoleEnv().cartpole_params()
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_cartpole(params)
    gids = np.arange(rank * n_roots, (rank + 1) * n_roots)
    x0 = np.random.Generator(np.random.PCG64(0)).uniform(-0.05, 0.05, size=(world * n_roots, 4))[gids]
    dev = torch.device("cuda", local)
    d_x0 = torch.from_numpy(np.ascontiguousarray(x0)).to(dev)
    rng0 = seed_states(gids)
    d_rng = torch.from_numpy(rng0.view(np.int64)).to(dev)
    mpl = 8
    d_plans = torch.empty((n_roots, mpl), dtype=torch.int32, device=dev)
    d_len = torch.empty(n_roots, dtype=torch.int32, device=dev)
    d_val = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_steps = torch.empty(n_roots, dtype=torch.int64, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    p = np.ones(2) / 2

    def step():
        ctx.uct_plan_device(model, n_roots, d_x0, episodes, horizon, gamma, temperature, p, p, d_rng, mpl,
                            plans=d_plans, plan_len=d_len, root_value=d_val, env_steps=d_steps)
        d_total.add_(d_steps.sum())

    for _ in range(args.warmup):
        step()
    d_total.zero_()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    timed = int(d_total.item())
    step()
    k_ms = ctx.last_kernel_ms()[0]
    env_steps = int(d_steps.sum().item())
    total = sum_over_ranks(float(timed), world) / args.steps
    # closed-form env: no model bytes; per root 32 B state in + tree terms: 16*A*d + 24*(d+1) + 24*A per episode
    alg = n_roots * (32.0 + episodes * (16 * 2 * 3 + 24 * 4 + 24 * 2))
    res = dict(
        metric="rollout env-steps/sec (UCT plan(), CartPole-v0, budget=1000)", unit="env-steps/s",
        value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="uct_cartpole_v0_budget1000_e{}xh{}_roots{}_per_gpu".format(episodes, horizon, n_roots),
                    n_roots_per_gpu=n_roots, n_roots_total=n_roots * world, episodes=episodes, horizon=horizon,
                    gamma=gamma, env_steps_per_step=total, plan_ms_per_root=1e3 * dt / args.steps / n_roots,
                    parallelism="roots sharded over {} GPU(s)".format(world)),
        roofline=dict(bound="hbm", achieved=alg / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", traffic=None,
                      kernel="uct_kernel<2, ENV_CARTPOLE>", kernel_ms=k_ms, algorithmic_bytes_per_launch=alg,
                      note="state lives in registers: compute/latency bound by construction"),
    )
    add_traffic(res["roofline"], "uct_cartpole", "uct_kernel<2, 2", None)
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        d_rng.copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        idx = sample_rows(n_roots)
        ti = torch.from_numpy(idx).to(dev)
        ref = oracle.uct_plan_batch(None, None, None, x0[idx], episodes, horizon, gamma, temperature, p, p, rng0[idx],
                                    max_plan_len=mpl, n_threads=host_cores(), cartpole=params)
        same = ((d_plans[ti].cpu().numpy() == ref["plans"]).all(axis=1) & (d_steps[ti].cpu().numpy() == ref["env_steps"])
                & (d_val[ti].cpu().numpy() == ref["root_value"]))
        # sin / cos of the pole angle are the host libm's algorithm restated on the device (csrc/libm_sincos.hpp): bit for bit
        # when one of the two forms reproduces this host's libm (variant 1 / 2), else the device math library and the old tolerance
        variant = native.libm_sincos_variant()
        # (no matching form = a FAILING sample: the line must not say "ok" on a host where bit-exactness was not even attempted)
        need = 1.0
        res["parity_sample"] = parity_record(bool(same.mean() >= need) and variant in (1, 2), "{} roots of a {}-root launch vs oracle.uct_plan_batch "
                                             "(CartPole): plans, env_steps, root value; {}".format(
                                                 len(idx), n_roots, "bit for bit (host libm's sin / cos restated on the device, form {})".format(variant)
                                                 if need == 1.0 else "tolerance >= 98 % of the sample identical (device sincos: no restated form matched this host's libm)"),
                                             identical_fraction=float(same.mean()), libm_sincos_variant=variant)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # the host baseline is an N = 1 figure
        from oracle import oracle
        cores = host_cores()
        n_cpu = 64 * cores
        xs = np.resize(x0, (n_cpu, 4))
        cpu_rng = seed_states(np.arange(n_cpu))
        done, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds:
            o = oracle.uct_plan_batch(None, None, None, xs, episodes, horizon, gamma, temperature, p, p, cpu_rng,
                                      n_threads=cores, cartpole=params)
            done += int(o["env_steps"].sum())
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c uct_plan_batch (CartPole), OpenMP over {} roots "
                                          "per batch for {:.1f} s".format(n_cpu, cdt))
    return res


def bench_uct_stoch(args, rank, world, local):
    """MCTS on a STOCHASTIC finite MDP, closed loop (uct_stoch.hip; VERDICT r2 task 7): the highway-shaped table made
    `sparse` -- every (s, a) reaches its intended next state with probability 0.8 and the IDLE successor with 0.2 -- budget
    1000 as 33 episodes x horizon 30, observation nodes keyed by the sampled next state.  A step = one batched plan()."""
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    n_roots = args.roots or 262144      # (as the headline workload; 65 536 roots are one wave per SIMD: 0.85 ms)
    episodes, horizon, gamma, temperature = 33, 30, 0.8, 2 / (1 - 0.8)
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    nxt = np.stack([t, np.repeat(t[:, 1:2], a_, axis=1)], axis=-1).astype(np.int64)        # [S, A, 2]: intended, IDLE's
    pr = np.broadcast_to(np.array([0.8, 0.2]), (s_, a_, 2)).copy()
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    model = ctx.load_sparse(pr, nxt, r, term)
    gids = np.arange(rank * n_roots, (rank + 1) * n_roots)
    non_term = np.flatnonzero(~np.asarray(term))
    all_roots = np.random.Generator(np.random.PCG64(12345)).choice(non_term, size=world * n_roots).astype(np.int32)
    s0 = all_roots[gids]
    rng0 = seed_states(gids)
    erng0 = seed_states(gids, base_seed=10 ** 6)
    dev = torch.device("cuda", local)
    d_s0 = torch.from_numpy(s0).to(dev)
    d_rng = torch.from_numpy(rng0.view(np.int64)).to(dev)
    d_erng = torch.from_numpy(erng0.view(np.int64)).to(dev)
    mpl = 8
    d_plans = torch.empty((n_roots, mpl), dtype=torch.int32, device=dev)
    d_len = torch.empty(n_roots, dtype=torch.int32, device=dev)
    d_val = torch.empty(n_roots, dtype=torch.float64, device=dev)
    d_steps = torch.empty(n_roots, dtype=torch.int64, device=dev)
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    p = np.ones(a_) / a_
    lib = ctx._lib

    def step():
        native._check(lib.mp_uct_plan_stochastic(ctx._h, model._h, n_roots, native._ptr(d_s0), None, episodes, horizon, gamma,
                                                 temperature, native._ptr(p), native._ptr(p), 1, native._ptr(d_rng),
                                                 native._ptr(d_erng), mpl, native._ptr(d_plans), native._ptr(d_len),
                                                 native._ptr(d_val), None, None, native._ptr(d_steps), native.MP_MEM_DEVICE))
        d_total.add_(d_steps.sum())

    for _ in range(args.warmup):
        step()
    d_total.zero_()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    timed = int(d_total.item())
    step()
    k_ms = ctx.last_kernel_ms()[0]
    env_steps = int(d_steps.sum().item())
    total = sum_over_ranks(float(timed), world) / args.steps
    # algorithmic bytes, measured quantities: per env step the fused 16-byte record of (s, a) (threshold, two successors,
    # reward index, terminal flags: this model has 136 distinct rewards); per scored level the |A| children's 16-byte halves {value, count, first}; per path node a
    # 16-byte read-modify-write of that half; per created node both halves (32 B)
    sample = np.unique(np.linspace(0, n_roots - 1, 65).astype(np.int64))
    nodes = sel = 0
    for root in sample:
        tr = ctx.uct_stoch_tree(int(root))
        nodes += len(tr["parent"])
        sel += int(tr["count"][(tr["is_obs"] == 0) & (tr["parent"] >= 0)].sum())        # visits of action nodes = selection steps
    smp_env = float(d_steps[torch.from_numpy(sample).to(dev)].sum().item())
    bytes_per_step = (16.0 * smp_env + 16.0 * a_ * sel + 2 * 16.0 * (2 * sel + len(sample) * episodes) + 32.0 * nodes) / smp_env
    res = dict(
        metric="rollout env-steps/sec (UCT plan() on a stochastic model, closed loop, budget=1000)", unit="env-steps/s",
        value=total * args.steps / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64",
        config=dict(workload="uct_stochastic_sparse_highway_shaped_S{}_A{}_B2_closed_loop_budget1000_e{}xh{}_roots{}_per_gpu".format(
            s_, a_, episodes, horizon, n_roots), n_roots_per_gpu=n_roots, n_roots_total=n_roots * world, episodes=episodes,
            horizon=horizon, gamma=gamma, env_steps_per_step=total, measured_nodes_per_tree=nodes / float(len(sample)),
            algorithmic_bytes_per_env_step=bytes_per_step, parallelism="roots sharded over {} GPU(s)".format(world)),
        roofline=dict(bound="hbm", achieved=bytes_per_step * env_steps / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel="uct_stoch_kernel", kernel_ms=k_ms, algorithmic_bytes_per_launch=bytes_per_step * env_steps,
                      traffic=None, traffic_frac=None,
                      note="one root per lane, root-major trees (closed-loop node ids do not advance in lock-step); the texture-"
                           "address units are 47 % busy at 65 536 roots and 83 % at 262 144 (profiles/r03_uct_stoch_units.txt, "
                           "before the 16-byte records): the bound is the count of scattered vector-memory instructions"),
    )
    add_traffic(res["roofline"], "uct_stoch", "uct_stoch_kernel", n_roots)
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        d_rng.copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        idx = sample_rows(n_roots)
        ti = torch.from_numpy(idx).to(dev)
        ref = oracle.uct_plan_stoch_batch("sparse", pr, r, term, s0[idx], episodes, horizon, gamma, temperature, p, p, rng0[idx],
                                          erng0[idx], next_states=nxt, closed_loop=True, max_plan_len=mpl, n_threads=host_cores())
        ok = (np.array_equal(d_plans[ti].cpu().numpy(), ref["plans"]) and np.array_equal(d_len[ti].cpu().numpy(), ref["plan_len"])
              and np.array_equal(d_val[ti].cpu().numpy(), ref["root_value"])
              and np.array_equal(d_steps[ti].cpu().numpy(), ref["env_steps"]))
        res["parity_sample"] = parity_record(ok, "{} roots of a {}-root launch vs oracle.uct_plan_stoch_batch (closed loop): plans "
                                             "with observation keys, plan_len, root value, env_steps bit for bit".format(len(idx), n_roots))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        cores = host_cores()
        n_cpu = 64 * cores
        done, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < args.cpu_seconds:
            o = oracle.uct_plan_stoch_batch("sparse", pr, r, term, np.resize(all_roots, n_cpu), episodes, horizon, gamma,
                                            temperature, p, p, seed_states(np.arange(n_cpu)), seed_states(np.arange(n_cpu), 10 ** 6),
                                            next_states=nxt, closed_loop=True, n_threads=cores)
            done += int(o["env_steps"].sum())
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="env-steps/s", cores=cores, kind="port",
                                   sample="oracle/planning_oracle.c orc_uct_plan_stoch_batch, OpenMP over {} roots per batch for "
                                          "{:.1f} s".format(n_cpu, cdt))
    return res


def bench_uct_per_root_model(args, rank, world, local):
    """UCT with ONE MDP PER ROOT (round 5): every root of the batch plans on its own highway-shaped (3, 4, 10) table -- the batch
    of highway episodes of trainer/evaluation.py:139-194, one environment each -- through mp_uct_plan_models on a batch model;
    budget 1000 as 33 x 30.  Beside it: the same roots on ONE shared table (the kernel the other UCT rows measure)."""
    import torch
    from rl_agents_amd import native
    n_roots = args.roots or 4096
    episodes, horizon, gamma, temperature = 33, 30, 0.8, 2 / (1 - 0.8)
    tr, rw, tm = _episode_tables(n_roots, (3, 4, 10), seed0=7 + 100000 * rank, distinct=4096)
    s_, a_ = tr.shape[1:]
    dev = torch.device("cuda", local)
    ctx = native.Context(local, torch.cuda.current_stream().cuda_stream)
    t_load = time.perf_counter()
    model = ctx.load_table_batch(tr, rw, tm)
    load_ms = 1e3 * (time.perf_counter() - t_load)
    t_upd = time.perf_counter()
    model.update_tables(0, tr, rw, tm)              # what a step of the episodes costs on the upload side: every table replaced
    ctx.synchronize()
    upd_ms = 1e3 * (time.perf_counter() - t_upd)
    g = np.random.Generator(np.random.PCG64(1 + rank))
    s0 = g.integers(0, s_, n_roots).astype(np.int32)
    rng0 = seed_states(np.arange(rank * n_roots, (rank + 1) * n_roots))
    mpl = 8
    p = np.ones(a_) / a_
    d = dict(mi=torch.arange(n_roots, dtype=torch.int32, device=dev), s0=torch.from_numpy(s0).to(dev),
             rng=torch.from_numpy(rng0.view(np.int64)).to(dev), plans=torch.full((n_roots, mpl), -1, dtype=torch.int32, device=dev),
             plan_len=torch.zeros(n_roots, dtype=torch.int32, device=dev), value=torch.zeros(n_roots, dtype=torch.float64, device=dev),
             steps=torch.zeros(n_roots, dtype=torch.int64, device=dev))
    d_total = torch.zeros(1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def step(m=None, mi=True):
        ctx.uct_plan_device(m or model, n_roots, d["s0"], episodes, horizon, gamma, temperature, p, p, d["rng"], mpl, plans=d["plans"],
                            plan_len=d["plan_len"], root_value=d["value"], env_steps=d["steps"], model_index=d["mi"] if mi else None)
        d_total.add_(d["steps"].sum())

    for _ in range(args.warmup):
        step()
    barrier(world)
    d_total.zero_()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier(world)
    dt = max_over_ranks(time.perf_counter() - t0, world)
    timed = sum_over_ranks(float(d_total.item()), world)
    k_ms = []
    for _ in range(5):
        step()
        k_ms.append(ctx.last_kernel_ms()[0])
    k_ms = float(np.mean(k_ms))
    variant = ctx.last_kernel_variant()
    env_steps = int(d["steps"].sum().item())
    sample = np.unique(np.linspace(0, n_roots - 1, 129).astype(np.int64))
    sel_steps = expansions = 0
    smp_steps = int(d["steps"][torch.from_numpy(sample).to(dev)].sum().item())
    for root in sample:
        tree = ctx.uct_tree(int(root))
        sel_steps += int(tree["count"][1:].sum())
        expansions += int((tree["first_child"] >= 0).sum())
    bytes_per_step = (13.0 * smp_steps + 16.0 * a_ * sel_steps + 24.0 * (sel_steps + len(sample) * episodes) + 24.0 * a_ * expansions) / smp_steps
    # the same roots on ONE shared table: the kernel every other UCT row of this file measures
    shared = ctx.load_table(tr[0], rw[0], tm[0])
    sh_ms = []
    step(shared, mi=False)
    for _ in range(5):
        step(shared, mi=False)
        sh_ms.append(ctx.last_kernel_ms()[0])
    sh_ms, sh_variant, sh_steps = float(np.mean(sh_ms)), ctx.last_kernel_variant(), int(d["steps"].sum().item())
    shared.close()
    res = dict(
        metric="rollout env-steps/sec (UCT plan(), budget=1000, one MDP per root)", unit="env-steps/s",
        value=timed / dt, ms_per_step=1e3 * dt / args.steps, dtype="f64", variant=variant,
        vs_shared_model_kernel=dict(per_root_model_kernel_ms=k_ms, shared_model_kernel_ms=sh_ms, shared_model_variant=sh_variant,
                                    ratio=k_ms / sh_ms, per_root_env_steps_per_s=env_steps / (k_ms * 1e-3),
                                    shared_env_steps_per_s=sh_steps / (sh_ms * 1e-3),
                                    note="same roots, budget and policies; `shared` plans every root on table 0"),
        config=dict(workload="uct_per_root_model_highway_shaped_S{}_A{}_budget1000_e{}xh{}_roots{}_per_gpu".format(s_, a_, episodes, horizon, n_roots),
                    n_roots_per_gpu=n_roots, states_per_mdp=s_, actions=a_, episodes=episodes, horizon=horizon, gamma=gamma,
                    model_bytes=int(n_roots) * s_ * a_ * 16, model_load_ms=load_ms, replace_every_table_ms=upd_ms,
                    algorithmic_bytes_per_env_step=bytes_per_step,
                    parallelism="roots (episodes) sharded over {} GPU(s), no collective".format(world)),
        roofline=dict(bound="hbm", achieved=bytes_per_step * env_steps / (k_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                      kernel={"uct_row_each": "uct_row_kernel<5> (4 roots per wavefront, each root's MDP + tree in LDS)",
                              "uct_lone_each": "uct_lone_kernel<5, EACH> (a wavefront per root, its MDP + tree in LDS)"}.get(
                                  variant, "uct_kernel<5, ENV_TABLE> on the union model ({})".format(variant)), kernel_ms=k_ms,
                      kernel_variant=variant, algorithmic_bytes_per_launch=bytes_per_step * env_steps,
                      # what must cross HBM when the MDP and the tree live in LDS: every root's 16-byte records staged once, its
                      # tree written out once (export / re-rooting), roots / generator records / results
                      hbm_side_bytes_per_launch=float(n_roots) * (s_ * a_ * 16 + (1 + episodes * a_) * 16 + 4 + 96 + mpl * 4 + 24),
                      note="algorithmic bytes = SURVEY 8(d) terms with depth / expansions measured on this launch's trees; the "
                           "LDS-resident kernels serve them from LDS after staging each root's OWN {} B table once: `frac` is the "
                           "rate at which the algorithm's bytes are consumed, `traffic` what the counters saw; the kernel is bound "
                           "by the dependency chain of one episode (profiles/r06_uct_row.md)".format(s_ * a_ * 16)),
    )
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
    add_traffic(res["roofline"], "uct_per_root_model", {"uct_row_each": "uct_row_kernel", "uct_lone_each": "uct_lone_kernel"}.get(variant, "uct_kernel"),
                {"uct_row_each": -(-n_roots // 4) * 64, "uct_lone_each": n_roots * (64 if s_ * a_ <= 4096 else 256)}.get(variant, n_roots))
    if not args.no_parity_sample and rank == 0:
        from oracle import oracle
        d["rng"].copy_(torch.from_numpy(rng0.view(np.int64)).to(dev))
        step()
        idx = sample_rows(n_roots)
        ti = torch.from_numpy(idx).to(dev)
        ref = oracle.uct_plan_each(tr, rw, tm, idx, s0[idx], episodes, horizon, gamma, temperature, p, p, rng0[idx], max_plan_len=mpl)
        ok = (np.array_equal(d["plans"][ti].cpu().numpy(), ref["plans"]) and np.array_equal(d["value"][ti].cpu().numpy(), ref["root_value"])
              and np.array_equal(d["steps"][ti].cpu().numpy(), ref["env_steps"])
              and np.array_equal(d["rng"][ti].cpu().numpy().view(np.uint64), ref["rng_after"]))
        res["parity_sample"] = parity_record(ok, "{} roots of the timed {}-root launch vs per-root oracle plans on each root's own table: "
                                             "plans, root value, env_steps, generator state bit for bit".format(len(idx), n_roots))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        t1, done, i = time.perf_counter(), 0, 0
        cpu_rng = seed_states(np.arange(256))
        while time.perf_counter() - t1 < args.cpu_seconds:
            o = oracle.uct_plan_batch(tr[i % n_roots], rw[i % n_roots], tm[i % n_roots], np.resize(s0, 256), episodes, horizon, gamma,
                                      temperature, p, p, cpu_rng, n_threads=host_cores())
            done += int(o["env_steps"].sum())
            i += 1
        cdt = time.perf_counter() - t1
        res["cpu_baseline"] = dict(value=done / cdt, unit="env-steps/s", cores=host_cores(), kind="port",
                                   sample="oracle/planning_oracle.c uct_plan_batch, 256 roots per table, {} tables in {:.1f} s".format(i, cdt))
    model.close()
    return res
