#!/usr/bin/env python3
"""Benchmark of the planning hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload uct|opd|vi|vi_dense] [--roots R]

Default workload = the BASELINE.json headline: MCTS/UCT on a highway-shaped finite MDP
(S = 10 000, |A| = 5), budget 1000 as 33 episodes x horizon 30, 262 144 independent roots per GPU (the saturated batch;
SURVEY 8(d)'s own batch sizes -- 4096 roots and a single root -- are measured in the same run and printed beside it:
`value_roots4096`, `plan_wall_ms_per_root`).
A "step" is one batched plan() call over all roots of this rank (inputs already in HBM).
`value` = environment transitions executed inside plan() by ALL ranks / wall time (max over ranks).

Multi-GPU: `python bench.py --gpus N` launches its own N ranks (one per GPU, `python -m torch.distributed.run` on
127.0.0.1) when it is not already running under a launcher; the driver's own `torch.distributed.run ... bench.py --gpus N`
form works unchanged.  A run whose process group does not have exactly --gpus ranks exits non-zero.  Roots are sharded
over ranks with no data-path collective (weak scaling: the same roots per GPU); the only exchange is ONE RCCL
all_gather_into_tensor of the packed per-root rows per step ({plan[0], root value, env_steps}: 20 B per root; `exchange`
in the line prices it: pack + collective + unpack in ms, and whether it ran under the next launch) -- the PRODUCT's sharded path
(rl_agents_amd.distributed.ShardedDevicePlan: agent -> planner -> mp_uct_plan -> mp_pack_rows -> all_gather ->
mp_unpack_rows, nothing through the host), cross-checked inside the run: rank 0 re-plans a sample of ANOTHER rank's roots
and compares it with what the gather delivered (`ranks.cross_check`).

With N > 1 the default run also shards BASELINE configs C4 (OPD, 1024 roots per GPU) and C5 (dense robust VI, one 25 GB row
block per GPU with the per-sweep all_gather of V) and attaches them under `workloads`.
The default run (headline workload, one GPU) also runs every other workload of the path for a bounded slice and attaches
`workloads: {name: {value, kernel_ms, frac, traffic_frac, parity_sample}}` to the one JSON line; `parity_sample` replays a
sample of the timed launch's own roots (or three sweeps) through the CPU oracle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
DENSE_SHARD_MODE = "mfma"       # --workload rvi_dense_shard without --dense-mode
MFMA_F64_PEAK_TFLOPS = 78.6    # MI355X FP64 matrix peak (vendor figure; the guide lists no f64 row)
# algorithmic HBM bytes per unit of work, SURVEY.md §8(d) / DESIGN.md §Kernels


def calibration():
    """FETCH_SIZE / WRITE_SIZE correction factors measured on this repo's own access patterns
    (tools/gather_calib.hip -> profiles/*_gather_calib.json, 'factors'): true fabric bytes per byte the counter tallies,
    for wide coalesced streams and for the scattered 16-byte records of the tree-search kernels.  Without a committed
    calibration only the guide's stream factor (x2 on FETCH_SIZE) is known and scattered traffic is reported raw."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_gather_calib.json")))
    if files:
        try:
            f = json.load(open(files[-1])).get("factors")
            if f:
                return dict(f, source=os.path.basename(files[-1]))
        except (OSError, ValueError):
            pass
    return dict(fetch_stream=2.0, write_stream=1.0, fetch_scattered=1.0, write_scattered=1.0, source="uncalibrated (raw)")


def pmc_traffic(workload, kernel_substr, grid_threads, pattern="scattered"):
    """HBM bytes per launch of one kernel from the committed PMC summary (profiles/*_pmc.json, produced by
    tools/profile_gpu.sh + tools/summarize_profiles.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
    passes of this same command): (f_fetch * FETCH_SIZE + f_write * WRITE_SIZE) * 1024 with the factors of
    calibration() for this kernel's access pattern ("stream": wide coalesced loads, e.g. vi_dense_q; "scattered":
    16-byte records at random addresses, the tree-search kernels).  -> (bytes, raw dict) or (None, None) when no
    summary for this launch geometry is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.json")))
    if not files:
        return None, None
    entry = {}
    for path in reversed(files):                     # the newest summary that holds this workload
        try:
            entry = json.load(open(path)).get(workload, {})
        except (OSError, ValueError):
            entry = {}
        if entry:
            files = [path]
            break
    cal = calibration()
    for key, v in entry.items():
        # grid_threads None: the kernel is launched on one geometry only in this workload (dense VI: the column split
        # decides the grid, not the bench)
        if kernel_substr in key and (grid_threads is None or key.endswith("grid={}".format(grid_threads))):
            if "FETCH_SIZE_KB_per_launch" in v and "WRITE_SIZE_KB_per_launch" in v:
                ff, fw = cal["fetch_" + pattern], cal["write_" + pattern]
                raw = dict(FETCH_SIZE_bytes=v["FETCH_SIZE_KB_per_launch"] * 1024.0,
                           WRITE_SIZE_bytes=v["WRITE_SIZE_KB_per_launch"] * 1024.0, fetch_factor=ff, write_factor=fw,
                           calibration=cal["source"], summary=os.path.basename(files[-1]))
                return ff * raw["FETCH_SIZE_bytes"] + fw * raw["WRITE_SIZE_bytes"], raw
    return None, None


def live_pmc_traffic(kernel_substr, grid_threads, pattern, extra_args):
    """HBM bytes per launch of the headline kernel MEASURED IN THIS RUN: this very script is run twice more for a few steps
    under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, counters only, as the guide
    prescribes), and the kernel's launches at the benchmarked grid are averaged.  -> (bytes, raw dict) or (None, reason).
    BENCH_NO_LIVE_PMC=1 skips it (and so does running under it)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("BENCH_NO_LIVE_PMC") or os.environ.get("BENCH_UNDER_PMC"):
        return None, "skipped"
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    cal = calibration()
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = [prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--headline-only", "--no-cpu-baseline", "--no-parity-sample", "--steps", "3",
               "--warmup", "1"] + extra_args
        try:
            subprocess.run(cmd, env=dict(os.environ, BENCH_UNDER_PMC="1", TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=240, check=False)
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and kernel_substr in row["Kernel_Name"] and int(row["Grid_Size"]) == grid_threads:
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, "no {} rows for the kernel".format(ctr)
            vals[ctr] = (sum(got) / len(got), len(got))
        except (OSError, subprocess.SubprocessError, ValueError, KeyError) as e:
            return None, "{}: {}".format(type(e).__name__, e)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    ff, fw = cal["fetch_" + pattern], cal["write_" + pattern]
    raw = dict(FETCH_SIZE_bytes=vals["FETCH_SIZE"][0] * 1024.0, WRITE_SIZE_bytes=vals["WRITE_SIZE"][0] * 1024.0, fetch_factor=ff,
               write_factor=fw, calibration=cal["source"], launches=[vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]],
               source="measured in this run: two rocprofv3 --pmc passes of `bench.py --headline-only --steps 3`")
    return ff * raw["FETCH_SIZE_bytes"] + fw * raw["WRITE_SIZE_bytes"], raw


def add_traffic(roofline, workload, kernel_substr, grid_threads, pattern="scattered"):
    """roofline.traffic (+ traffic_frac = traffic / kernel time / peak, the MEASURED HBM fraction, next to the contract's
    algorithmic one) from the committed PMC passes."""
    traffic, raw = pmc_traffic(workload, kernel_substr, grid_threads, pattern)
    roofline["traffic"] = traffic
    roofline["traffic_frac"] = None if traffic is None else traffic / (roofline["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    roofline["traffic_counters"] = raw
    roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS


def reference_python(workload):
    """profiles/reference_cpu.json[workload]: the unmodified reference timed by tests/golden/gen/time_reference.py."""
    try:
        rec = json.load(open(os.path.join(REPO, "profiles", "reference_cpu.json")))
    except (OSError, ValueError):
        return None
    entry = rec.get("workloads", {}).get(workload)
    if entry is None:
        return None
    return dict(entry, host=rec.get("host"), generated_by="tests/golden/gen/time_reference.py")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="uct", choices=["uct", "uct_prior", "uct_cartpole", "uct_stoch", "opd", "ropd", "saopd", "vi", "rvi", "vi_dense", "vi_dense_exact", "rvi_dense_shard", "vi_batch",
                                                         "uct_per_root_model"])
    ap.add_argument("--roots", type=int, default=None, help="roots per GPU (default 262144 uct, 1024 opd)")
    ap.add_argument("--states", type=int, default=None, help="|S| override (vi_dense default 10000)")
    ap.add_argument("--dense-mode", default=None, choices=["mfma", "exact"],
                    help="dense VI workloads: contraction on the f64 matrix cores or in numpy's order of additions (bit-exact)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--headline-only", action="store_true", help="skip the bounded slices of the other workloads")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip the oracle replay of a sample of the timed launch")
    return ap.parse_args()


def host_cores():
    """Cores this process may really use: min(affinity, cgroup cpu.max quota) -- the GPU box reports 256 logical
    CPUs but its container is capped (cpu.max 1600000/100000 = 16 CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def visible_devices():
    import torch
    return torch.cuda.device_count()


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: become the launcher -- N ranks of this very command line under
    torch.distributed.run on 127.0.0.1 (one process per GPU, RCCL) -- and return its exit code.  On a box with fewer
    than N GPUs the ranks share device 0 over a gloo group (RCCL refuses two ranks per device): a DRY RUN of the N > 1
    code path, flagged in the JSON line (`ranks.dry_run_same_device`), never a scaling measurement."""
    import socket
    import subprocess
    n_dev = visible_devices()
    env = dict(os.environ)
    if n_dev < args.gpus:
        print("bench.py: --gpus {} but {} device(s) visible: DRY RUN with all ranks on device 0 (gloo group); not a "
              "scaling measurement".format(args.gpus, n_dev), file=sys.stderr)
        env["BENCH_SAME_DEVICE"] = "1"
        env.setdefault("BENCH_BACKEND", "gloo")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dist_setup(n_gpus):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != n_gpus:
        # a SCALE line must be what it says: refuse rather than print a 1-rank number labelled N
        print("bench.py: --gpus {} but the process group has WORLD_SIZE {}: refusing to run".format(n_gpus, world),
              file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # BENCH_SAME_DEVICE=1 (set by self_launch on a box with fewer GPUs than ranks): dry run on device 0, gloo
        if os.environ.get("BENCH_SAME_DEVICE"):
            local = 0
            os.environ["BENCH_LAUNCH_LOCAL_RANK"] = os.environ.get("LOCAL_RANK", "0")
            os.environ["LOCAL_RANK"] = "0"      # the package's process-wide context follows LOCAL_RANK
            os.environ.setdefault("BENCH_BACKEND", "gloo")
        elif local >= visible_devices():
            print("bench.py: rank {} has no device {} ({} visible)".format(rank, local, visible_devices()), file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local)
        backend = os.environ.get("BENCH_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        local = 0
    return rank, world, local


def ranks_record(rank, world, local):
    """Proof of N ranks for a SCALE record: the size and backend of the process group the timed region ran on and, per
    rank, the device it computed on (index, name, PCI bus id / uuid where torch exposes them) -- gathered, so a run whose
    ranks all sat on one GPU (BENCH_SAME_DEVICE dry runs) is visible as such."""
    import torch
    prop = torch.cuda.get_device_properties(local)
    mine = dict(rank=rank, local_rank=int(os.environ.get("BENCH_LAUNCH_LOCAL_RANK", os.environ.get("LOCAL_RANK", "0"))), device_index=local, device_name=prop.name,
                pci_bus_id=getattr(prop, "pci_bus_id", None), uuid=str(getattr(prop, "uuid", "")) or None,
                pid=os.getpid())
    if world == 1:
        return dict(ranks_seen=1, backend=None, devices=[mine], distinct_devices=1)
    import torch.distributed as dist
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    distinct = len({(d["device_index"], d["pci_bus_id"], d["uuid"]) for d in everyone})
    rec = dict(ranks_seen=dist.get_world_size(), backend=dist.get_backend(), devices=everyone, distinct_devices=distinct)
    if distinct < world:
        rec["dry_run_same_device"] = True
    return rec


def barrier(world):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x, world):
    import torch
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def seed_states(global_ids, base_seed=0):
    """numpy PCG64 state records for roots with the given global ids (root i <- SeedSequence(base_seed + i))."""
    from rl_agents_amd import native
    ids = np.asarray(global_ids, dtype=np.int64)
    if len(ids) and np.array_equal(ids, ids[0] + np.arange(len(ids))):     # contiguous ids: one C call (numpy-exact)
        return native.seed_sequence_states((), base_seed + int(ids[0]), len(ids))
    out = np.zeros((len(ids), 6), dtype=np.uint64)
    for j, i in enumerate(ids):
        out[j] = native.seed_sequence_states((), base_seed + int(i), 1)[0]
    return out


# ---------------------------------------------------------------------------------------------
PARITY_ROOTS = 512       # roots / planners of a timed launch replayed through the CPU oracle (each replay stays under ~1 s)
PARITY_DENSE_ROWS = 64   # dense VI: sampled (s, a) rows per sweep (a row is |S| doubles: the sample is copied to the host)


def sample_rows(n, k=PARITY_ROOTS):
    """k indices spread over a batch of n (first and last wavefront included)."""
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def parity_record(ok, what, **extra):
    return dict(extra, result="ok" if ok else "MISMATCH", sample=what)


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
                      kernel="uct_kernel<5, {}>".format("ENV_TABLE, per-state policies" if with_prior else
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
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
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
        need = 1.0 if variant in (1, 2) else 0.98
        res["parity_sample"] = parity_record(bool(same.mean() >= need), "{} roots of a {}-root launch vs oracle.uct_plan_batch "
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
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
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
        res["roofline"].update(traffic=None, traffic_frac=None, frac=res["roofline"]["achieved"] / HBM_PEAK_GBS)
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


def _episode_tables(n, shape, seed0=0, distinct=None):
    """n highway-shaped tables of one (V, L, T) grid -- the finite MDPs of n episodes -- of which `distinct` are generated (the
    rest repeat them: every episode still owns its copy on the device)."""
    from rl_agents_amd.envs import generators
    distinct = n if distinct is None else min(n, distinct)
    cfgs = [generators.highway_shaped(*shape, collision_rate=0.03 + 0.02 * (i % 5), seed=seed0 + i) for i in range(distinct)]
    idx = np.arange(n) % distinct
    return (np.stack([c["transition"] for c in cfgs])[idx], np.stack([c["reward"] for c in cfgs])[idx],
            np.stack([c["terminal"] for c in cfgs])[idx])


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
    if "reg" in variant:
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
                           "resident) after one lane-major re-layout pass.  SURVEY 8(d)'s per-sweep formula (12 S A + 17 S) x the "
                           "sweeps really run is beside it (`survey_formula_*`): for the register form that rate exceeds the HBM peak "
                           "because those bytes never leave the CU -- it is not HBM traffic"),
    )
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
    block = 1024 if "wg" in variant else int(variant.split(",")[-1].rstrip(">"))
    kernel_name = "vi_det_batch_reg" if "reg" in variant else ("vi_det_batch_wgr" if "stream" in variant else "vi_det_batch_wg<")
    add_traffic(res["roofline"], "vi_batch", kernel_name, n * block, pattern="stream")
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
                      kernel="uct_kernel<5, ENV_TABLE> on the union model ({})".format(variant), kernel_ms=k_ms,
                      algorithmic_bytes_per_launch=bytes_per_step * env_steps,
                      note="algorithmic bytes = SURVEY 8(d) terms with depth / expansions measured on this launch's trees; every root "
                           "gathers the 16-byte records of ITS OWN {} B table".format(s_ * a_ * 16)),
    )
    res["roofline"]["frac"] = res["roofline"]["achieved"] / HBM_PEAK_GBS
    add_traffic(res["roofline"], "uct_per_root_model", "uct_kernel", n_roots)
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


def run_workload(args, rank, world, local):
    if args.workload == "vi_batch":
        return bench_vi_batch(args, rank, world, local)
    if args.workload == "uct_per_root_model":
        return bench_uct_per_root_model(args, rank, world, local)
    if args.workload == "uct_prior":
        return bench_uct(args, rank, world, local, with_prior=True)
    if args.workload == "uct":
        return bench_uct(args, rank, world, local)
    if args.workload == "uct_cartpole":
        return bench_uct_cartpole(args, rank, world, local)
    if args.workload == "uct_stoch":
        return bench_uct_stoch(args, rank, world, local)
    if args.workload == "opd":
        return bench_opd(args, rank, world, local)
    if args.workload == "ropd":
        return bench_ropd(args, rank, world, local)
    if args.workload == "saopd":
        return bench_saopd(args, rank, world, local)
    if args.workload == "rvi_dense_shard":
        return bench_rvi_dense_shard(args, rank, world, local)
    return bench_vi(args, rank, world, local, dense=args.workload in ("vi_dense", "vi_dense_exact"), robust=args.workload == "rvi",
                    exact=args.workload == "vi_dense_exact")


# the other workloads of the path, each run for a bounded slice after the headline in the default run: (name, workload,
# steps, roots override).  Sized so that a slice (setup, warm-up, timed steps, oracle replay of a sample) stays in seconds.
SLICES = [("uct_prior", "uct_prior", 5, None), ("uct_cartpole", "uct_cartpole", 10, None), ("uct_stoch", "uct_stoch", 5, None),
          ("opd", "opd", 10, None), ("opd8192", "opd", 3, 8192), ("ropd", "ropd", 10, None), ("saopd", "saopd", 3, None),
          ("vi", "vi", 10, None), ("rvi", "rvi", 10, None), ("vi_dense", "vi_dense", 3, None),
          ("vi_dense_exact", "vi_dense_exact", 3, None), ("rvi_dense_shard", "rvi_dense_shard", 10, None),
          ("rvi_dense_shard_exact", "rvi_dense_shard", 10, None, "exact"),
          # round 5: one finite MDP per episode -- N value-iteration agents in one launch, UCT with one MDP per root
          ("vi_batch", "vi_batch", 10, 4096, None, 120), ("vi_batch_s10000", "vi_batch", 5, 64, None, 10000),
          ("uct_per_root_model", "uct_per_root_model", 10, 4096)]


SLICES_MULTI_GPU = [("opd", "opd", 10, None), ("rvi_dense_shard_exact", "rvi_dense_shard", 10, None, "exact")]


def run_slices(args, rank, world, local, slices=None):
    """Every other workload for a bounded slice (no CPU baseline) -> {name: {value, unit, ms_per_step, kernel, kernel_ms,
    frac, traffic_frac, parity_sample, workload}}: the numbers of README / DESIGN, driver-timed in the one line."""
    import copy
    import gc
    import torch
    out = {}
    for name, workload, steps, roots, *mode in (slices or SLICES):
        sub = copy.copy(args)
        # (every slice carries its own bounded CPU baseline -- the C port on this host's cores for about a second -- and the
        # unmodified Python reference's committed timing of the same workload beside it: VERDICT r4)
        sub.workload, sub.steps, sub.warmup, sub.roots = workload, steps, 1, roots
        sub.no_cpu_baseline, sub.cpu_seconds = args.no_cpu_baseline or world > 1, 1.0
        sub.dense_mode = mode[0] if mode and mode[0] else ("mfma" if workload == "rvi_dense_shard" else None)
        if len(mode) > 1:
            sub.states = mode[1]
        t0 = time.perf_counter()
        try:
            res = run_workload(sub, rank, world, local)
            roof = res.get("roofline", {})
            out[name] = dict(workload=res["config"]["workload"], value=res["value"], unit=res["unit"],
                             ms_per_step=res["ms_per_step"], steps=steps, kernel=roof.get("kernel"), kernel_ms=roof.get("kernel_ms"),
                             frac=roof.get("frac"), traffic_frac=roof.get("traffic_frac"),
                             parity_sample=(res.get("parity_sample") or {}).get("result"),
                             parity_detail=(res.get("parity_sample") or {}).get("sample"))
            for k in ("mfma_frac_of_f64_peak",):
                if k in roof:
                    out[name][k] = roof[k]
            extra = res["config"].get("kernel_ms_first_and_following_plans")
            if extra is not None:
                out[name]["kernel_ms_first_and_following_plans"] = extra
            for k in ("exchange", "cpu_baseline", "variant", "speedup_vs_single_solve", "vs_shared_model_kernel"):
                if res.get(k) is not None:
                    out[name][k] = res[k]
            if isinstance(out[name].get("cpu_baseline"), dict):
                ref = reference_python(workload)
                if ref is not None:
                    out[name]["cpu_baseline"]["reference_python"] = ref
            if world > 1:
                rec = ranks_record(rank, world, local)
                out[name]["ranks_seen"] = rec["ranks_seen"]
                out[name]["cross_check"] = (res.get("_cross") or {}).get("cross_check")
                out[name]["parallelism"] = res["config"].get("parallelism")
                if workload == "rvi_dense_shard":
                    out[name]["all_gather_ms"] = res["config"].get("all_gather_ms")
        except Exception as e:                                    # a slice must not cost the headline its line
            out[name] = dict(error="{}: {}".format(type(e).__name__, e))
        out[name]["slice_seconds"] = round(time.perf_counter() - t0, 2)
        del sub
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def main():
    if os.environ.get("BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG"]), exit=True)
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                # plain `python bench.py --gpus N`: launch the N ranks ourselves
    rank, world, local = dist_setup(args.gpus)
    import torch
    side = torch.cuda.Stream(device=local)          # one stream for torch ops, RCCL and the HIP kernels
    with torch.cuda.stream(side):
        res = run_workload(args, rank, world, local)
        res.update(n_gpus=world, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                   vs_baseline=None, data="synthetic (highway-shaped finite MDP; real highway_env absent)")
        res["ranks"] = ranks_record(rank, world, local)
        res["ranks"].update(res.pop("_cross", None) or {})
        if res["ranks"].get("dry_run_same_device"):
            res["scaling"] = "weak (DRY RUN: all ranks on one device, not a scaling measurement)"
        if args.workload == "uct" and world == 1 and not args.headline_only:
            res["workloads"] = run_slices(args, rank, world, local)
        elif args.workload == "uct" and world > 1 and not args.headline_only:
            # the other two BASELINE configurations that are sharded over the GPUs of a node, in the same launch the driver
            # times: C4 (OPD, 1024 roots per GPU) and C5 (dense robust VI, one row block per GPU, all_gather of V per sweep)
            res["workloads"] = run_slices(args, rank, world, local, slices=SLICES_MULTI_GPU)
    res.setdefault("cpu_baseline", None)
    if isinstance(res["cpu_baseline"], dict):
        # the reference's own (pure Python) CPU path on the same tables: it cannot travel to the GPU box, so its timing is
        # a committed, script-generated record of the build container (tests/golden/gen/time_reference.py), printed
        # beside the C port that is timed live on this host
        ref = reference_python(args.workload)
        if ref is not None:
            res["cpu_baseline"]["reference_python"] = ref
    if rank == 0:
        order = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]
        order += [k for k in res if k not in order]
        print(json.dumps({k: res[k] for k in order}))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
