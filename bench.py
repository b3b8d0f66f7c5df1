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

This file is the driver (arguments, rank launch, the one JSON line); the workloads live in benchmarks/{uct,opd,vi}.py over
benchmarks/common.py (peaks, PMC-traffic lookup, rank helpers, oracle-replay sampling).
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

from benchmarks.common import ranks_record, reference_python, visible_devices     # noqa: E402
from benchmarks.uct import bench_uct, bench_uct_cartpole, bench_uct_stoch, bench_uct_per_root_model     # noqa: E402
from benchmarks.opd import bench_opd, bench_ropd, bench_saopd     # noqa: E402
from benchmarks.vi import bench_vi, bench_rvi_dense_shard, bench_vi_batch     # noqa: E402
from benchmarks.eval import bench_per_episode_eval     # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="uct", choices=["uct", "uct_prior", "uct_cartpole", "uct_stoch", "opd", "ropd", "saopd", "vi", "rvi", "vi_dense", "vi_dense_exact", "rvi_dense_shard", "vi_batch",
                                                         "uct_per_root_model", "per_episode_eval"])
    ap.add_argument("--roots", type=int, default=None, help="roots per GPU (default 262144 uct, 1024 opd)")
    ap.add_argument("--states", type=int, default=None, help="|S| override (vi_dense default 10000)")
    ap.add_argument("--dense-mode", default=None, choices=["mfma", "exact"],
                    help="dense VI workloads: contraction on the f64 matrix cores or in numpy's order of additions (bit-exact)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--headline-only", action="store_true", help="skip the bounded slices of the other workloads")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip the oracle replay of a sample of the timed launch")
    return ap.parse_args()


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


def run_workload(args, rank, world, local):
    if args.workload == "vi_batch":
        return bench_vi_batch(args, rank, world, local)
    if args.workload == "uct_per_root_model":
        return bench_uct_per_root_model(args, rank, world, local)
    if args.workload == "per_episode_eval":
        return bench_per_episode_eval(args, rank, world, local)
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
          ("uct_per_root_model", "uct_per_root_model", 10, 4096),
          # round 6: the per-episode evaluation LOOP end to end (extraction, upload, plan, env.step), 4096 episodes x 2 lock-steps
          ("per_episode_eval", "per_episode_eval", 2, 4096)]


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
        # the long form first (one JSON object per workload, also written to bench_detail.json), then the ONE compact line the
        # driver parses: <= 4 KB, strict JSON (benchmarks/report.py; tests/test_bench_line.py)
        from benchmarks.report import compact_line, detail_lines
        detail = detail_lines(res)
        for ln in detail:
            print(ln)
        try:
            side = os.environ.get("BENCH_DETAIL_FILE") or os.path.join(REPO, "gpurun_out", "bench_detail.json")
            os.makedirs(os.path.dirname(side), exist_ok=True)
            with open(side, "w") as f:
                f.write("\n".join(detail) + "\n")
        except OSError:
            pass
        sys.stdout.flush()
        print(compact_line(res))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
