"""Shared pieces of the benchmark (bench.py is the driver; one module per planner family beside this one):
peaks, PMC-traffic lookup and correction factors, rank bookkeeping and max/sum over ranks, root seeding, oracle-replay sampling.
"""
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
BENCH_PY = os.path.join(REPO, "bench.py")

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
               sys.executable, BENCH_PY, "--headline-only", "--no-cpu-baseline", "--no-parity-sample", "--steps", "3",
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


def _episode_tables(n, shape, seed0=0, distinct=None):
    """n highway-shaped tables of one (V, L, T) grid -- the finite MDPs of n episodes -- of which `distinct` are generated (the
    rest repeat them: every episode still owns its copy on the device)."""
    from rl_agents_amd.envs import generators
    distinct = n if distinct is None else min(n, distinct)
    cfgs = [generators.highway_shaped(*shape, collision_rate=0.03 + 0.02 * (i % 5), seed=seed0 + i) for i in range(distinct)]
    idx = np.arange(n) % distinct
    return (np.stack([c["transition"] for c in cfgs])[idx], np.stack([c["reward"] for c in cfgs])[idx],
            np.stack([c["terminal"] for c in cfgs])[idx])
