#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/prof_<tag>/, scratch) into the tracked
summaries under profiles/: <tag>_kernel_stats.md (per-kernel time, `--kernel-trace --stats`) and
<tag>_pmc.json (HBM traffic per launch from the FETCH_SIZE / WRITE_SIZE passes).

    python tools/summarize_profiles.py r01
"""
import collections
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(path, top=8):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows[:top]:
        out.append((r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def pmc(path, counter):
    """mean counter value per (kernel, grid size): one bench run launches the same kernel on several batch sizes"""
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and "mp::" in row["Kernel_Name"]:
            agg["{} grid={}".format(row["Kernel_Name"], row["Grid_Size"])].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def trace_by_grid(path):
    """(kernel, grid) -> (calls, mean us, min us, max us) for this library's kernels, from the raw kernel trace"""
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if "mp::" in row["Kernel_Name"]:
            key = (row["Kernel_Name"][:70], int(row["Grid_Size_X"]), int(row["Workgroup_Size_X"]), int(row["VGPR_Count"]),
                   int(row["LDS_Block_Size"]))
            agg[key].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return {k: (len(v), sum(v) / len(v), min(v), max(v)) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = ["# rocprofv3 --kernel-trace --stats summaries ({})".format(tag), "",
             "Command per workload: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
             "--workload <wl> --steps 5 --warmup 1 --no-cpu-baseline --headline-only --no-parity-sample` on one MI355X (tools/profile_gpu.sh; BENCH_NO_LIVE_PMC unset is harmless: --headline-only skips it).", ""]
    for wl in ("uct", "uct4096", "uct1024", "uct256", "uct1", "uct_per_root_model", "uct_prior", "uct_cartpole", "uct_stoch", "opd", "opd8192", "ropd", "saopd", "vi", "rvi",
               "vi_batch", "vi_batch_s10000", "vi_batch_s10000_256", "vi_dense", "vi_dense_exact", "rvi_dense_shard", "rvi_dense_shard_exact"):
        f = os.path.join(src, "trace_" + wl, wl + "_kernel_stats.csv")
        if not os.path.exists(f):
            continue
        lines += ["## " + wl + (" (= --workload opd --roots 8192)" if wl == "opd8192" else
                                " (= --workload uct --roots 4096: the row kernel on a shared model, uct_row_kernel<5, true>)" if wl == "uct4096" else
                                " (= --workload uct --roots 1024: four roots per workgroup, a wavefront each, uct_lone_kernel<5, false, true>)" if wl == "uct1024" else
                                " (= --workload uct --roots 256: one root per workgroup, uct_lone_kernel)" if wl == "uct256" else
                                " (= --workload uct --roots 1: a single agent's plan, uct_lone_kernel)" if wl == "uct1" else
                                " (= --workload vi_batch --states 120 --roots 4096)" if wl == "vi_batch" else
                                " (= --workload vi_batch --states 10000 --roots 64)" if wl == "vi_batch_s10000" else
                                " (= --workload vi_batch --states 10000 --roots 256)" if wl == "vi_batch_s10000_256" else
                                " (= --workload rvi_dense_shard --dense-mode exact)" if wl == "rvi_dense_shard_exact" else
                                " (--dense-mode mfma)" if wl == "rvi_dense_shard" else ""), "",
                  "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
        for name, calls, avg, pct in kernel_stats(f):
            lines.append("| `{}` | {} | {:.2f} | {:.2f} |".format(name.replace("|", "/"), calls, avg, pct))
        lines.append("")
        tr = os.path.join(src, "trace_" + wl, wl + "_kernel_trace.csv")
        if os.path.exists(tr):
            lines += ["Per launch geometry (same kernel, different batch sizes are separate rows):", "",
                      "| kernel | grid (threads) | block | VGPRs | LDS B | calls | avg us | min us | max us |",
                      "|---|---|---|---|---|---|---|---|---|"]
            for (name, grid, wg, vgpr, ldsb), (n, mean, lo, hi) in sorted(trace_by_grid(tr).items(), key=lambda kv: -kv[1][1]):
                lines.append("| `{}` | {} | {} | {} | {} | {} | {:.2f} | {:.2f} | {:.2f} |".format(
                    name, grid, wg, vgpr, ldsb, n, mean, lo, hi))
            lines.append("")
    traffic = {}
    for wl in ("uct", "uct4096", "uct_per_root_model", "uct_prior", "uct_cartpole", "uct_stoch", "vi", "rvi", "vi_batch", "vi_batch_s10000", "vi_batch_s10000_256", "vi_dense", "vi_dense_exact",
               "rvi_dense_shard", "rvi_dense_shard_exact", "opd", "opd8192", "ropd", "saopd"):
        entry = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(src, "pmc_{}_{}".format(wl, ctr), wl + "_counter_collection.csv")
            if os.path.exists(f):
                for k, (n, mean_kb) in pmc(f, ctr).items():
                    entry.setdefault(k, {})[ctr + "_KB_per_launch"] = mean_kb
                    entry[k]["launches_" + ctr] = n
        if entry:  # (bench.py looks a launch up by workload and grid size: the 8192-root pass belongs to "opd")
            traffic.setdefault("opd" if wl == "opd8192" else ("vi_batch" if wl.startswith("vi_batch") else ("uct" if wl == "uct4096" else wl)), {}).update(entry)
    if traffic:
        lines += ["## HBM traffic (PMC, separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`)", "",
                  "Per launch, KB as rocprofv3 reports them.  To bytes: x2 for FETCH_SIZE (streams AND scattered 16-byte gathers: "
                  "every 128-byte line request is tallied at 64 B), x1 for WRITE_SIZE -- calibrated on known request counts in "
                  "these kernels' own access patterns, profiles/r02_gather_calib.md.", "",
                  "| workload | kernel | FETCH_SIZE KB | WRITE_SIZE KB |", "|---|---|---|---|"]
        for wl, entry in traffic.items():
            for k, v in entry.items():
                lines.append("| {} | `{}` | {:.1f} | {:.1f} |".format(wl, k[:70], v.get("FETCH_SIZE_KB_per_launch", float("nan")),
                                                                    v.get("WRITE_SIZE_KB_per_launch", float("nan"))))
        lines.append("")
        with open(os.path.join(dst, tag + "_pmc.json"), "w") as f:
            json.dump(traffic, f, indent=1, sort_keys=True)
    with open(os.path.join(dst, tag + "_kernel_stats.md"), "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
