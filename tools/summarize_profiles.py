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
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter and "mp::" in row["Kernel_Name"]:
            agg[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
    dst = os.path.join(REPO, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = ["# rocprofv3 --kernel-trace --stats summaries ({})".format(tag), "",
             "Command per workload: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
             "--workload <wl> --steps 5 --warmup 1 --no-cpu-baseline` on one MI355X (tools/profile_gpu.sh).", ""]
    for wl in ("uct", "opd", "vi", "vi_dense"):
        f = os.path.join(src, "trace_" + wl, wl + "_kernel_stats.csv")
        if not os.path.exists(f):
            continue
        lines += ["## " + wl, "", "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
        for name, calls, avg, pct in kernel_stats(f):
            lines.append("| `{}` | {} | {:.2f} | {:.2f} |".format(name.replace("|", "/"), calls, avg, pct))
        lines.append("")
    traffic = {}
    for wl in ("uct", "vi_dense", "opd"):
        entry = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(src, "pmc_{}_{}".format(wl, ctr), wl + "_counter_collection.csv")
            if os.path.exists(f):
                for k, (n, mean_kb) in pmc(f, ctr).items():
                    entry.setdefault(k, {})[ctr + "_KB_per_launch"] = mean_kb
                    entry[k]["launches_" + ctr] = n
        if entry:
            traffic[wl] = entry
    if traffic:
        lines += ["## HBM traffic (PMC, separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`)", "",
                  "Per launch, KB as rocprofv3 reports them. gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE "
                  "counts half the bytes of a wide (16 B/lane) coalesced stream -- vi_dense_q streams 4.0 GB per "
                  "sweep with dwordx4 loads and reads 1.95 GB raw, i.e. exactly that factor 2 (calibration on a known "
                  "byte count); narrow random gathers (uct) are uncalibrated, raw values are shown.", "",
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
