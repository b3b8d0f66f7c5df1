#!/usr/bin/env python3
"""Addendum to profiles/r04_kernel_stats.md and profiles/r04_pmc.json for the dense VI kernel in numpy's order of additions
(vi_dense_exact_q), from the rocprofv3 output of tools/gpu_r04_full2.sh (gpurun_out/r04y/, scratch):

    python tools/summarize_dense_exact.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_profiles import REPO, kernel_stats, pmc, trace_by_grid  # noqa: E402

SRC = os.path.join(REPO, "gpurun_out", "r04y")
MARK = "# Addendum: dense VI in numpy's order of additions (vi_dense_exact_q)"


def main():
    stats_path = os.path.join(REPO, "profiles", "r04_kernel_stats.md")
    text = open(stats_path).read()
    if MARK in text:
        text = text[:text.index(MARK)].rstrip() + "\n"
    lines = ["", MARK, "",
             "Commands: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload vi_dense_exact --steps 5 "
             "--warmup 1 --no-cpu-baseline --headline-only --no-parity-sample` and `... --workload rvi_dense_shard --dense-mode exact ...` "
             "(tools/gpu_r04_full2.sh), one MI355X.  HBM counters: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same "
             "commands with --steps 3.", ""]
    traffic = json.load(open(os.path.join(REPO, "profiles", "r04_pmc.json")))
    for wl, key in (("vi_dense_exact", "vi_dense_exact"), ("rvi_dense_shard", "rvi_dense_shard_exact")):
        d = os.path.join(SRC, "trace_{}_exact".format(wl))
        lines += ["## {} ({})".format(key, "S = 10 000, 4.0 GB per sweep" if wl == "vi_dense_exact" else "one C5 rank's 25 GB row block, M = 2"), "",
                  "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
        for name, calls, avg, pct in kernel_stats(os.path.join(d, wl + "_kernel_stats.csv"), top=5):
            lines.append("| `{}` | {} | {:.2f} | {:.2f} |".format(name.replace("|", "/"), calls, avg, pct))
        lines += ["", "| kernel | grid (threads) | block | VGPRs | LDS B | calls | avg us | min us | max us |", "|---|---|---|---|---|---|---|---|---|"]
        for (name, grid, wg, vgpr, ldsb), (n, mean, lo, hi) in sorted(trace_by_grid(os.path.join(d, wl + "_kernel_trace.csv")).items(),
                                                                         key=lambda kv: -kv[1][1])[:3]:
            lines.append("| `{}` | {} | {} | {} | {} | {} | {:.2f} | {:.2f} | {:.2f} |".format(name, grid, wg, vgpr, ldsb, n, mean, lo, hi))
        entry = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(SRC, "pmc_{}_exact_{}".format(wl, ctr), wl + "_counter_collection.csv")
            for k, (n, mean_kb) in pmc(f, ctr).items():
                if "exact" in k:
                    entry.setdefault(k, {})[ctr + "_KB_per_launch"] = mean_kb
                    entry[k]["launches_" + ctr] = n
        traffic[key] = entry
        lines += ["", "| kernel | FETCH_SIZE KB per launch (x2 = bytes) | WRITE_SIZE KB per launch |", "|---|---|---|"]
        for k, v in entry.items():
            lines.append("| `{}` | {:.1f} | {:.1f} |".format(k[:80], v["FETCH_SIZE_KB_per_launch"], v["WRITE_SIZE_KB_per_launch"]))
        lines.append("")
    with open(stats_path, "w") as f:
        f.write(text + "\n".join(lines))
    with open(os.path.join(REPO, "profiles", "r04_pmc.json"), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
