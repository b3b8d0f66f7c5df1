#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of build_variants/gather_calib -> profiles/<tag>_gather_calib.{json,md}:
what the two counters tally per request of each access pattern of the tree-search kernels (tools/gather_calib.hip).

    python tools/summarize_calib.py gpurun_out/calib_r02 r02 [log2_lines=25]
"""
import collections
import csv
import glob
import json
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USEFUL = {"cal_stream_rd16": 16, "cal_stream_wr16": 16, "cal_gather16": 16, "cal_gather16_pair": 32, "cal_gather16_x4": 64,
          "cal_rmw16": 32, "cal_scatter16": 16, "cal_scatter12": 12}


def counter_means(root, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.match(r"(cal_\w+)", row["Kernel_Name"])
            if m and row["Counter_Name"] == counter:
                agg[m.group(1)].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    lg = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    n_lines = 1 << lg
    fetch = counter_means(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "write"), "WRITE_SIZE")
    timing = {}
    log = os.path.join(src, "plain.log")
    if os.path.exists(log):
        for line in open(log):
            m = re.match(r"(cal_\w+)\s+requests (\d+)\s+useful B/request (\d+)\s+([\d.]+) ms\s+([\d.]+) G requests/s", line)
            if m:
                timing[m.group(1)] = dict(ms=float(m.group(4)), g_requests_per_s=float(m.group(5)))
    out = {}
    for k, useful in USEFUL.items():
        reqs = n_lines * 8 if k.startswith("cal_stream") else n_lines
        e = dict(requests=reqs, useful_bytes_per_request=useful)
        if k in fetch:
            e["FETCH_SIZE_KB"] = fetch[k]
            e["fetch_bytes_tallied_per_request"] = fetch[k] * 1024.0 / reqs
        if k in write:
            e["WRITE_SIZE_KB"] = write[k]
            e["write_bytes_tallied_per_request"] = write[k] * 1024.0 / reqs
        e.update(timing.get(k, {}))
        out[k] = e
    # correction factors = true fabric bytes per byte the counter tallies, per access pattern (bench.py calibration()):
    #  * streams: every byte is used, so true = useful;
    #  * scattered 16-B loads: a miss is ONE request whether one or both 64-B halves of the 128-B line are read (the
    #    pair kernel tallies what the single kernel does), i.e. the fabric moves the whole line: true = 128 B;
    #  * scattered 16-B / 12-B stores: tallied in 32-B units with NO fill read (FETCH_SIZE = 0): the fabric's partial
    #    write granule, taken as the true traffic.
    g, gp = out["cal_gather16"], out["cal_gather16_pair"]
    factors, notes = {}, []
    if "fetch_bytes_tallied_per_request" in out["cal_stream_rd16"]:
        factors["fetch_stream"] = 16.0 / out["cal_stream_rd16"]["fetch_bytes_tallied_per_request"]
    if "write_bytes_tallied_per_request" in out["cal_stream_wr16"]:
        factors["write_stream"] = 16.0 / out["cal_stream_wr16"]["write_bytes_tallied_per_request"]
    if "fetch_bytes_tallied_per_request" in g and "fetch_bytes_tallied_per_request" in gp:
        whole_line = gp["fetch_bytes_tallied_per_request"] < 1.25 * g["fetch_bytes_tallied_per_request"]
        true_bytes = 128.0 if whole_line else 64.0
        factors["fetch_scattered"] = true_bytes / g["fetch_bytes_tallied_per_request"]
        notes.append("a scattered 16-B load miss moves {:.0f} B over the fabric (two loads in different 64-B halves of one "
                     "line tally {:.1f} B vs {:.1f} B for one load) and is tallied at {:.1f} B".format(
                         true_bytes, gp["fetch_bytes_tallied_per_request"], g["fetch_bytes_tallied_per_request"],
                         g["fetch_bytes_tallied_per_request"]))
    if "write_bytes_tallied_per_request" in out["cal_scatter16"]:
        factors["write_scattered"] = 1.0
        notes.append("a scattered 16-B (or 8+4-B) store is tallied at {:.1f} B by WRITE_SIZE and causes no fill read "
                     "(FETCH_SIZE {:.1f} B per request)".format(out["cal_scatter16"]["write_bytes_tallied_per_request"],
                                                               out["cal_scatter16"].get("fetch_bytes_tallied_per_request", 0.0)))
    os.makedirs(os.path.join(REPO, "profiles"), exist_ok=True)
    with open(os.path.join(REPO, "profiles", tag + "_gather_calib.json"), "w") as f:
        json.dump(dict(log2_lines=lg, table_bytes=n_lines * 128, kernels=out, factors=factors, notes=notes), f, indent=1,
                  sort_keys=True)
    lines = ["# FETCH_SIZE / WRITE_SIZE calibration on the tree-search access patterns ({})".format(tag), "",
             "`tools/gather_calib.hip`, table of 2^{} lines x 128 B = {:.1f} GiB (>> L2 + Infinity Cache), one request per "
             "line and launch; separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes.".format(
                 lg, n_lines * 128 / 2 ** 30), "",
             "| kernel | requests | useful B/req | FETCH_SIZE B/req | WRITE_SIZE B/req | ms | G req/s |", "|---|---|---|---|---|---|---|"]
    for k, e in out.items():
        lines.append("| `{}` | {} | {} | {} | {} | {} | {} |".format(
            k, e["requests"], e["useful_bytes_per_request"],
            "{:.1f}".format(e["fetch_bytes_tallied_per_request"]) if "fetch_bytes_tallied_per_request" in e else "-",
            "{:.1f}".format(e["write_bytes_tallied_per_request"]) if "write_bytes_tallied_per_request" in e else "-",
            "{:.3f}".format(e["ms"]) if "ms" in e else "-", "{:.2f}".format(e["g_requests_per_s"]) if "g_requests_per_s" in e else "-"))
    lines += ["", "Correction factors (true fabric bytes per tallied byte) used by `bench.py`: " +
              ", ".join("{} = {:.2f}".format(k, v) for k, v in sorted(factors.items())) + "."] + ["* " + n for n in notes]
    lines += ["* timing cross-check: scattered loads run at the rate in the table; x 128 B that is the HBM bandwidth they "
              "draw -- a scattered 16-byte gather costs a full line of DRAM traffic."]
    with open(os.path.join(REPO, "profiles", tag + "_gather_calib.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
