#!/usr/bin/env python3
"""gpurun_out/uct_layout (tools/profile_uct_layout.sh) -> profiles/<tag>_uct_tree_layout.md: the UCT tree-layout A/B."""
import collections
import csv
import glob
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(REPO, "gpurun_out", "uct_layout")
    rows = collections.OrderedDict()
    for lay in ("rootmajor", "interleaved", "group"):
        for n in (262144, 4096):
            key = (lay, n)
            rows[key] = {}
            log = os.path.join(src, "{}_{}_plain.log".format(lay, n))
            if os.path.exists(log):
                ms = [float(m.group(1)) for m in re.finditer(r"kernel ([\d.]+) ms", open(log).read())]
                if ms:
                    rows[key]["kernel_ms"] = min(ms)
            for f in glob.glob(os.path.join(src, "{}_{}_set*".format(lay, n), "**", "*counter_collection.csv"), recursive=True):
                agg = collections.defaultdict(list)
                for r in csv.DictReader(open(f)):
                    if "uct_kernel" in r["Kernel_Name"]:
                        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, v in agg.items():
                    rows[key][k] = sum(v) / len(v)
    names = ["kernel_ms", "GRBM_GUI_ACTIVE", "TA_BUSY_avr", "TA_BUSY_max", "TCP_TOTAL_CACHE_ACCESSES_sum",
             "TCP_PENDING_STALL_CYCLES_sum", "SQ_INSTS_VALU", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "FETCH_SIZE", "WRITE_SIZE"]
    lines = ["# UCT tree layout A/B ({}): root-major Node[root][cap] vs wave-interleaved Node[root/64][cap][64]".format(tag), "",
             "`tools/profile_uct_layout.sh`: `rocprofv3 --kernel-trace --pmc <set> -- python tools/micro_uct_opd.py uct <roots>` with "
             "`MP_UCT_TREE=rootmajor|interleaved` (headline configuration, 33 episodes x horizon 30, S = 10 000, A = 5); counter "
             "values are means over the run's 3 launches, kernel ms the best un-profiled launch. FETCH_SIZE / WRITE_SIZE in KB as "
             "reported (x2 / x1 to bytes, profiles/r02_gather_calib.md).", "",
             "| counter | " + " | ".join("{} {}".format(l, n) for (l, n) in rows) + " |", "|---|" + "---|" * len(rows)]
    for nm in names:
        lines.append("| {} | ".format(nm) + " | ".join("{:.4g}".format(rows[k][nm]) if nm in rows[k] else "-" for k in rows) + " |")
    for n in (262144, 4096):
        a = rows[("rootmajor", n)]
        for other in ("interleaved", "group"):
            b = rows[(other, n)]
            if "TA_BUSY_avr" in a and "GRBM_GUI_ACTIVE" in a and "TA_BUSY_avr" in b:
                lines.append("")
                lines.append("{} roots, {} vs root-major: TA busy {:.0f} % -> {:.0f} % of the kernel's cycles (avr), L1 line accesses x{:.2f}, "
                             "HBM bytes (2 FETCH + WRITE) x{:.2f}, kernel time x{:.3f}.".format(
                    n, other, 100 * a["TA_BUSY_avr"] / (a["GRBM_GUI_ACTIVE"] / 8), 100 * b["TA_BUSY_avr"] / (b["GRBM_GUI_ACTIVE"] / 8),
                    b.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")) / a.get("TCP_TOTAL_CACHE_ACCESSES_sum", float("nan")),
                    (2 * b.get("FETCH_SIZE", float("nan")) + b.get("WRITE_SIZE", float("nan"))) / (2 * a.get("FETCH_SIZE", float("nan")) + a.get("WRITE_SIZE", float("nan"))),
                    b.get("kernel_ms", float("nan")) / a.get("kernel_ms", float("nan"))))
    with open(os.path.join(REPO, "profiles", tag + "_uct_tree_layout.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
