#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py rl_agents_amd/csrc/vi.hip [name-filter]
"""
import re
import subprocess
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from rl_agents_amd import build  # noqa: E402


def main():
    src, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    cmd = [build.hipcc()] + build.FLAGS + ["-I", build.INCLUDE, "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL).stderr.decode()
    cur, rows = None, []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = dict(name=subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip())
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([^:]+?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("{:<90} {:>5} {:>5} {:>7} {:>7} {:>4} {:>7}".format("kernel", "VGPR", "AGPR", "spillV", "scratch", "occ", "LDS"))
    for r in rows:
        if flt in r["name"]:
            print("{:<90} {:>5} {:>5} {:>7} {:>7} {:>4} {:>7}".format(
                r["name"][:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("VGPRs Spill", -1),
                r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))


if __name__ == "__main__":
    main()
