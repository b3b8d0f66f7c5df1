#!/usr/bin/env python3
"""Build-time check of hand-counted `s_waitcnt vmcnt(N)` sites (rl_agents_amd/csrc/opd.hip: OPD_LEAF_SLOAD).

    python tools/check_isa.py rl_agents_amd/lib/opd.o

Every `s_waitcnt vmcnt(N)` that is directly followed by `s_load_dwordx4 ... glc` must be preceded -- walking the listing
backwards -- by exactly N - 1 `global_load_dwordx2` (the row loads of this expansion) and then one `global_store_dwordx2` (the
tag store into the leaf table) as its N nearest vector-memory instructions, all within WINDOW instructions.  Anything else
means the compiler emitted a different number of vector-memory operations than the source counted, and the scalar load could
read a node record whose store is still in flight.  check() returns the number of sites checked and raises on a violation."""
import os
import re
import subprocess
import sys
import tempfile

WINDOW = 120
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)(load|store|atomic)\w*")


def device_listing(obj):
    """Disassembly of the gfx950 code object bundled in a hipcc object file."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        dev = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not dev:
            raise RuntimeError("no device code object in " + obj)
        return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, dev[0])], stdout=subprocess.PIPE, check=True).stdout.decode()


def check(obj, min_sites=1):
    lines = [ln.split("//")[0].strip() for ln in device_listing(obj).splitlines()]
    sites = 0
    for i, ln in enumerate(lines[:-1]):
        m = re.match(r"s_waitcnt vmcnt\((\d+)\)$", ln)
        if not m or not re.match(r"s_load_dwordx4 .* glc$", lines[i + 1]):
            continue
        n = int(m.group(1))
        if n == 0:
            continue                                  # the safe form waits for everything
        sites += 1
        seen = []
        for j in range(i - 1, max(i - 1 - WINDOW, -1), -1):
            v = VMEM.match(lines[j])
            if v:
                seen.append(lines[j].split()[0])
                if len(seen) == n:
                    break
        want = ["global_load_dwordx2"] * (n - 1) + ["global_store_dwordx2"]
        if seen != want:
            raise RuntimeError("{}: the {} vector-memory instructions before `s_waitcnt vmcnt({})` at listing line {} are {} "
                               "(expected {}): the hand count in OPD_LEAF_SLOAD no longer matches the generated code -- build "
                               "with MP_EXTRA_FLAGS=-DMP_OPD_SAFE_WAITCNT or fix the count".format(obj, n, n, i + 1, seen, want))
    if sites < min_sites:
        raise RuntimeError("{}: no hand-counted leaf-record load found (expected at least {})".format(obj, min_sites))
    return sites


if __name__ == "__main__":
    print("{} site(s) ok".format(check(sys.argv[1])))
