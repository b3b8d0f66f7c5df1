"""Build libmi355plan.so (HIP, gfx950 only) in-tree with hipcc.

    python -m rl_agents_amd.build [--force]

The library is the product's only compute path; there is no fallback.  It is built in-tree
(rl_agents_amd/lib/) so that it travels with the source snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmi355plan.so")
SOURCES = ["api.hip", "vi.hip", "uct.hip", "uct_stoch.hip", "opd.hip", "ropd.hip", "saopd.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".hpp"))   # every header: a stale library is a wrong library
# -ffp-contract=off: the reference evaluates a*b+c with two roundings (Python floats); a fused
# multiply-add would change the last bit of bounds and Q values and break bit-exact parity.
# MP_PROFILE=1 in the environment builds the phase-instrumented kernels (device printf of clock64 ticks)
FLAGS = (["-DMP_PROFILE"] if os.environ.get("MP_PROFILE") else []) + \
        [f for f in os.environ.get("MP_EXTRA_FLAGS", "").split() if f] + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value", "-Wno-pass-failed"]


# Per-file code-generation flags.  The UCT kernels are long straight-line blocks run by few wavefronts per SIMD (a lone wave issues
# in order: what is not interleaved in the instruction stream is not overlapped at all), so uct.hip is scheduled for instruction-
# level parallelism instead of occupancy (LLVM's AMDGPU `max-ilp` strategy).  A/B on one box (tools/ab_libs.sh, two rounds each):
# headline kernel 0.783 -> 0.767 ms, 4096 roots 0.275 -> 0.268, single root 0.0575 -> 0.056; uct_stoch no consistent change (its
# time is bimodal, 2.38 / 2.65 ms, under either strategy); opd 0.756 -> 0.781 and saopd 8.8 -> 9.25 SLOWER: those keep the default.
FILE_FLAGS = {"uct.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}
if os.environ.get("MP_NO_FILE_FLAGS"):
    FILE_FLAGS = {}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 to build libmi355plan.so)")


STAMP_PATH = LIB_PATH + ".stamp"


def source_digest():
    """sha256 over every source, header and flag that goes into the library (mtimes do not survive the
    snapshot to the GPU box, content does)."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(FILE_FLAGS.items()))).encode())
    for d in [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(INCLUDE, "mi355plan.h")]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return True
    with open(STAMP_PATH) as f:
        return f.read().strip() != source_digest()


def check_generated_code():
    """The hand-counted `s_waitcnt vmcnt(N)` before opd.hip's scalar leaf-record loads against the code hipcc emitted
    (tools/check_isa.py): a library whose count is off is refused, not shipped."""
    if "-DMP_OPD_SAFE_WAITCNT" in FLAGS:
        return 0
    import importlib.util
    path = os.path.join(os.path.dirname(HERE), "tools", "check_isa.py")
    spec = importlib.util.spec_from_file_location("check_isa", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.check(os.path.join(LIB_DIR, "opd.o"), min_sites=12)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the shared library. Returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cc = hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        cmd = [cc] + FLAGS + FILE_FLAGS.get(src, []) + ["-I", INCLUDE, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on {}:\n{}".format(src, out.decode()))
        if verbose and out:
            print(out.decode())
    check_generated_code()
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:
        f.write(source_digest())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
