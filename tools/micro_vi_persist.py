"""A/B of the deterministic VI paths on the GPU: chained launches (MP_VI_NO_PERSIST=1) vs the persistent kernel.
    python tools/micro_vi_persist.py   -> us per sweep for C2 (S=10000) and C5-det robust (S=50000, M=2)"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    ctx = native.Context(0)
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    big = generators.highway_shaped(10, 50, 100, seed=2)
    big2 = generators.rewire(big, 0.1, seed=3)
    cases = [("C2 S=10000 A=5", ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"]), False),
             ("C5-det S=50000 A=5 M=2", ctx.load_table(np.stack([big["transition"], big2["transition"]]),
                                                       np.stack([big["reward"], big2["reward"] * 0.97])), True)]
    for name, model, robust in cases:
        for sweeps in (200, 1000):
            ctx.vi_sweeps(model, 0.95, sweeps, robust=robust)
            ctx.synchronize()
            ts, ks = [], []
            for _ in range(10):
                t0 = time.perf_counter()
                ctx.vi_sweeps(model, 0.95, sweeps, robust=robust)
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
                ks.append(ctx.last_kernel_ms()[0])
            print("{:28s} {:5d} sweeps: wall {:8.3f} ms  kernels {:8.3f} ms  -> {:6.3f} us/sweep (kernel), {:6.3f} us/sweep (wall)".format(
                name, sweeps, 1e3 * np.median(ts), np.median(ks), 1e3 * np.median(ks) / sweeps, 1e6 * np.median(ts) / sweeps))
        t0 = time.perf_counter()
        q, n = ctx.vi_solve(model, 0.95, 200, robust=robust)
        print("   solve(200): {} sweeps run, {:.3f} ms wall (host arrays)".format(n, 1e3 * (time.perf_counter() - t0)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        for label, env in (("persistent kernel", {}), ("chained launches (MP_VI_NO_PERSIST=1)", {"MP_VI_NO_PERSIST": "1"})):
            print("== " + label)
            sys.stdout.flush()
            subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env))
