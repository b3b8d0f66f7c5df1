"""Small-MDP VI latency: MI355PLAN_NO_TORCH=1 python tools/micro_vi_small.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402

ctx = native.Context(0)
for s, a in ((100, 5), (1000, 5), (2400, 5)):
    cfg = generators.random_deterministic(s, a, seed=1)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    for rep in range(3):
        t0 = time.perf_counter()
        q, sw = ctx.vi_solve(model, 0.9, 200)
        dt = time.perf_counter() - t0
        ms, n = ctx.last_kernel_ms()
    print("vi S={} A={}: {} sweeps, kernels {:.3f} ms in {} launch(es), wall {:.3f} ms".format(s, a, sw, ms, n, dt * 1e3), flush=True)
