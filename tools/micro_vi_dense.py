"""Dense VI micro-benchmark without torch: MI355PLAN_NO_TORCH=1 python tools/micro_vi_dense.py [S] [A]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
A = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.Generator(np.random.PCG64(0))
t = rng.random((S, A, S))
t /= t.sum(-1, keepdims=True)
r = rng.random((S, A))
ctx = native.Context(0)
model = ctx.load_dense(t, r, None)
v = rng.random(S)
for rep in range(3):
    ctx.vi_sweeps(model, 0.95, 20)
    ms, n = ctx.last_kernel_ms()
    print("dense S={} A={}: {:.3f} ms/sweep -> {:.2f} TB/s".format(S, A, ms / 20, 8.0 * S * S * A / (ms / 20 * 1e-3) / 1e12), flush=True)
