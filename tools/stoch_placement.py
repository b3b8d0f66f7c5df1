"""Is uct_stoch_kernel's time (2.38 .. 2.72 ms between processes and boxes) a matter of where its arrays land?  One process,
the same plan after device allocations of different sizes made BEFORE the planner's workspaces exist (each round: fresh context).
    MI355PLAN_NO_TORCH=1 python tools/stoch_placement.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402

native.load()
hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
n = 262144
cfg = generators.highway_shaped(10, 10, 100, seed=0)
t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
s_, a_ = r.shape
nxt = np.stack([t, np.repeat(t[:, 1:2], a_, axis=1)], axis=-1).astype(np.int64)
pr = np.broadcast_to(np.array([0.8, 0.2]), (s_, a_, 2)).copy()
non_term = np.flatnonzero(~np.asarray(term))
s0 = np.random.Generator(np.random.PCG64(12345)).choice(non_term, size=n).astype(np.int32)
p = np.ones(a_) / a_
for pad_mb in (0, 1, 2, 5, 33, 128, 513, 1000, 0, 2049):
    pads = []
    if pad_mb:
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(pad_mb << 20)) == 0
        pads.append(ptr)
    ctx = native.Context(0)
    model = ctx.load_sparse(pr, nxt, r, term)
    rng = native.seed_sequence_states((), 0, n)
    erng = native.seed_sequence_states((), 10 ** 6, n)
    d_rng = ctx.device_rng(rng)
    ms = []
    for rep in range(4):
        ctx.uct_plan_stochastic(model, s0, 33, 30, 0.8, 10.0, p, p, d_rng, env_rng_state=erng, closed_loop=True, max_plan_len=8)
        ms.append(ctx.last_kernel_ms()[0])
    print("pad %5d MB: kernel ms %s" % (pad_mb, " ".join("%.3f" % m for m in ms)), flush=True)
    model.close()
    ctx.close()
    for ptr in pads:
        hip.hipFree(ptr)
