"""Stochastic-model UCT kernel, torch-free micro-benchmark (the bench's sparse highway-shaped model: intended successor 0.8,
IDLE's successor 0.2).    MI355PLAN_NO_TORCH=1 python tools/micro_uct_stoch.py [n_roots] [closed|open]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    closed = (sys.argv[2] if len(sys.argv) > 2 else "closed") == "closed"
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    s_, a_ = r.shape
    nxt = np.stack([t, np.repeat(t[:, 1:2], a_, axis=1)], axis=-1).astype(np.int64)
    pr = np.broadcast_to(np.array([0.8, 0.2]), (s_, a_, 2)).copy()
    ctx = native.Context(0)
    model = ctx.load_sparse(pr, nxt, r, term)
    non_term = np.flatnonzero(~np.asarray(term))
    s0 = np.random.Generator(np.random.PCG64(12345)).choice(non_term, size=n).astype(np.int32)
    rng = native.seed_sequence_states((), 0, n)
    erng = native.seed_sequence_states((), 10 ** 6, n)
    p = np.ones(a_) / a_
    d_rng = ctx.device_rng(rng)
    for rep in range(4):
        t0 = time.perf_counter()
        out = ctx.uct_plan_stochastic(model, s0, 33, 30, 0.8, 10.0, p, p, d_rng, env_rng_state=erng, closed_loop=closed, max_plan_len=8)
        dt = time.perf_counter() - t0
        ms, _ = ctx.last_kernel_ms()
        print("uct_stoch n={} {} kernel {:.3f} ms wall {:.1f} ms env_steps {} -> {:.3e} steps/s".format(
            n, "closed" if closed else "open", ms, dt * 1e3, int(out["env_steps"].sum()), out["env_steps"].sum() / (ms * 1e-3)))


if __name__ == "__main__":
    main()
