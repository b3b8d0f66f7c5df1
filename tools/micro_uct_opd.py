"""Kernel micro-benchmarks without torch (MI355PLAN_NO_TORCH=1): quick A/B of launch geometry on the GPU box.

    MI355PLAN_NO_TORCH=1 python tools/micro_uct_opd.py uct|opd [n_roots]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "uct"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    shape = [int(x) for x in os.environ.get("SHAPE", "10,10,100").split(",")]   # SHAPE=3,4,10: a table that fits L1
    cfg = generators.highway_shaped(*shape, seed=0)
    ctx = native.Context(0)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    non_term = np.flatnonzero(~cfg["terminal"])
    s0 = np.random.Generator(np.random.PCG64(12345)).choice(non_term, size=n).astype(np.int32)
    rng = np.random.Generator(np.random.PCG64(1)).integers(1, 2 ** 62, size=(n, 6)).astype(np.uint64)
    rng[:, 3] |= 1
    rng[:, 4:] = 0
    p = np.ones(5) / 5
    policy = None
    if os.environ.get("POLICY"):  # per-state prior / rollout tables (mp_uct_plan_policy)
        w = np.random.Generator(np.random.PCG64(2)).random((2, shape[0] * shape[1] * shape[2], 5)) ** 2
        if os.environ["POLICY"] == "uniform":  # same plans as the state-independent uniform policy: kernel A/B
            w = np.ones((2, shape[0] * shape[1] * shape[2], 5))
        policy = ctx.load_policy(model, w[0] / w[0].sum(1, keepdims=True), w[1] / w[1].sum(1, keepdims=True))
    if what == "saopd":  # state-aware OPD, the reference's GridWorld config (budget 500, gamma 0.8), 3 consecutive plans
        cfg = generators.gridworld()
        model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
        budget = int(sys.argv[3]) if len(sys.argv) > 3 else 500
        planners = native.StateAwarePlanners(ctx, model, n)
        states = np.random.Generator(np.random.PCG64(3)).integers(0, 100, size=n).astype(np.int32)
        for rep in range(int(os.environ.get("PLANS", "3"))):   # PLANS=12: a longer receding-horizon episode
            t0 = time.perf_counter()
            out = planners.plan(states, budget, 0.8, 0.0, rng, max_plan_len=8)
            dt = time.perf_counter() - t0
            ms, _ = ctx.last_kernel_ms()
            ok = out["status"] == 0
            print("saopd n={} budget={} plan#{} kernel {:.3f} ms wall {:.1f} ms  ok {}  updates/planner {:.0f}  -> {:.3e} "
                  "expansion-steps/s".format(n, budget, rep, ms, dt * 1e3, int(ok.sum()), out["updates"][ok].mean(),
                                             out["env_steps"].sum() / (ms * 1e-3)))
            states = np.where(out["plan_len"] > 0, cfg["transition"][states, np.maximum(out["plans"][:, 0], 0)],
                              states).astype(np.int32)
        return
    for rep in range(3):
        t0 = time.perf_counter()
        if what == "uct":
            out = ctx.uct_plan(model, s0, int(os.environ.get("EPISODES", "33")), int(os.environ.get("HORIZON", "30")), 0.8, 10.0, p, p,
                               rng, max_plan_len=8, policy=policy)
        else:
            budget = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
            out = ctx.opd_plan(model, s0, budget, 0.8, 0.0, rng, max_plan_len=32)
        dt = time.perf_counter() - t0
        ms, _ = ctx.last_kernel_ms()
        print("{} n={} lanes={} kernel {:.3f} ms  wall {:.1f} ms  env_steps {}  -> {:.3e} steps/s".format(
            what, n, os.environ.get("MP_UCT_LANES", "auto"), ms, dt * 1e3, int(out["env_steps"].sum()),
            out["env_steps"].sum() / (ms * 1e-3)), flush=True)


if __name__ == "__main__":
    main()
