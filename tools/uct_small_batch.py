#!/usr/bin/env python3
"""Kernel time of the headline UCT plan (S = 10 000, |A| = 5, 33 x 30) at SURVEY 8(d)'s batch sizes, per kernel variant.

    [UCT_SB_SHAPE=10,10,10] [MP_UCT_QUAD=0|1] [MI355PLAN_LIB=rl_agents_amd/lib/prof/libmi355plan.so] python tools/uct_small_batch.py ROOTS...
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402


def main():
    shape = tuple(int(x) for x in os.environ.get("UCT_SB_SHAPE", "10,10,100").split(","))   # (lanes, speeds, cells): S = their product
    cfg = generators.highway_shaped(*shape, seed=0)
    ctx = native.Context(0)
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    nt = np.flatnonzero(~np.asarray(cfg["terminal"]))
    p = np.ones(5) / 5
    for n in [int(a) for a in sys.argv[1:]] or [1, 4096]:
        g = np.random.Generator(np.random.PCG64(1))
        s0 = torch.from_numpy(g.choice(nt, n).astype(np.int32)).cuda()
        rng0 = torch.from_numpy(native.seed_sequence_states((), 0, n).view(np.int64)).cuda()
        rng = rng0.clone()
        plans = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
        pl = torch.zeros(n, dtype=torch.int32, device="cuda")
        val = torch.zeros(n, dtype=torch.float64, device="cuda")
        es = torch.zeros(n, dtype=torch.int64, device="cuda")
        ms = []
        for _ in range(8 if (not os.environ.get("MI355PLAN_LIB") or os.environ.get("MI355PLAN_AB")) else 1):
            rng.copy_(rng0)
            torch.cuda.synchronize()
            ctx.uct_plan_device(model, n, s0, 33, 30, 0.8, 10.0, p, p, rng, 8, plans=plans, plan_len=pl, root_value=val, env_steps=es)
            ms.append(ctx.last_kernel_ms()[0])
        steps = int(es.sum().item())
        k = float(np.median(ms[2:])) if len(ms) > 2 else ms[0]
        print("quad={} roots={:6d} variant={:10s} kernel_ms={:.4f} env-steps/s={:.3g} env_steps={}".format(
            os.environ.get("MP_UCT_QUAD", "-"), n, ctx.last_kernel_variant(), k, steps / (k * 1e-3), steps), flush=True)


if __name__ == "__main__":
    main()
