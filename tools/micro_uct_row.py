#!/usr/bin/env python3
"""Kernel time of the one-MDP-per-root UCT plan over (episodes, horizon, roots): separates per-launch, per-episode and per-step
cost of uct_row_kernel / uct_lone_kernel<EACH> / the gather kernel.  usage: python tools/micro_uct_row.py [roots ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from benchmarks.common import _episode_tables, seed_states  # noqa: E402
from rl_agents_amd import native  # noqa: E402


def main():
    roots = [int(a) for a in sys.argv[1:]] or [4096]
    ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    for n in roots:
        tr, rw, tm = _episode_tables(n, (3, 4, 10), seed0=7, distinct=min(n, 4096))
        model = ctx.load_table_batch(tr, rw, tm)
        g = np.random.Generator(np.random.PCG64(1))
        s0 = g.integers(0, tr.shape[1], n).astype(np.int32)
        rng0 = seed_states(np.arange(n))
        d = dict(mi=torch.arange(n, dtype=torch.int32, device=dev), s0=torch.from_numpy(s0).to(dev),
                 rng=torch.from_numpy(rng0.view(np.int64)).to(dev), plans=torch.full((n, 8), -1, dtype=torch.int32, device=dev),
                 plan_len=torch.zeros(n, dtype=torch.int32, device=dev), value=torch.zeros(n, dtype=torch.float64, device=dev),
                 steps=torch.zeros(n, dtype=torch.int64, device=dev))
        p = np.ones(5) / 5
        for e, h in ((33, 30), (33, 15), (33, 2), (16, 30), (1, 30), (1, 1)):
            ms = []
            for _ in range(6):
                ctx.uct_plan_device(model, n, d["s0"], e, h, 0.8, 10.0, p, p, d["rng"], 8, plans=d["plans"], plan_len=d["plan_len"],
                                    root_value=d["value"], env_steps=d["steps"], model_index=d["mi"])
                torch.cuda.synchronize()
                ms.append(ctx.last_kernel_ms()[0])
            print("roots {:6d} E {:2d} H {:2d}  {:14s} kernel {:.4f} ms  env steps {}".format(
                n, e, h, ctx.last_kernel_variant(), float(np.median(ms[1:])), int(d["steps"].sum().item())))
        model.close()


if __name__ == "__main__":
    main()
