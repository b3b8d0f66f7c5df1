"""How much of the UCT kernel's time is lanes waiting for the slowest rollout of their wave?  The headline batch on the
highway-shaped table (a rollout ends at a terminal cell with ~5 % per step) against the SAME table without terminal
states (every rollout runs to the horizon: all 64 lanes of a wave busy in every rollout step).
    MI355PLAN_NO_TORCH=1 python tools/micro_uct_divergence.py    -> profiles/r03_uct_divergence.txt (through gpurun)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402


def main():
    import torch
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
    p = np.ones(5) / 5
    for n in (4096, 262144):
        s0 = np.random.Generator(np.random.PCG64(12345)).choice(np.flatnonzero(~term), size=n).astype(np.int32)
        d_s0 = torch.from_numpy(s0).cuda()
        d_rng = torch.from_numpy(native.seed_sequence_states((), 0, n).view(np.int64)).cuda()
        d_steps = torch.zeros(n, dtype=torch.int64, device="cuda")
        d_plans = torch.zeros((n, 8), dtype=torch.int32, device="cuda")
        d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
        for label, terminal in (("terminal cells (headline)", term), ("no terminal states", np.zeros_like(term))):
            model = ctx.load_table(t, r, terminal)
            ms = []
            for _ in range(6):
                ctx.uct_plan_device(model, n, d_s0, 33, 30, 0.8, 10.0, p, p, d_rng, 8, plans=d_plans, plan_len=d_len, env_steps=d_steps)
                ms.append(ctx.last_kernel_ms()[0])
            steps = int(d_steps.sum().item())
            k = float(np.median(ms[1:]))
            print("{:7d} roots  {:28s} kernel {:.3f} ms  env steps per root {:6.1f}  {:7.2f} G env-steps/s  {:.3f} ns per lane-step".format(
                n, label, k, steps / n, steps / k / 1e6, 1e6 * k / steps * (n / 64) / (n / 64)))
            model.close()


if __name__ == "__main__":
    main()
