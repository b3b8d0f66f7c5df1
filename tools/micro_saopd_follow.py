"""State-aware OPD: one batch of planners through its first and N following plans (receding horizon), kernel ms per plan."""
import sys
import numpy as np
import torch
from rl_agents_amd import native
from rl_agents_amd.envs import generators

cfg = generators.gridworld()
t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
plans = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
model = ctx.load_table(t, r, term)
states = np.random.Generator(np.random.PCG64(12345)).integers(0, r.shape[0], size=n).astype(np.int32)
pl = native.StateAwarePlanners(ctx, model, n)
rng = native.seed_sequence_states((), 7, n)
ms = []
for _ in range(plans):
    o = pl.plan(states, 500, 0.8, 0.0, rng, max_plan_len=8)
    ms.append(ctx.last_kernel_ms()[0])
    states = np.where(o["plan_len"] > 0, t[states, np.maximum(o["plans"][:, 0], 0)], states).astype(np.int32)
print("saopd %d planners, plans 1..%d kernel ms: %s; mean updates of the last %.0f" % (n, plans, " ".join("%.2f" % m for m in ms), o["updates"].mean()))
pl.close()
