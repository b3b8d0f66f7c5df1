"""One configuration of tools/micro_saopd_mix.py (for counter passes): micro_saopd_one.py light|heavy|mix [n] [reps]"""
import sys
import numpy as np
import torch
from rl_agents_amd import native
from rl_agents_amd.envs import generators

cfg = generators.gridworld()
t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
S = r.shape[0]
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = native.Context(0, torch.cuda.current_stream().cuda_stream)
model = ctx.load_table(t, r, term)
roots = {"light": np.full(n, 77, np.int32), "heavy": np.full(n, 10, np.int32),
         "mix": np.random.Generator(np.random.PCG64(12345)).integers(0, S, size=n).astype(np.int32)}[which]
ms = []
for _ in range(reps):
    pl = native.StateAwarePlanners(ctx, model, n)
    o = pl.plan(roots, 500, 0.8, 0.0, native.seed_sequence_states((), 7, n), max_plan_len=8)
    ms.append(ctx.last_kernel_ms()[0])
    pl.close()
print("saopd first plan, %s, n %d: mean updates %.0f, kernel ms %s" % (which, n, o["updates"].mean(), " ".join("%.2f" % m for m in ms)))
