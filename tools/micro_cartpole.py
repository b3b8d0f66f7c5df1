"""CartPole UCT micro-benchmark without torch: MI355PLAN_NO_TORCH=1 python tools/micro_cartpole.py [n_roots]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import CartPoleEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = native.Context(0)
model = ctx.load_cartpole(CartPoleEnv().cartpole_params())
x0 = np.random.Generator(np.random.PCG64(0)).uniform(-0.05, 0.05, size=(n, 4))
rng = np.random.Generator(np.random.PCG64(1)).integers(1, 2 ** 62, size=(n, 6)).astype(np.uint64)
rng[:, 3] |= 1
rng[:, 4:] = 0
p = np.ones(2) / 2
best = 1e9
for rep in range(int(os.environ.get('REPS', 40))):
    out = ctx.uct_plan(model, x0, 20, 50, 0.8, 10.0, p, p, rng, max_plan_len=8)
    ms, _ = ctx.last_kernel_ms()
    best = min(best, ms)
    if rep % 13:
        continue
    print("cartpole uct n={}: kernel {:.3f} ms, env_steps {} -> {:.3e} steps/s".format(n, ms, int(out["env_steps"].sum()),
                                                                                   out["env_steps"].sum() / (ms * 1e-3)), flush=True)
print('best of reps: {:.4f} ms'.format(best))
