import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from rl_agents_amd import native
g = np.random.Generator(np.random.PCG64(5))
s, a = 10000, 5
cfg = dict(transition=g.integers(0, s, size=(s, a)), reward=g.random((s, a)), terminal=g.random(s) < 0.02)
ctx = native.Context(0)
model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
p = np.ones(a) / a
for n in (1, 64, 256, 1024):
    s0 = torch.from_numpy(g.integers(0, s, n).astype(np.int32)).cuda()
    rng0 = torch.from_numpy(native.seed_sequence_states((), 0, n).view(np.int64)).cuda()
    ms = []
    for _ in range(6):
        rng = rng0.clone(); torch.cuda.synchronize()
        ctx.uct_plan_device(model, n, s0, 33, 30, 0.8, 10.0, p, p, rng, 8)
        ms.append(ctx.last_kernel_ms()[0])
    print("LONE_WAVES=%s roots=%5d variant=%-14s kernel_ms=%.4f" % (os.environ.get("MP_UCT_LONE_WAVES", "-"), n, ctx.last_kernel_variant(), float(np.median(ms[2:]))))
