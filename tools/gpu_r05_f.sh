#!/bin/bash
cd /root/repo
O=gpurun_out/r05f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_uct_quad.py tests/test_gpu_visits.py -x -q > $O/pytest_quad.log 2>&1
echo "pytest rc=$?" >> $O/pytest_quad.log
tail -30 $O/pytest_quad.log | cut -c1-250
for q in 1 0; do
  for n in 1 4096 16384 65536; do
    MP_UCT_QUAD=$q timeout 120 python - <<PY
import os, numpy as np, torch
from rl_agents_amd import native
from rl_agents_amd.envs import generators
cfg = generators.highway_shaped(10, 10, 100, seed=0)
ctx = native.Context(0)
model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
n = $n
g = np.random.Generator(np.random.PCG64(1))
nt = np.flatnonzero(~np.asarray(cfg["terminal"]))
s0 = torch.from_numpy(g.choice(nt, n).astype(np.int32)).cuda()
rng0 = torch.from_numpy(native.seed_sequence_states((), 0, n).view(np.int64)).cuda()
rng = rng0.clone()
plans = torch.zeros((n, 8), dtype=torch.int32, device="cuda"); pl = torch.zeros(n, dtype=torch.int32, device="cuda")
val = torch.zeros(n, dtype=torch.float64, device="cuda"); es = torch.zeros(n, dtype=torch.int64, device="cuda")
p = np.ones(5) / 5
ms = []
for i in range(8):
    rng.copy_(rng0); torch.cuda.synchronize()
    ctx.uct_plan_device(model, n, s0, 33, 30, 0.8, 10.0, p, p, rng, 8, plans=plans, plan_len=pl, root_value=val, env_steps=es)
    ms.append(ctx.last_kernel_ms()[0])
steps = int(es.sum().item())
print("quad=$q roots=%6d variant=%-10s kernel_ms=%.4f env-steps/s=%.3g" % (n, ctx.last_kernel_variant(), np.median(ms[2:]), steps / (np.median(ms[2:]) * 1e-3)))
PY
  done
done 2>&1 | grep -v amdgpu.ids | tee $O/quad_timing.txt
