#!/bin/bash
cd /root/repo
O=gpurun_out/r05h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu_full.log 2>&1
echo "pytest rc=$?" >> $O/pytest_gpu_full.log
tail -25 $O/pytest_gpu_full.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
python bench.py --workload uct_cartpole --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cartpole kernel_ms', d['roofline']['kernel_ms'], d['value'], d['parity_sample'])"
MP_CARTPOLE_SINCOS=device python bench.py --workload uct_cartpole --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cartpole(device sincos) kernel_ms', d['roofline']['kernel_ms'], d['value'])"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -30
import cProfile, pstats, time, sys
sys.path.insert(0, '.')
import numpy as np
from rl_agents_amd.agents.common.factory import agent_factory
from rl_agents_amd.envs import FiniteMDPEnv, generators
cfg = generators.highway_shaped(10, 10, 100, seed=0)
env = FiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"]))
obs = env.reset()[0]
agent = agent_factory(env, dict(__class__="<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>", budget=1000, gamma=0.8, horizon=30, episodes=33))
agent.seed(0)
for _ in range(3):
    obs = env.step(agent.act(obs))[0]
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    a = agent.act(obs)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
PY
