"""cProfile of MCTSAgent.act() on the headline table (what a user of the reference's agent pays per environment step).
    python tools/act_profile.py [n_calls]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_agents_amd.agents.common.factory import agent_factory  # noqa: E402
from rl_agents_amd.envs import FiniteMDPEnv, generators  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cfg = generators.highway_shaped(10, 10, 100, seed=0)
env = FiniteMDPEnv(dict(mode="deterministic", transition=cfg["transition"], reward=cfg["reward"], terminal=cfg["terminal"]))
obs = env.reset()
obs = obs[0] if isinstance(obs, tuple) else obs
agent = agent_factory(env, dict(__class__="<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>", budget=1000, gamma=0.8, horizon=30,
                                episodes=33))
agent.seed(0)
for _ in range(20):
    agent.act(obs)
t0 = time.perf_counter()
for _ in range(n):
    agent.act(obs)
wall = (time.perf_counter() - t0) / n
print("act(): {:.4f} ms per call (no profiler)".format(1e3 * wall))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    agent.act(obs)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
