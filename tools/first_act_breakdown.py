"""Where does the FIRST act() of a process go (profiles/r05_agent_latency.txt: ~0.9 s for MCTSAgent, 0.18 ms afterwards)?
The reference's benchmark mode starts one process per run (scripts/experiments.py:102-106), so a run pays it once.
    MI355PLAN_NO_TORCH=1 python tools/first_act_breakdown.py
"""
import os
import sys
import time

t_start = time.perf_counter()
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
marks = []


def mark(name, t0):
    marks.append((name, 1e3 * (time.perf_counter() - t0)))
    return time.perf_counter()


t = mark("import numpy", t_start)
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402
t = mark("import rl_agents_amd.native", t)
native.load()
t = mark("dlopen libmi355plan.so", t)
ctx = native.Context(0)
t = mark("Context(0): hipInit, device, stream", t)
cfg = generators.highway_shaped(10, 10, 100, seed=0)
t = mark("build tables (host)", t)
model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
t = mark("load_table: first kernels of api.hip's code object + uploads", t)
p = np.ones(5) / 5
rng = native.seed_sequence_states((), 0, 1)
s0 = np.array([0], dtype=np.int32)
out = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, p, p, rng, max_plan_len=1)
t = mark("first uct_plan: uct.hip's code object, workspaces, kernel", t)
out = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, p, p, rng, max_plan_len=1)
t = mark("second uct_plan", t)
out = ctx.opd_plan(model, s0, 5000, 0.8, 0, rng, max_plan_len=1)
t = mark("first opd_plan: opd.hip's code object", t)
q, sw = ctx.vi_solve(model, 0.95, 200)
t = mark("first vi_solve: vi.hip's code object", t)
for name, ms in marks:
    print("{:70s} {:9.2f} ms".format(name, ms))
