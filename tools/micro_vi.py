"""VI micro-benchmark without torch: MI355PLAN_NO_TORCH=1 python tools/micro_vi.py [sweeps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from rl_agents_amd import native  # noqa: E402
from rl_agents_amd.envs import generators  # noqa: E402

sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = generators.highway_shaped(10, 10, 100, seed=0)
ctx = native.Context(0)
model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
for rep in range(4):
    t0 = time.perf_counter()
    ctx.vi_sweeps(model, 0.95, sweeps)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ms, n = ctx.last_kernel_ms()
    print("vi det S=10000 A=5: {} sweeps kernel-batch {:.3f} ms ({:.2f} us/sweep), wall {:.3f} ms".format(
        sweeps, ms, 1e3 * ms / sweeps, dt * 1e3), flush=True)
t0 = time.perf_counter()
q, sw = ctx.vi_solve(model, 0.95, 200)
print("vi_solve: {} sweeps run, wall {:.3f} ms".format(sw, (time.perf_counter() - t0) * 1e3))
