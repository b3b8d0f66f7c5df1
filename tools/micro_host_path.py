"""Host-inclusive mp_uct_plan at 262 144 roots with pinned (mp_host_alloc) arrays and device-resident generator records:
the default path (two pipelined chunks), zero-copy forced, copies after one launch, other chunkings.
    MI355PLAN_NO_TORCH=1 python tools/micro_host_path.py      -> profiles/r03_host_path.txt (through gpurun)"""
import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from rl_agents_amd import native
from rl_agents_amd.envs import generators
cfg = generators.highway_shaped(10, 10, 100, seed=0)
t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
n = 262144
ctx = native.Context(0)
model = ctx.load_table(t, r, term)
s0 = np.random.Generator(np.random.PCG64(12345)).choice(np.flatnonzero(~term), size=n).astype(np.int32)
rng0 = native.seed_sequence_states((), 0, n)
p = np.ones(5) / 5
def run(label, env):
    for k in ("MP_PIPE_CHUNK", "MP_PIPE_STREAMS", "MP_NO_ZERO_COPY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bufs = ctx.plan_buffers(n, 8, outputs=("plans", "plan_len", "root_value", "env_steps"))
    bufs["root_state"][:] = s0
    rngd = ctx.device_rng(rng0)
    ctx.uct_plan(model, bufs["root_state"], 33, 30, 0.8, 10.0, p, p, rngd, out=bufs)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.uct_plan(model, bufs["root_state"], 33, 30, 0.8, 10.0, p, p, rngd, out=bufs)
        ts.append(time.perf_counter() - t0)
    print("%-50s %.3f ms (min %.3f) kernel-bracket %.3f ms launches %d" % (label, 1e3 * np.median(ts), 1e3 * min(ts), *ctx.last_kernel_ms()), flush=True)
    rngd.close(); bufs.close()
run("default (pinned)", {}); run("zero-copy forced", {"MP_ZERO_COPY_MAX": "100000000"})
run("copies, one launch", {"MP_NO_ZERO_COPY": "1", "MP_PIPE_CHUNK": "0"})
for chunk, streams in ((131072, 2), (65536, 4)):
    run("copies, chunk %d x %d streams" % (chunk, streams), {"MP_NO_ZERO_COPY": "1", "MP_PIPE_CHUNK": str(chunk), "MP_PIPE_STREAMS": str(streams)})
