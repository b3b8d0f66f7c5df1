"""N > 1 with the REAL kernels (task 1): world-size-2 process groups whose ranks both run the HIP planners /
the dense backup on the one GPU of the box (gloo rendezvous; RCCL refuses two ranks on one device), asserted equal to
the single-process result -- root sharding with global-index random streams + the packed result gather for UCT, OPD and
discrete robust OPD, and the row-sharded dense (robust) value iteration with an uneven shard.  A second group of tests
runs RCCL itself (backend "nccl", one rank, collectives forced) on the very tensors the product exchanges.
tests/test_distributed_gloo.py covers the same host logic on CPU with stand-ins; this file has no stand-ins."""
import os
import socket
import sys

import numpy as np
import pytest
from tests.helpers import first_result

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

UCT = "<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>"
OPD = "<class 'rl_agents_amd.agents.tree_search.deterministic.DeterministicPlannerAgent'>"
DRP = "<class 'rl_agents_amd.agents.robust.robust.DiscreteRobustPlannerAgent'>"

N_ROOTS = 203           # uneven over two ranks (102 + 101); more than three wavefronts per rank


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _models():
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 12, seed=5)
    cfg2 = generators.rewire(cfg, 0.15, seed=6)
    return cfg, cfg2


def _agents():
    """(name, agent, keys) for the three tree-search planners, seeded: the same objects in every process."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv
    cfg, cfg2 = _models()

    def table(c):
        return dict(mode="deterministic", transition=c["transition"].tolist(), reward=c["reward"].tolist(),
                    terminal=np.asarray(c["terminal"]).astype(int).tolist())
    env = FiniteMDPEnv(table(cfg))
    env.reset()
    out = []
    uct = agent_factory(env, dict(__class__=UCT, budget=400, gamma=0.8))
    opd = agent_factory(env, dict(__class__=OPD, budget=300, gamma=0.8))
    drp = agent_factory(env, dict(__class__=DRP, budget=300, gamma=0.8,
                                  models=[[], [{"method": "copy_with_config", "args": table(cfg2)}]]))
    for name, agent in (("uct", uct), ("opd", opd), ("ropd", drp)):
        agent.seed(77)
        out.append((name, agent))
    return out, cfg


def _plan_all(force_collective=False):
    from rl_agents_amd.distributed import plan_batch_sharded
    agents, cfg = _agents()
    non_term = np.flatnonzero(~np.asarray(cfg["terminal"]))
    roots = np.random.Generator(np.random.PCG64(3)).choice(non_term, size=N_ROOTS).astype(np.int32)
    res = {}
    for name, agent in agents:
        res[name] = plan_batch_sharded(agent, roots, force_collective=force_collective)
    # the device-resident form of the same exchange (mp_pack_rows -> one all_gather_into_tensor -> mp_unpack_rows, no host
    # hop under RCCL): fresh agents (same seeds), global roots as a device tensor
    import torch
    from rl_agents_amd.distributed import plan_batch_sharded_device
    agents, _ = _agents()
    d_roots = torch.from_numpy(roots).cuda()
    for name, agent in agents[:2]:
        out = plan_batch_sharded_device(agent, d_roots, max_plan_len=res[name]["plans"].shape[1], force_collective=force_collective)
        res[name + "_device"] = {k: out[k].cpu().numpy() for k in ("plans", "plan_len", "value", "env_steps", "status")}
    return res


def _plan_sequence(force_collective=False):
    """ShardedDevicePlan with the exchange on its side stream (overlap=True), THREE consecutive plan() calls on changing
    roots, consumed one call late (call k is read after call k + 1 was enqueued: the double-buffered path), in both
    payloads.  -> {payload/agent: [per call {plans, plan_len, value, env_steps, status}]}"""
    import torch
    from rl_agents_amd.distributed import ShardedDevicePlan
    _, cfg = _agents()
    non_term = np.flatnonzero(~np.asarray(cfg["terminal"]))
    g = np.random.Generator(np.random.PCG64(11))
    roots = [torch.from_numpy(g.choice(non_term, size=N_ROOTS).astype(np.int32)).cuda() for _ in range(3)]
    res = {}
    for payload in ("compact", "full"):
        agents, _ = _agents()
        for name, agent in agents[:2]:
            sp = ShardedDevicePlan(agent, N_ROOTS, max_plan_len=6, force_collective=force_collective, overlap=True,
                                   payload=payload, time_exchange=True)
            calls, pending = [], None
            for k in range(3):
                out = sp.plan(roots[k])
                if pending is not None:
                    sp.wait(pending)
                    calls.append({key: pending[key].cpu().numpy().copy() for key in ("plans", "plan_len", "value", "env_steps", "status")})
                pending = out
            sp.wait(pending)
            calls.append({key: pending[key].cpu().numpy().copy() for key in ("plans", "plan_len", "value", "env_steps", "status")})
            ms = sp.last_exchange_ms()
            assert ms is None or ms >= 0.0
            res["{}/{}".format(payload, name)] = calls
    return res


def _vi_problem(robust):
    from rl_agents_amd.envs import generators
    s = 301                                         # 151 + 150 rows
    cfg = generators.random_stochastic(s, 3, seed=21, terminal_rate=0.08)
    if not robust:
        return cfg["transition"], cfg["reward"], cfg["terminal"], s
    cfg2 = generators.random_stochastic(s, 3, seed=22)
    return (np.stack([cfg["transition"], cfg2["transition"]]), np.stack([cfg["reward"], cfg2["reward"] * 0.9]), None, s)


def _vi_sharded(robust, force_collective=False):
    """This rank's row block through vi_solve_row_sharded_device (every sweep on the device, V exchanged per sweep)."""
    import torch
    from rl_agents_amd import native
    from rl_agents_amd.distributed import rank_world, shard_bounds, vi_solve_row_sharded_device
    t, r, term, s = _vi_problem(robust)
    rank, world = rank_world()
    lo, hi = shard_bounds(s, rank, world)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        c = native.Context(0, torch.cuda.current_stream().cuda_stream)
        d_t = torch.from_numpy(np.ascontiguousarray(t[..., lo:hi, :, :])).cuda()
        d_r = torch.from_numpy(np.ascontiguousarray(r[..., lo:hi, :])).cuda()
        d_term = None if term is None else torch.from_numpy(np.asarray(term[lo:hi]).astype(np.uint8)).cuda()
        q, sweeps = vi_solve_row_sharded_device(c, d_t, d_r, d_term, s, (lo, hi), gamma=0.9, iterations=120,
                                                robust=robust, check_every=4, force_collective=force_collective)
        q = q.cpu().numpy()
        c.close()
    return q, sweeps


def _evaluate(sharded):
    """BatchedEvaluation of 150 episodes (device-resident loop), optionally sharded over the process group."""
    from rl_agents_amd.agents.common.factory import agent_factory
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    from rl_agents_amd.trainer.batched_evaluation import BatchedEvaluation
    cfg = dict(generators.highway_shaped(3, 4, 10, seed=3), state=2, max_steps=8)
    env = FiniteMDPEnv(cfg)
    env.reset()
    agent = agent_factory(env, dict(__class__=UCT, budget=100, gamma=0.9))
    starts = (np.arange(150) * 7 % 100).astype(np.int32)
    out = BatchedEvaluation(env, agent, num_episodes=150, sim_seed=9, max_steps=8, device_resident=True, sharded=sharded).run(
        initial_states=starts)
    return {k: out[k] for k in ("returns", "discounted_returns", "lengths", "actions", "planner_env_steps")}


def _worker(rank, world, port, backend, queue):
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:{}".format(port), rank=rank, world_size=world, **kw)
    try:
        force = world == 1
        res = dict(plans=_plan_all(force_collective=force), vi=_vi_sharded(False, force), rvi=_vi_sharded(True, force),
                   evaluation=_evaluate(sharded=True), sequence=_plan_sequence(force_collective=force),
                   backend=dist.get_backend(), world=dist.get_world_size())
        from rl_agents_amd import native
        res["lib"] = native.lib_path()
        if rank == 0:
            queue.put(res)
    finally:
        dist.destroy_process_group()


def _run_group(world, backend):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    queue = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, queue)) for r in range(world)]
    for p in procs:
        p.start()
    res = first_result(queue, procs, 300)
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    return res


@pytest.fixture(scope="module")
def single():
    """World size 1, no process group: the reference result of every comparison below."""
    return dict(plans=_plan_all(), vi=_vi_sharded(False), rvi=_vi_sharded(True), evaluation=_evaluate(sharded=False),
                sequence=_plan_sequence())


def _assert_same(res, single):
    for name in ("uct", "opd", "ropd"):
        a, b = res["plans"][name], single["plans"][name]
        assert set(a) == set(b), (name, set(a), set(b))
        for k in a:
            assert a[k].shape[0] == N_ROOTS
            np.testing.assert_array_equal(a[k], b[k], err_msg="{}/{}".format(name, k))
    for name, vkey in (("uct", "root_value"), ("opd", "root_lower")):      # device-resident exchange == host-packed exchange
        for r in (res, single):
            d, h = r["plans"][name + "_device"], r["plans"][name]
            assert d["plans"].shape == (N_ROOTS, h["plans"].shape[1])
            np.testing.assert_array_equal(d["plans"], h["plans"], err_msg=name + "_device/plans")
            np.testing.assert_array_equal(d["plan_len"], h["plan_len"], err_msg=name + "_device/plan_len")
            np.testing.assert_array_equal(d["env_steps"], h["env_steps"], err_msg=name + "_device/env_steps")
            np.testing.assert_array_equal(d["value"], h[vkey], err_msg=name + "_device/value")
            assert not d["status"].any()
    # three consecutive plans through the side-stream, double-buffered exchange (it had no multi-call test)
    for key, calls in res["sequence"].items():
        ref_calls = single["sequence"][key]
        assert len(calls) == len(ref_calls) == 3
        compact = key.startswith("compact")
        for k, (a, b) in enumerate(zip(calls, ref_calls)):
            tag = "sequence/{}/call{}".format(key, k)
            assert a["plans"].shape[0] == N_ROOTS
            width = a["plans"].shape[1]
            assert width == (1 if compact else 6)
            np.testing.assert_array_equal(a["plans"], b["plans"][:, :width], err_msg=tag)
            np.testing.assert_array_equal(a["value"], b["value"], err_msg=tag)
            np.testing.assert_array_equal(a["env_steps"], b["env_steps"], err_msg=tag)
            np.testing.assert_array_equal(a["status"], b["status"], err_msg=tag)
            np.testing.assert_array_equal(a["plan_len"], np.minimum(b["plan_len"], width) if compact else b["plan_len"], err_msg=tag)
        assert not np.array_equal(calls[0]["value"], calls[1]["value"])      # (the calls do differ: different roots)
    for k in ("returns", "discounted_returns", "lengths", "actions"):     # sharded device-resident evaluation
        np.testing.assert_array_equal(res["evaluation"][k], single["evaluation"][k], err_msg="evaluation/" + k)
    assert res["evaluation"]["planner_env_steps"] == single["evaluation"]["planner_env_steps"]
    for k in ("vi", "rvi"):
        assert res[k][1] == single[k][1], k                       # same sweep count
        np.testing.assert_array_equal(res[k][0], single[k][0], err_msg=k)


def test_single_process_results_match_the_oracle(single):
    """Anchor: what the sharded runs are compared with is itself the oracle's result (UCT / OPD plans; dense VI Q)."""
    from oracle import oracle
    cfg, _ = _models()
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    agents, _ = _agents()
    non_term = np.flatnonzero(~np.asarray(term))
    roots = np.random.Generator(np.random.PCG64(3)).choice(non_term, size=N_ROOTS).astype(np.int32)
    uct = agents[0][1]
    rng = uct.planner.batch_rng_states(N_ROOTS)
    c = uct.planner.config
    p = np.ones(r.shape[1]) / r.shape[1]
    ref = oracle.uct_plan_batch(t, r, term, roots, c["episodes"], c["horizon"], c["gamma"], c["temperature"], p, p, rng,
                                max_plan_len=c["horizon"])
    np.testing.assert_array_equal(single["plans"]["uct"]["plans"], ref["plans"])
    np.testing.assert_array_equal(single["plans"]["uct"]["env_steps"], ref["env_steps"])
    opd = agents[1][1]
    rng = opd.planner.batch_rng_states(N_ROOTS)
    ref = oracle.opd_plan_batch(t, r, term, roots, 300, 0.8, 0.0, rng, max_plan_len=single["plans"]["opd"]["plans"].shape[1])
    np.testing.assert_array_equal(single["plans"]["opd"]["plans"], ref["plans"])
    np.testing.assert_array_equal(single["plans"]["opd"]["root_lower"], ref["root_lower"])
    tt, rr, tm, _ = _vi_problem(False)
    q_ref, sweeps_ref = oracle.vi_solve("stochastic", tt, rr, tm, gamma=0.9, iterations=120)
    assert single["vi"][1] == sweeps_ref
    assert np.array_equal(single["vi"][0], q_ref)       # dense VI: numpy's order of additions on the device


def test_world2_real_kernels_equal_world1(single):
    """Two ranks, both on this GPU, gloo rendezvous: sharded UCT / OPD / robust-OPD plans and row-sharded dense VI / robust
    VI (uneven shards) are bit-identical to the single-process results."""
    res = _run_group(2, "gloo")
    assert res["world"] == 2 and res["lib"].endswith("libmi355plan.so")
    _assert_same(res, single)


def test_rccl_single_rank_collectives_on_product_tensors(single):
    """backend "nccl" (= RCCL) with one rank and the collectives forced: all_gather_into_tensor of the packed per-root
    results and of V, all_reduce of the allclose verdict run through RCCL on the product's device tensors."""
    res = _run_group(1, "nccl")
    assert res["backend"] == "nccl" and res["world"] == 1
    _assert_same(res, single)


def test_c_abi_collective_delivers_the_same_bytes():
    """mp_comm_unique_id / mp_comm_init / mp_gather_results (the collective for a consumer without
    torch.distributed; RCCL resolved with dlopen) on a single-rank communicator: pack -> gather -> unpack is the identity,
    i.e. exactly what ShardedDevicePlan's torch.distributed exchange delivers for world 1."""
    import torch
    from rl_agents_amd import native
    ctx = native.Context(0)
    dev = torch.device("cuda", 0)
    n, mpl = 1000, 3
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    plans = torch.randint(-1, 5, (n, mpl), dtype=torch.int32, device=dev, generator=g)
    value = torch.rand(n, dtype=torch.float64, device=dev, generator=g)
    steps = torch.randint(0, 10 ** 6, (n,), dtype=torch.int64, device=dev, generator=g)
    row = 4 * mpl + 8 + 8
    packed = torch.empty((n, row), dtype=torch.uint8, device=dev)
    gathered = torch.zeros((n, row), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.comm_init(0, 1, native.Context.comm_unique_id())
    ctx.pack_rows([plans, value, steps], n, packed)
    ctx.gather_results(packed, gathered)
    out = [torch.empty_like(plans), torch.empty_like(value), torch.empty_like(steps)]
    ctx.unpack_rows(gathered, n, 1, out)
    ctx.synchronize()
    assert torch.equal(gathered, packed)
    assert torch.equal(out[0], plans) and torch.equal(out[1], value) and torch.equal(out[2], steps)
    ctx.comm_destroy()
    with pytest.raises(native.NativeError):
        ctx.gather_results(packed, gathered)          # no communicator any more
    ctx.close()
