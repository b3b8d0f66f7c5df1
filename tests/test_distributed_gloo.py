"""World-size-2 gloo test (CPU) of the N > 1 path: root sharding, global-index random streams and the
result all_gather.  The device planner is replaced by the CPU oracle behind the same planner interface,
so the test checks the distributed host logic: the gathered plans must equal the unsharded plans."""
import os
import socket
import sys

import numpy as np
import pytest
from tests.helpers import first_result

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OraclePlanner(object):
    """Stands in for the device MCTS planner: same plan_batch / batch_rng_states contract, oracle compute."""

    def __init__(self, cfg):
        from rl_agents_amd.agents.tree_search.abstract import AbstractPlanner
        self.cfg = cfg
        self._entropy = 1234
        self.batch_rng_states = AbstractPlanner.batch_rng_states.__get__(self)

    def plan_batch(self, env, root_states, root_steps=None, rng_states=None):
        from oracle import oracle
        p = np.ones(5) / 5
        out = oracle.uct_plan_batch(self.cfg["transition"], self.cfg["reward"], self.cfg["terminal"], root_states, 10, 8,
                                    0.8, 10.0, p, p, rng_states, max_plan_len=8)
        return dict(plans=out["plans"], plan_len=out["plan_len"], env_steps=out["env_steps"],
                    root_value=out["root_value"])


class FakeAgent(object):
    def __init__(self, cfg):
        self.env = type("E", (), {"unwrapped": None})()
        self.env.unwrapped = self.env
        self.config = {"env_preprocessors": []}
        self.planner = OraclePlanner(cfg)


def _worker(rank, world, port, n_roots, queue):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from rl_agents_amd.distributed import all_gather_rows, plan_batch_sharded, shard_bounds
    from rl_agents_amd.envs import generators
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{}".format(port), rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_roots, rank, world)
        rows = np.arange(n_roots * 3, dtype=np.int64).reshape(n_roots, 3)
        full = all_gather_rows(rows[lo:hi], n_roots)
        assert np.array_equal(full, rows)
        cfg = generators.highway_shaped(3, 4, 10, seed=3)
        roots = np.arange(n_roots, dtype=np.int32) % 120
        out = plan_batch_sharded(FakeAgent(cfg), roots)
        if rank == 0:
            queue.put({k: v for k, v in out.items()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_roots", [9, 16])
def test_sharded_plans_equal_unsharded(n_roots):
    import torch.multiprocessing as mp
    from rl_agents_amd.distributed import plan_batch_sharded
    from rl_agents_amd.envs import generators
    ctx = mp.get_context("spawn")
    queue = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_roots, queue)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = first_result(queue, procs, 240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = generators.highway_shaped(3, 4, 10, seed=3)
    single = plan_batch_sharded(FakeAgent(cfg), np.arange(n_roots, dtype=np.int32) % 120)   # world size 1
    for k in ("plans", "plan_len", "env_steps", "root_value"):
        assert np.array_equal(sharded[k], single[k]), k


class NumpyBackupCtx(object):
    """Stands in for the device context of the row-sharded VI driver: same load_dense_rows / vi_backup contract."""

    class Block(object):
        def __init__(self, t, r, term):
            self.t, self.r, self.term = np.asarray(t), np.asarray(r), term

        def close(self):
            pass

    def load_dense_rows(self, t, r, term=None):
        return self.Block(t, r, term)

    def vi_backup(self, model, gamma, v, robust=False):
        nv = (model.t * v.reshape((1,) * (model.t.ndim - 1) + (-1,))).sum(axis=-1)
        if robust:
            return (model.r + gamma * nv).min(axis=0)
        if model.term is not None:
            nv[np.asarray(model.term, bool)] = 0
        return model.r + gamma * nv


def _vi_worker(rank, world, port, queue):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from rl_agents_amd.distributed import vi_solve_row_sharded
    from rl_agents_amd.envs import generators
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{}".format(port), rank=rank, world_size=world)
    try:
        cfg = generators.random_stochastic(57, 3, seed=2, terminal_rate=0.1)
        q, sweeps = vi_solve_row_sharded(NumpyBackupCtx(), cfg["transition"], cfg["reward"], cfg["terminal"],
                                         gamma=0.9, iterations=200)
        if rank == 0:
            queue.put((q, sweeps))
    finally:
        dist.destroy_process_group()


def test_row_sharded_value_iteration_matches_reference_semantics():
    """World size 2: per-sweep all_gather of V + all_reduce of the allclose test reproduce the reference's
    fixed_point_iteration (value_iteration.py:65-73), i.e. the oracle's Q and sweep count."""
    import torch.multiprocessing as mp
    from oracle import oracle
    from rl_agents_amd.envs import generators
    ctx = mp.get_context("spawn")
    queue = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_vi_worker, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    q, sweeps = first_result(queue, procs, 240)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = generators.random_stochastic(57, 3, seed=2, terminal_rate=0.1)
    q_ref, sweeps_ref = oracle.vi_solve("stochastic", cfg["transition"], cfg["reward"], cfg["terminal"], gamma=0.9,
                                        iterations=200)
    assert sweeps == sweeps_ref
    np.testing.assert_allclose(q, q_ref, rtol=1e-13, atol=1e-13)
