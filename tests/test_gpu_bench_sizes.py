"""Parity AT THE BENCHMARKED GEOMETRIES (task 1): every configuration a bench.py line is quoted on is run
here at its full batch size through the C ABI and a seeded random sample of >= 2048 roots / planners of that very
launch is compared with the CPU oracle bit for bit (a root's result depends on its own state and stream only, so the
oracle can replay any subset).  The launch geometry, variant switches (packed 16-byte policy records above 16 384
roots, the high-occupancy OPD variant above the LDS residency limit) and tree strides are the ones the bench uses --
nothing is forced through an environment knob.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLE = 2048


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def fast_rng_states(n, seed):
    """n valid numpy-PCG64 state records without n SeedSequence constructions: any 128-bit state with an odd
    increment is a state the generator can be in (has_uint32 = 0: no buffered half)."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = g.integers(0, 2 ** 63, size=(n, 6), dtype=np.int64).astype(np.uint64)
    out[:, 1] ^= g.integers(0, 2 ** 63, size=n, dtype=np.int64).astype(np.uint64) << np.uint64(1)
    out[:, 3] |= np.uint64(1)
    out[:, 4:] = 0
    return np.ascontiguousarray(out)


def headline_model(ctx):
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    return t, r, term, ctx.load_table(t, r, term)


def bench_roots(term, n, seed=12345):
    non_term = np.flatnonzero(~np.asarray(term))
    return np.random.Generator(np.random.PCG64(seed)).choice(non_term, size=n).astype(np.int32)


@pytest.mark.parametrize("with_prior", [False, True], ids=["uct", "uct_prior"])
def test_uct_262144_roots_sample_vs_oracle(ctx, with_prior):
    """bench.py default (and --workload uct_prior): highway-shaped S = 10 000, A = 5, 33 episodes x horizon 30,
    262 144 roots in one launch; with per-state policies this size selects the packed 16-byte records by itself."""
    from oracle import oracle
    t, r, term, model = headline_model(ctx)
    n = 262144
    s0 = bench_roots(term, n)
    rng = fast_rng_states(n, 1)
    rng0 = rng.copy()
    p = np.ones(5) / 5
    policy, tables = None, None
    if with_prior:
        q, _ = ctx.vi_solve(model, 0.95, 200)
        z = np.exp((q - q.max(axis=1, keepdims=True)) / 0.3)
        tables = z / z.sum(axis=1, keepdims=True)
        policy = ctx.load_policy(model, tables, tables)
    out = ctx.uct_plan(model, s0, 33, 30, 0.8, 2 / (1 - 0.8), p, p, rng, max_plan_len=8, policy=policy)
    idx = np.sort(np.random.Generator(np.random.PCG64(7)).choice(n, size=SAMPLE, replace=False))
    idx[:3] = (0, 1, 63)
    idx[-3:] = (n - 64, n - 2, n - 1)          # first / last waves of the launch are in the sample
    pp = tables if with_prior else p
    ref = oracle.uct_plan_batch(t, r, term, s0[idx], 33, 30, 0.8, 2 / (1 - 0.8), pp, pp, rng0[idx], max_plan_len=8,
                                n_threads=16)
    for k in ("plans", "plan_len", "root_child_count", "env_steps"):
        np.testing.assert_array_equal(out[k][idx], ref[k], err_msg=k)
    assert np.array_equal(out["root_value"][idx], ref["root_value"])
    assert np.array_equal(out["root_child_value"][idx], ref["root_child_value"])
    np.testing.assert_array_equal(rng[idx], ref["rng_after"])
    # whole-batch sanity: every root ran its 33 episodes (the root is visited once per episode)
    assert (out["root_child_count"].sum(axis=1) == 32).all()   # episode 0 expands the root and visits no child
    assert out["env_steps"].min() > 0 and out["env_steps"].max() <= 33 * 30
    # a sampled tree of the full-stride batch equals the oracle's tree
    from tests.helpers import bfs_order
    for root in (int(idx[5]), n - 1):
        tree = ctx.uct_tree(root)
        o = oracle.uct_plan(t, r, term, int(s0[root]), 33, 30, 0.8, 2 / (1 - 0.8), pp, pp, rng0[root], max_plan_len=8)
        order, _, _ = bfs_order(tree["parent"], tree["first_child"], 5)
        oorder, _, _ = bfs_order(o["tree"]["parent"], o["tree"]["first_child"], 5)
        assert np.array_equal(tree["count"][order], o["tree"]["count"][oorder])
        assert np.array_equal(tree["value"][order], o["tree"]["value"][oorder])
    if policy is not None:
        policy.close()
    model.close()


def test_opd_8192_roots_budget5000_sample_vs_oracle(ctx):
    """BASELINE C4 on one GPU (bench.py --workload opd --roots 8192): the host picks the high-occupancy variant."""
    from oracle import oracle
    t, r, term, model = headline_model(ctx)
    n, budget = 8192, 5000
    s0 = bench_roots(term, n)
    rng = fast_rng_states(n, 2)
    rng0 = rng.copy()
    out = ctx.opd_plan(model, s0, budget, 0.8, 0.0, rng, max_plan_len=32)
    assert (out["status"] == 0).all() and (out["env_steps"] == 5000).all()
    idx = np.sort(np.random.Generator(np.random.PCG64(8)).choice(n, size=SAMPLE, replace=False))
    idx[0], idx[-1] = 0, n - 1
    ref = oracle.opd_plan_batch(t, r, term, s0[idx], budget, 0.8, 0.0, rng0[idx], max_plan_len=32, n_threads=16)
    for k in ("plans", "plan_len", "env_steps", "status"):
        np.testing.assert_array_equal(out[k][idx], ref[k], err_msg=k)
    assert np.array_equal(out["root_lower"][idx], ref["root_lower"])
    assert np.array_equal(out["root_upper"][idx], ref["root_upper"])
    np.testing.assert_array_equal(rng[idx], ref["rng_after"])
    model.close()


def test_opd_1024_roots_budget5000_all_vs_oracle(ctx):
    """The 1024-root shard of C4 (bench.py --workload opd default): LDS bounds with the parent map in HBM."""
    from oracle import oracle
    t, r, term, model = headline_model(ctx)
    n, budget = 1024, 5000
    s0 = bench_roots(term, n)
    rng = fast_rng_states(n, 3)
    rng0 = rng.copy()
    out = ctx.opd_plan(model, s0, budget, 0.8, 0.0, rng, max_plan_len=32)
    ref = oracle.opd_plan_batch(t, r, term, s0, budget, 0.8, 0.0, rng0, max_plan_len=32, n_threads=16)
    for k in ("plans", "plan_len", "env_steps", "status"):
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    assert np.array_equal(out["root_lower"], ref["root_lower"]) and np.array_equal(out["root_upper"], ref["root_upper"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()


def test_saopd_16384_planners_sample_vs_oracle(ctx):
    """bench.py --workload saopd: the reference's GridWorld configuration (budget 500, gamma 0.8), first plan of
    16 384 fresh planners."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.gridworld()
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    n, budget = 16384, 500
    s0 = np.random.Generator(np.random.PCG64(12345)).integers(0, 100, size=n).astype(np.int32)
    rng = fast_rng_states(n, 4)
    rng0 = rng.copy()
    planners = native.StateAwarePlanners(ctx, model, n)
    out = planners.plan(s0, budget, 0.8, 0.0, rng, max_plan_len=8)
    assert (out["status"] == 0).all()
    idx = np.sort(np.random.Generator(np.random.PCG64(9)).choice(n, size=SAMPLE, replace=False))
    idx[0], idx[-1] = 0, n - 1
    ref = oracle.saopd_plan_batch(t, r, term, s0[idx], budget, 0.8, rng_states=rng0[idx], max_plan_len=8, n_threads=16)
    for k in ("plans", "plan_len", "env_steps", "updates", "status"):
        np.testing.assert_array_equal(out[k][idx], ref[k], err_msg=k)
    np.testing.assert_array_equal(rng[idx], ref["rng_after"])
    planners.close()
    model.close()


def test_vi_dense_S10000_three_sweeps_vs_oracle(ctx):
    """C2-dense (bench.py --workload vi_dense): S = 10 000, |A| = 5 ALWAYS (4.0 GB of transitions, three V chunks of
    4096 columns per row; VERDICT r2: no fallback to |A| = 2).  The model is built on the device and borrowed by the
    library; the oracle -- numpy's pairwise order -- never needs the whole array: a dense backup is independent per
    source row, so it replays the three sweeps one 500-row block (200 MB) at a time.  Matrix-core form: Q within 1e-12,
    identical greedy actions, same sweep count; default form (numpy's order): bit for bit."""
    import torch
    from oracle import oracle
    s, a, gamma, block = 10000, 5, 0.95, 500
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    t = torch.rand((s, a, s), dtype=torch.float64, device=dev, generator=g)
    t /= t.sum(-1, keepdim=True)
    host = np.random.Generator(np.random.PCG64(0))
    r = host.random((s, a))
    term = host.random(s) < 0.02
    d_r = torch.from_numpy(r).to(dev)
    d_term = torch.from_numpy(term.astype(np.uint8)).to(dev)
    torch.cuda.synchronize()
    model = ctx.load_dense(t, d_r, d_term)
    ctx.vi_dense_mode("mfma")
    try:
        q, sweeps = ctx.vi_solve(model, gamma, 3)
    finally:
        ctx.vi_dense_mode("exact")
    assert (model.S, model.A) == (s, a) and sweeps == 3
    # the reference's fixed_point_iteration (value_iteration.py:65-73) on the same numbers, block by block
    q_ref = np.zeros((s, a))
    for _ in range(3):
        v = q_ref.max(axis=1)
        q_next = np.empty_like(q_ref)
        for lo in range(0, s, block):
            rows = t[lo:lo + block].cpu().numpy()
            q_next[lo:lo + block] = oracle.dense_backup_rows(rows, r[lo:lo + block], term[lo:lo + block], v, gamma)
        assert not np.allclose(q_ref, q_next, rtol=1e-5, atol=1e-8)      # no early exit within three sweeps
        q_ref = q_next
    np.testing.assert_allclose(q, q_ref, rtol=1e-12, atol=1e-12)
    assert np.array_equal(q.argmax(axis=1), q_ref.argmax(axis=1))
    # the same three sweeps in numpy's own order of additions (vi_dense_exact_q; rows of 10 000 = a piece of 8192 and one
    # of 1808 elements, see tests/test_oracle_vi_long_rows.py): bit for bit
    q_x, sweeps_x = ctx.vi_solve(model, gamma, 3)          # (the default form)
    assert sweeps_x == 3 and np.array_equal(q_x, q_ref)
    model.close()
