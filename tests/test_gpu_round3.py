"""Round-3 additions to the C-ABI parity tests (one topic per test; see the docstrings)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("robust", [False, True])
def test_persistent_vi_gives_up_and_the_call_solves_again(ctx, robust, monkeypatch):
    """vi_det_persist needs its grid co-resident; when it gives up (raised timeout word -- injected
    here, on a shared GPU the bounded spins end in it) a host-mode mp_vi_solve must not hand NaN / sweeps = -1 to the
    caller: it solves again on the chained launches.  Same Q and sweep count as the oracle either way."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(10, 10, 100, seed=0)                  # C2: S = 10 000 -> the persistent kernel
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    if robust:
        cfg2 = generators.rewire(cfg, 0.1, seed=4)
        t, r, term = np.stack([t, cfg2["transition"]]), np.stack([r, cfg2["reward"] * 0.9]), None
    model = ctx.load_table(t, r, term)
    q_ref, sweeps_ref = oracle.vi_solve("deterministic", t, r, term, gamma=0.95, iterations=200, robust=robust)
    q, sweeps = ctx.vi_solve(model, 0.95, 200, robust=robust)
    assert ctx.last_kernel_ms()[1] == 1, "C2 is expected to run on the single persistent launch"
    assert sweeps == sweeps_ref and np.array_equal(q, q_ref)
    monkeypatch.setenv("MP_VI_PERSIST_INJECT_TIMEOUT", "1")
    q2, sweeps2 = ctx.vi_solve(model, 0.95, 200, robust=robust)
    assert ctx.last_kernel_ms()[1] > 1, "the fallback runs the chained launches"
    assert sweeps2 == sweeps_ref and np.array_equal(q2, q_ref)
    v = ctx.vi_solve_v(model, 0.95, 200) if not robust else None
    if v is not None:
        assert np.array_equal(v, oracle.vi_solve("deterministic", t, r, term, gamma=0.95, iterations=200, state_value=True))
    # device mode is asynchronous: the failure is REPORTED (sweeps = -1), and the Python helper raises on it
    import torch
    from rl_agents_amd import native
    d_q = torch.zeros((model.S, model.A), dtype=torch.float64, device="cuda")
    d_sw = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.vi_solve_device(model, 0.95, 200, d_q, d_sw, robust=robust)
    ctx.synchronize()
    with pytest.raises(native.NativeError):
        native.check_device_sweeps(d_sw)
    monkeypatch.delenv("MP_VI_PERSIST_INJECT_TIMEOUT")
    ctx.vi_solve_device(model, 0.95, 200, d_q, d_sw, robust=robust)
    ctx.synchronize()
    assert native.check_device_sweeps(d_sw) == sweeps_ref and np.array_equal(d_q.cpu().numpy(), q_ref)
    model.close()


def _pipe_case(ctx, n, seed):
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.highway_shaped(4, 5, 12, seed=seed)
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    model = ctx.load_table(t, r, term)
    roots = np.random.Generator(np.random.PCG64(seed)).choice(np.flatnonzero(~term), size=n).astype(np.int32)
    rng0 = native.seed_sequence_states([99], 0, n)
    return (t, r, term), model, roots, rng0


@pytest.mark.parametrize("n,chunk,streams", [(140000, None, None), (5000, 1024, 3), (2049, 1024, 8), (3072, 1024, 1)])
def test_pipelined_host_plan_equals_the_single_launch(ctx, n, chunk, streams, monkeypatch):
    """VERDICT r2 task 3: mp_uct_plan with host arrays pipelines chunks of roots (H2D -> kernel -> D2H) over side streams.
    A chunk is the same launch on shifted pointers, so everything must be bit-identical to the one-launch call: plans,
    values, counts, env steps, generator records, and the trees left on the device (roots from different chunks, incl.
    the ragged last one) -- with pageable numpy arrays, and with pinned buffers + device-resident generator records.
    A sample is also replayed by the oracle."""
    from oracle import oracle
    from rl_agents_amd import native
    (t, r, term), model, roots, rng0 = _pipe_case(ctx, n, 11)
    p = np.ones(model.A) / model.A
    args = (6, 7, 0.8, 10.0, p, p)
    monkeypatch.setenv("MP_PIPE_CHUNK", "0")
    rng_a = rng0.copy()
    ref = ctx.uct_plan(model, roots, *args, rng_a, max_plan_len=7)
    assert ctx.last_kernel_ms()[1] == 1
    probe = sorted({0, 63, 64, 1023, 1024, n // 2, n - 1})
    trees_ref = [ctx.uct_tree(i) for i in probe]
    if chunk is None:
        monkeypatch.delenv("MP_PIPE_CHUNK")
    else:
        monkeypatch.setenv("MP_PIPE_CHUNK", str(chunk))
        monkeypatch.setenv("MP_PIPE_STREAMS", str(streams))
    rng_b = rng0.copy()
    out = ctx.uct_plan(model, roots, *args, rng_b, max_plan_len=7)
    assert ctx.last_kernel_ms()[1] == -(-n // (chunk or 65536)), "the call is expected to run chunked"
    for k in ref:
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(rng_b, rng_a)
    for i, tr in zip(probe, trees_ref):
        got = ctx.uct_tree(i)
        for k in tr:
            np.testing.assert_array_equal(got[k], tr[k], err_msg="tree of root {} / {}".format(i, k))
    # pinned result buffers (a subset of the outputs) + generator records resident on the device
    bufs = ctx.plan_buffers(n, 7, outputs=("plans", "plan_len", "env_steps"))
    bufs["root_state"][:] = roots
    dev_rng = ctx.device_rng(rng0)
    out2 = ctx.uct_plan(model, bufs["root_state"], *args, dev_rng, out=bufs)
    # mp_host_alloc arrays: read / written in place up to 65 536 roots (one launch, no copies), two pipelined chunks beyond
    want = 1 if n <= 65536 else (2 if chunk is None else -(-n // chunk))
    assert ctx.last_kernel_ms()[1] == want, "launches for pinned arrays"
    assert set(out2) == {"root_state", "plans", "plan_len", "env_steps"}
    for k in ("plans", "plan_len", "env_steps"):
        np.testing.assert_array_equal(out2[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(dev_rng.get(), rng_a)
    # second plan continues the resident stream exactly like the host records
    rng_c = rng_a.copy()
    ref2 = ctx.uct_plan(model, roots, *args, rng_c, max_plan_len=7)
    out3 = ctx.uct_plan(model, bufs["root_state"], *args, dev_rng, out=bufs)
    np.testing.assert_array_equal(out3["plans"], ref2["plans"])
    np.testing.assert_array_equal(dev_rng.get(first=n - 5), rng_c[n - 5:])
    # the same pinned arrays through the copy path (zero-copy switched off), and pinned generator records on the host side
    monkeypatch.setenv("MP_NO_ZERO_COPY", "1")
    dev_rng.set(rng0)
    out4 = ctx.uct_plan(model, bufs["root_state"], *args, dev_rng, out=bufs)
    np.testing.assert_array_equal(out4["plans"], ref["plans"])
    monkeypatch.delenv("MP_NO_ZERO_COPY")
    pin = ctx.pinned(dict(rng=((n, 6), np.uint64), root_value=((n,), np.float64), counts=((n, model.A), np.int64)))
    pin["rng"][:] = rng0
    out5 = ctx.uct_plan(model, bufs["root_state"], *args, pin["rng"], out=dict(root_value=pin["root_value"],
                                                                              root_child_count=pin["counts"]))
    np.testing.assert_array_equal(out5["root_value"], ref["root_value"])
    np.testing.assert_array_equal(out5["root_child_count"], ref["root_child_count"])
    np.testing.assert_array_equal(pin["rng"], rng_a)
    pin.close()
    idx = np.random.Generator(np.random.PCG64(1)).choice(n, size=min(n, 512), replace=False)
    chk = oracle.uct_plan_batch(t, r, term, roots[idx], *args, rng0[idx].copy(), max_plan_len=7)
    np.testing.assert_array_equal(ref["plans"][idx], chk["plans"])
    np.testing.assert_array_equal(ref["env_steps"][idx], chk["env_steps"])
    dev_rng.close()
    bufs.close()
    model.close()


def test_pipelined_plan_continues_kept_trees(ctx, monkeypatch):
    """step_strategy 'subtree' across pipelined calls: plan, re-root, plan again -- chunked equals unchunked."""
    from rl_agents_amd import native
    (t, r, term), model, roots, rng0 = _pipe_case(ctx, 3000, 12)
    p = np.ones(model.A) / model.A
    results = []
    for chunk in ("0", "1024"):
        monkeypatch.setenv("MP_PIPE_CHUNK", chunk)
        ctx.uct_reset_tree()
        rng = rng0.copy()
        a = ctx.uct_plan(model, roots, 8, 6, 0.8, 10.0, p, p, rng, max_plan_len=6)
        first = np.where(a["plan_len"] > 0, a["plans"][:, 0], 0).astype(np.int32)
        ctx.uct_step_tree(first)
        nxt = t[roots, first].astype(np.int32)
        b = ctx.uct_plan(model, nxt, 8, 6, 0.8, 10.0, p, p, rng, max_plan_len=6)
        results.append((a, b, rng, ctx.uct_tree(2999), ctx.uct_tree(1024)))
    for x, y in zip(results[0][:2], results[1][:2]):
        for k in x:
            np.testing.assert_array_equal(x[k], y[k], err_msg=k)
    np.testing.assert_array_equal(results[0][2], results[1][2])
    for i in (3, 4):
        for k in results[0][i]:
            np.testing.assert_array_equal(results[0][i][k], results[1][i][k])
    ctx.uct_reset_tree()
    model.close()


def test_device_rng_with_the_other_planners(ctx):
    """MP_MEM_RNG_DEVICE on mp_opd_plan / mp_ropd_plan / mp_saopd_plan: host arrays + resident generator records give the
    host-record results (the tie-breaks of get_plan draw from the records)."""
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.random_deterministic(40, 3, seed=5)
    t, r, term = cfg["transition"], np.round(cfg["reward"] * 2) / 2, cfg["terminal"]      # coarse rewards: many ties
    model = ctx.load_table(t, r, term)
    n = 130
    roots = (np.arange(n) % 40).astype(np.int32)
    rng0 = native.seed_sequence_states([5], 0, n)
    rng_h = rng0.copy()
    ref = ctx.opd_plan(model, roots, 60, 0.7, 0.0, rng_h, max_plan_len=24)
    assert not np.array_equal(rng_h, rng0), "the case is meant to draw tie-breaks"
    dev = ctx.device_rng(rng0)
    out = dict(plans=np.full((n, 24), -1, np.int32), plan_len=np.zeros(n, np.int32), root_lower=np.zeros(n), root_upper=np.zeros(n),
               env_steps=np.zeros(n, np.int64), status=np.zeros(n, np.int32))
    native._check(ctx._lib.mp_opd_plan(ctx._h, model._h, n, native._ptr(roots), 60, 0.7, 0.0, dev.ptr(), 24,
                                       native._ptr(out["plans"]), native._ptr(out["plan_len"]), native._ptr(out["root_lower"]),
                                       native._ptr(out["root_upper"]), native._ptr(out["env_steps"]), native._ptr(out["status"]),
                                       native.MP_MEM_HOST | native.MP_MEM_RNG_DEVICE))
    for k in out:
        np.testing.assert_array_equal(out[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(dev.get(), rng_h)
    with pytest.raises(native.NativeError):
        native._check(ctx._lib.mp_opd_plan(ctx._h, model._h, n, native._ptr(roots), 60, 0.7, 0.0, dev.ptr(), 24, None, None,
                                           None, None, None, None, 7))
    # zero-copy: every array in mp_host_alloc memory -> the kernel works on the caller's arrays (OPD through the shared
    # staging helpers; also the state-aware planners)
    pin = ctx.pinned(dict(roots=((n,), np.int32), rng=((n, 6), np.uint64), plans=((n, 24), np.int32), plan_len=((n,), np.int32),
                          root_lower=((n,), np.float64), root_upper=((n,), np.float64), env_steps=((n,), np.int64),
                          status=((n,), np.int32)))
    pin["roots"][:] = roots
    pin["rng"][:] = rng0
    native._check(ctx._lib.mp_opd_plan(ctx._h, model._h, n, native._ptr(pin["roots"]), 60, 0.7, 0.0, native._ptr(pin["rng"]), 24,
                                       native._ptr(pin["plans"]), native._ptr(pin["plan_len"]), native._ptr(pin["root_lower"]),
                                       native._ptr(pin["root_upper"]), native._ptr(pin["env_steps"]), native._ptr(pin["status"]),
                                       native.MP_MEM_HOST))
    for k in out:
        np.testing.assert_array_equal(pin[k], ref[k], err_msg="zero-copy " + k)
    np.testing.assert_array_equal(pin["rng"], rng_h)
    pin.close()
    dev.close()
    model.close()


def test_planner_blocks_are_recycled_and_survive_their_context():
    """The device blocks of a planner batch go back to the ctx's block cache at close() and serve the next batch of the same
    shape (same results, whatever the recycled memory holds); a batch that outlives its context is still freed cleanly."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.envs import generators
    cfg = generators.gridworld()
    t, r, term = cfg["transition"], cfg["reward"], cfg["terminal"]
    n = 130
    s0 = (np.arange(n) * 7 % r.shape[0]).astype(np.int32)
    rng0 = np.zeros((n, 6), np.uint64)
    rng0[:, 0] = np.arange(n) + 11
    rng0[:, 1] = 999
    rng0[:, 3] = 1
    c = native.Context(0)
    model = c.load_table(t, r, term)
    outs = []
    for _ in range(3):                                  # the second and third batch run on recycled blocks
        pl = native.StateAwarePlanners(c, model, n)
        rng = rng0.copy()
        outs.append((pl.plan(s0, 200, 0.8, 0.0, rng, max_plan_len=12), rng))
        pl.close()
    for out, rng in outs[1:]:
        for k in ("plans", "plan_len", "env_steps", "updates", "status"):
            np.testing.assert_array_equal(out[k], outs[0][0][k], err_msg=k)
        np.testing.assert_array_equal(rng, outs[0][1])
    rng = rng0.copy()
    ref = oracle.saopd_plan_batch(t, r, term, s0, 200, 0.8, 0.0, rng, max_plan_len=12)
    np.testing.assert_array_equal(outs[0][0]["plans"], ref["plans"])
    np.testing.assert_array_equal(outs[0][0]["updates"], ref["updates"])
    late = native.StateAwarePlanners(c, model, n)       # outlives the context
    model.close()
    c.close()
    late.close()
