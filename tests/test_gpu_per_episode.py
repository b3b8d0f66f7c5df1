"""One table per episode, replaced at every step (the real highway-v0-as-finite-mdp batch): batch models
(mp_model_load_table_batch / mp_model_update_tables / mp_model_update_rows), N value-iteration agents in one launch
(mp_vi_solve_batch), one MDP per root for UCT and OPD (mp_uct_plan_models / mp_opd_plan_models) -- against the golden of the
unmodified reference (tests/golden/per_episode.npz) and against N sequential oracle solves / plans on distinct seeded
tables.  Reference: value_iteration.py:29-35 (re-extraction on every act), trainer/evaluation.py:139-194 (one env per process)."""
import numpy as np
import pytest

from rl_agents_amd import native

pytestmark = pytest.mark.gpu

E, T_STEPS = 6, 3


@pytest.fixture(scope="module")
def ctx():
    from rl_agents_amd import native
    c = native.Context(0)
    yield c
    c.close()


def _rng_states(n, base=0):
    from rl_agents_amd import native
    return np.stack([native.rng_state_from_generator(
        np.random.Generator(np.random.PCG64(np.random.SeedSequence(base + i)))) for i in range(n)])


def _tables(n, shape=(3, 4, 10), seed0=0):
    from rl_agents_amd.envs import generators
    cfgs = [generators.highway_shaped(*shape, collision_rate=0.03 + 0.02 * (i % 5), seed=seed0 + i) for i in range(n)]
    return (np.stack([c["transition"] for c in cfgs]), np.stack([c["reward"] for c in cfgs]),
            np.stack([c["terminal"] for c in cfgs]))


# ---------------------------------------------------------------------------------------------- goldens of the reference
def test_golden_vi_batch_every_step(ctx, golden):
    """Six ValueIterationAgent objects of the reference, their tables replaced before every step: one batch model,
    mp_model_update_tables per step, one mp_vi_solve_batch per step."""
    z = golden["per_episode"]
    model = ctx.load_table_batch(z["transition"][:, 0], z["reward"][:, 0], z["terminal"][:, 0])
    for t in range(T_STEPS):
        if t > 0:
            model.update_tables(0, z["transition"][:, t], z["reward"][:, t], z["terminal"][:, t])
        q, sweeps = ctx.vi_solve_batch(model, float(z["vi/gamma"]), int(z["vi/iterations"]))
        for e in range(E):
            p = "vi/e{}/t{}".format(e, t)
            assert np.array_equal(q[e], z[p + "/Q"]), p
            assert int(sweeps[e]) == int(z[p + "/sweeps"]), p
            s = int(z["vi/e{}/states".format(e)][t])
            assert int(np.argmax(q[e][s])) == int(z[p + "/action"]), p
    model.close()


def test_golden_uct_per_episode_models(ctx, golden):
    z = golden["per_episode"]
    a = z["reward"].shape[-1]
    p_uniform = np.ones(a) / a
    model = ctx.load_table_batch(z["transition"][:, 0], z["reward"][:, 0], z["terminal"][:, 0])
    rng = np.stack([z["uct/e{}/rng_before".format(e)] for e in range(E)]).astype(np.uint64)
    total = np.zeros(E, np.int64)
    for t in range(T_STEPS):
        if t > 0:   # two episodes at a time: exercises first / count
            for first in range(0, E, 2):
                model.update_tables(first, z["transition"][first:first + 2, t], z["reward"][first:first + 2, t],
                                    z["terminal"][first:first + 2, t])
        s0 = np.array([int(z["uct/e{}/states".format(e)][t]) for e in range(E)], np.int32)
        out = ctx.uct_plan(model, s0, int(z["uct/episodes"]), int(z["uct/horizon"]), float(z["uct/gamma"]),
                           float(z["uct/temperature"]), p_uniform, p_uniform, rng, max_plan_len=int(z["uct/horizon"]),
                           model_index=np.arange(E))
        total += out["env_steps"]
        for e in range(E):
            p = "uct/e{}/t{}".format(e, t)
            np.testing.assert_array_equal(out["plans"][e, :out["plan_len"][e]], z[p + "/plan"], err_msg=p)
            np.testing.assert_array_equal(rng[e], z[p + "/rng_after"], err_msg=p)
            assert out["root_value"][e] == float(z[p + "/root_value"]), p
            assert int(total[e]) == int(z[p + "/env_steps_total"]), p
    model.close()


def test_golden_opd_per_episode_models(ctx, golden):
    z = golden["per_episode"]
    model = ctx.load_table_batch(z["transition"][:, 0], z["reward"][:, 0], z["terminal"][:, 0])
    rng = np.stack([z["opd/e{}/rng_before".format(e)] for e in range(E)]).astype(np.uint64)
    total = np.zeros(E, np.int64)
    for t in range(T_STEPS):
        if t > 0:
            model.update_tables(0, z["transition"][:, t], z["reward"][:, t], z["terminal"][:, t])
        s0 = np.array([int(z["opd/e{}/states".format(e)][t]) for e in range(E)], np.int32)
        out = ctx.opd_plan(model, s0, int(z["opd/budget"]), float(z["opd/gamma"]), 0.0, rng, max_plan_len=64,
                           model_index=np.arange(E))
        total += out["env_steps"]
        assert (out["status"] == 0).all()
        for e in range(E):
            p = "opd/e{}/t{}".format(e, t)
            np.testing.assert_array_equal(out["plans"][e, :out["plan_len"][e]], z[p + "/plan"], err_msg=p)
            np.testing.assert_array_equal(rng[e], z[p + "/rng_after"], err_msg=p)
            assert out["root_lower"][e] == float(z[p + "/root_lower"]) and out["root_upper"][e] == float(z[p + "/root_upper"]), p
            assert int(total[e]) == int(z[p + "/env_steps_total"]), p
    model.close()


# ---------------------------------------------------------------------------------------------- N sequential oracle solves
@pytest.mark.parametrize("shape,n,gamma,iters", [((3, 4, 10), 257, 0.95, 200), ((3, 4, 10), 33, 1.0, 100),
                                                 ((2, 3, 5), 64, 0.9, 50), ((5, 5, 20), 40, 0.95, 200),
                                                 ((6, 8, 30), 9, 0.99, 300), ((8, 8, 50), 5, 0.95, 60)])
def test_vi_batch_vs_sequential_oracle(ctx, shape, n, gamma, iters):
    """Every register-resident form (S = 30 .. 3200): Q and sweep counts of every MDP equal N sequential oracle solves."""
    from oracle import oracle
    tr, rw, tm = _tables(n, shape, seed0=7 * n)
    rw = rw * (10.0 ** -(2 * (np.arange(n) % 5)))[:, None, None]     # small rewards pass allclose (atol 1e-8) sooner
    model = ctx.load_table_batch(tr, rw, tm)
    q, sweeps = ctx.vi_solve_batch(model, gamma, iters)
    q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=gamma, iterations=iters)
    np.testing.assert_array_equal(sweeps, sw_ref)
    assert np.array_equal(q, q_ref)
    assert len(set(sweeps.tolist())) > 1 or n < 10, sweeps     # the MDPs of one launch do stop at different sweeps
    model.close()


@pytest.mark.parametrize("knob,variant", [(None, "vi_batch_wg_stream"), ("MP_VI_BATCH_NO_WGR", "vi_batch_wg_lds"),
                                          ("MP_VI_BATCH_NO_VLDS", "vi_batch_wg_global")])
def test_vi_batch_workgroup_forms(ctx, monkeypatch, knob, variant):
    """The one-workgroup-per-MDP forms (lane-major streamed tables + V double-buffered in LDS; state-major tables with
    V_{k-1}, V_k in LDS; three V buffers in global memory) at the C2 shape S = 10 000 and, forced, on small MDPs -- same Q
    and sweeps as sequential oracle solves."""
    from oracle import oracle
    monkeypatch.setenv("MP_VI_BATCH_CLUSTER", "0")                 # (few large MDPs would take the cluster form: next test)
    if knob:
        monkeypatch.setenv(knob, "1")
    tr, rw, tm = _tables(3, (10, 10, 100), seed0=50)
    rw = rw * np.array([1.0, 1e-3, 1e-6])[:, None, None]
    model = ctx.load_table_batch(tr, rw, tm)
    q, sweeps = ctx.vi_solve_batch(model, 0.95, 200)
    assert ctx.last_kernel_variant() == variant
    q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.95, iterations=200)
    np.testing.assert_array_equal(sweeps, sw_ref)
    assert np.array_equal(q, q_ref)
    model.close()
    monkeypatch.setenv("MP_VI_BATCH_NO_REG", "1")
    tr, rw, tm = _tables(21, (3, 4, 10), seed0=90)
    model = ctx.load_table_batch(tr, rw, tm)
    for iters in (0, 1, 2, 200):
        q, sweeps = ctx.vi_solve_batch(model, 0.95, iters)
        assert ctx.last_kernel_variant() == variant
        q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.95, iterations=iters)
        np.testing.assert_array_equal(sweeps, sw_ref)
        assert np.array_equal(q, q_ref), iters
    model.close()


@pytest.mark.parametrize("n,k", [(3, None), (5, 4), (64, None), (9, 8), (33, 4)])
def test_vi_batch_cluster_form(ctx, monkeypatch, n, k):
    """Round 6: K workgroups per MDP for few LARGE MDPs (S = 10 000: 64 MDPs x 4 workgroups fill the chip), V exchanged through
    write-through stores and a per-sweep counter that also carries the allclose vote -- Q and per-MDP sweep counts of
    sequential oracle solves, with MDPs of one launch stopping at different sweeps."""
    from oracle import oracle
    if k:
        monkeypatch.setenv("MP_VI_BATCH_CLUSTER", str(k))
    tr, rw, tm = _tables(n, (10, 10, 100), seed0=300 + n)
    rw = rw * (10.0 ** -(np.arange(n) % 4))[:, None, None]
    model = ctx.load_table_batch(tr, rw, tm)
    for iters in (1, 3, 200):
        q, sweeps = ctx.vi_solve_batch(model, 0.95, iters)
        assert ctx.last_kernel_variant() == "vi_batch_cluster{}".format(k or (8 if n <= 32 else 4))
        q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.95, iterations=iters)
        np.testing.assert_array_equal(sweeps, sw_ref)
        assert np.array_equal(q, q_ref), iters
    model.close()


def test_vi_batch_cluster_small_mdps_and_fallback(ctx, monkeypatch):
    """The cluster form forced on small MDPs (K = 2, 4, 8: slices of a few states, an empty slice at K = 8 x S = 30 is not), and
    clusters that never meet (test hook: one arrival too many is awaited): every MDP is solved again by the follow-up launch."""
    from oracle import oracle
    monkeypatch.setenv("MP_VI_BATCH_NO_REG", "1")
    for shape, n, k in (((3, 4, 10), 21, 2), ((3, 4, 10), 21, 4), ((2, 3, 5), 7, 8), ((5, 5, 20), 11, 4)):
        monkeypatch.setenv("MP_VI_BATCH_CLUSTER", str(k))
        tr, rw, tm = _tables(n, shape, seed0=90 + k)
        model = ctx.load_table_batch(tr, rw, tm)
        q, sweeps = ctx.vi_solve_batch(model, 0.95, 200)
        assert ctx.last_kernel_variant() == "vi_batch_cluster{}".format(k)
        q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.95, iterations=200)
        np.testing.assert_array_equal(sweeps, sw_ref)
        assert np.array_equal(q, q_ref), (shape, k)
        model.close()
    monkeypatch.setenv("MP_VI_BATCH_CLUSTER", "4")
    monkeypatch.setenv("MP_VI_BATCH_CLUSTER_NEVER_MEETS", "1")
    tr, rw, tm = _tables(6, (3, 4, 10), seed0=17)
    model = ctx.load_table_batch(tr, rw, tm)
    q, sweeps = ctx.vi_solve_batch(model, 0.95, 200)
    q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.95, iterations=200)
    np.testing.assert_array_equal(sweeps, sw_ref)
    assert np.array_equal(q, q_ref)
    model.close()


def test_vi_batch_any_action_count_and_no_terminals(ctx):
    """|A| without a compile-time form (7), tables without terminal flags, iterations 0 / 1."""
    from oracle import oracle
    g = np.random.Generator(np.random.PCG64(5))
    n, s, a = 12, 45, 7
    tr = g.integers(0, s, size=(n, s, a))
    rw = g.random((n, s, a))
    model = ctx.load_table_batch(tr, rw, None)
    for iters in (0, 1, 80):
        q, sweeps = ctx.vi_solve_batch(model, 0.9, iters)
        q_ref, sw_ref = oracle.vi_solve_each(tr, rw, None, gamma=0.9, iterations=iters)
        np.testing.assert_array_equal(sweeps, sw_ref)
        assert np.array_equal(q, q_ref), iters
    model.close()


def test_vi_solve_refuses_batch_model(ctx):
    from rl_agents_amd import native
    tr, rw, tm = _tables(4)
    model = ctx.load_table_batch(tr, rw, tm)
    with pytest.raises(native.NativeError):
        ctx.vi_solve(model, 0.9, 10)
    model.close()


@pytest.mark.parametrize("n_models,n_roots", [(64, 64), (500, 1500), (37, 4096)])
def test_uct_per_root_models_vs_oracle(ctx, n_models, n_roots):
    """Budget 1000 as 33 x 30 on distinct highway-shaped (3, 4, 10) tables, one per root (or shared by a few roots):
    plans, values, counts, env steps and generator states equal per-root oracle plans on the root's own table."""
    from oracle import oracle
    tr, rw, tm = _tables(n_models, seed0=300)
    model = ctx.load_table_batch(tr, rw, tm)
    g = np.random.Generator(np.random.PCG64(n_roots))
    mi = (np.arange(n_roots) % n_models).astype(np.int32) if n_models == n_roots else g.integers(0, n_models, n_roots).astype(np.int32)
    s0 = g.integers(0, tr.shape[1], n_roots).astype(np.int32)
    rng = _rng_states(n_roots, base=77)
    rng_ref = rng.copy()
    p = np.ones(5) / 5
    out = ctx.uct_plan(model, s0, 33, 30, 0.8, 10.0, p, p, rng, max_plan_len=30, model_index=mi)
    sample = np.arange(n_roots) if n_roots <= 1500 else g.choice(n_roots, 600, replace=False)
    ref = oracle.uct_plan_each(tr, rw, tm, mi[sample], s0[sample], 33, 30, 0.8, 10.0, p, p, rng_ref[sample], max_plan_len=30)
    np.testing.assert_array_equal(out["plans"][sample], ref["plans"])
    assert np.array_equal(out["root_value"][sample], ref["root_value"])
    np.testing.assert_array_equal(out["root_child_count"][sample], ref["root_child_count"])
    np.testing.assert_array_equal(out["env_steps"][sample], ref["env_steps"])
    np.testing.assert_array_equal(rng[sample], ref["rng_after"])
    # the same plan through global root states on the plain entry point
    rng2 = rng_ref.copy()
    out2 = ctx.uct_plan(model, mi * tr.shape[1] + s0, 33, 30, 0.8, 10.0, p, p, rng2, max_plan_len=30)
    np.testing.assert_array_equal(out2["plans"], out["plans"])
    np.testing.assert_array_equal(rng2, rng)
    model.close()


@pytest.mark.parametrize("shape,episodes,horizon,n_roots", [((3, 4, 10), 33, 30, 701), ((2, 3, 5), 9, 7, 130), ((4, 5, 12), 20, 63, 67), ((3, 4, 10), 12, 100, 33)])
def test_uct_per_root_models_each_kernel_equals_gather_kernel_and_oracle(ctx, monkeypatch, shape, episodes, horizon, n_roots):
    """Round 6: batch models plan on uct_row_kernel (four roots per wavefront, a DPP row each; MP_UCT_ROW=0: uct_lone_kernel<..,
    EACH>, a wavefront per root) with the root's own MDP staged in LDS (local uint16 next states + the rewards themselves) --
    same plans, statistics, env steps, trees and generator records as the one-lane-per-root gather kernel (MP_UCT_EACH=0) and
    as per-root oracle plans; batch sizes that leave spare rows in the last wavefront; rewards with more than 256 distinct values (no dictionary
    in this form), terminal root states, `done_rule = next`, a step limit."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    n_models = 37
    cfgs = [generators.highway_shaped(*shape, collision_rate=0.03 + 0.02 * (i % 5), seed=500 + i) for i in range(n_models)]
    tr = np.stack([c["transition"] for c in cfgs])
    g = np.random.Generator(np.random.PCG64(shape[0] * 1000 + n_roots))
    rw = np.stack([c["reward"] for c in cfgs]) * g.random((n_models,) + cfgs[0]["reward"].shape)   # S * A distinct values per MDP
    tm = np.stack([c["terminal"] for c in cfgs])
    mi = g.integers(0, n_models, n_roots).astype(np.int32)
    s0 = g.integers(0, tr.shape[1], n_roots).astype(np.int32)
    term_roots = np.flatnonzero(tm[mi, s0])
    assert len(term_roots) > 0                      # (terminal root states are part of the sample)
    p = np.array([0.1, 0.3, 0.2, 0.25, 0.15])
    for done_rule, max_steps in (("source", 0), ("next", 0), ("source", 11)):
        model = ctx.load_table_batch(tr, rw, tm, done_rule=done_rule, max_steps=max_steps)
        steps0 = g.integers(0, 8, n_roots).astype(np.int32) if max_steps else None
        outs = {}
        for name, env in (("row", {"MP_UCT_ROW": "1"}), ("lone", {"MP_UCT_ROW": "0"}), ("gather", {"MP_UCT_EACH": "0"})):
            monkeypatch.delenv("MP_UCT_ROW", raising=False)
            monkeypatch.delenv("MP_UCT_EACH", raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            rng = _rng_states(n_roots, base=31)
            out = ctx.uct_plan(model, s0, episodes, horizon, 0.8, 10.0, p, p, rng, max_plan_len=horizon, model_index=mi,
                               root_steps=steps0)
            outs[name] = (out, rng, ctx.last_kernel_variant(), ctx.uct_tree(n_roots // 2), ctx.uct_tree(int(term_roots[0])),
                          ctx.uct_tree(n_roots - 1))
        monkeypatch.delenv("MP_UCT_EACH", raising=False)
        monkeypatch.delenv("MP_UCT_ROW", raising=False)
        # (a wavefront per root draws a rollout's actions one per lane: horizons up to 63; beyond, the gather kernel stands in)
        assert [outs[k][2] for k in ("row", "lone", "gather")] == ["uct_row_each", "uct_lone_each" if horizon <= 63 else "uct_global", "uct_global"]
        for name in ("row", "lone"):
            for key in ("plans", "plan_len", "root_value", "root_child_count", "root_child_value", "env_steps"):
                np.testing.assert_array_equal(outs[name][0][key], outs["gather"][0][key], err_msg=name + " " + key)
            np.testing.assert_array_equal(outs[name][1], outs["gather"][1], err_msg=name + " rng")
            for j in (3, 4, 5):
                for key in ("count", "value", "first_child"):
                    np.testing.assert_array_equal(outs[name][j][key], outs["gather"][j][key], err_msg=name + " tree " + key)
        outs["1"] = outs["row"]
        rng_ref = _rng_states(n_roots, base=31)
        ref = oracle.uct_plan_each(tr, rw, tm, mi, s0, episodes, horizon, 0.8, 10.0, p, p, rng_ref, max_plan_len=horizon,
                                   done_rule=done_rule, max_steps=max_steps, steps0=steps0)
        np.testing.assert_array_equal(outs["1"][0]["plans"], ref["plans"])
        assert np.array_equal(outs["1"][0]["root_value"], ref["root_value"])
        np.testing.assert_array_equal(outs["1"][0]["env_steps"], ref["env_steps"])
        np.testing.assert_array_equal(outs["1"][1], ref["rng_after"])
        model.close()


def test_opd_per_root_models_vs_oracle(ctx):
    from oracle import oracle
    n = 300
    tr, rw, tm = _tables(n, seed0=900)
    model = ctx.load_table_batch(tr, rw, tm)
    g = np.random.Generator(np.random.PCG64(3))
    s0 = g.integers(0, tr.shape[1], n).astype(np.int32)
    rng = _rng_states(n, base=5)
    rng_ref = rng.copy()
    out = ctx.opd_plan(model, s0, 500, 0.8, 0.0, rng, max_plan_len=64, model_index=np.arange(n))
    ref = oracle.opd_plan_each(tr, rw, tm, np.arange(n), s0, 500, 0.8, 0.0, rng_ref, max_plan_len=64)
    np.testing.assert_array_equal(out["plans"], ref["plans"])
    assert np.array_equal(out["root_lower"], ref["root_lower"]) and np.array_equal(out["root_upper"], ref["root_upper"])
    np.testing.assert_array_equal(out["env_steps"], ref["env_steps"])
    np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()


def test_per_root_models_device_arrays(ctx):
    """model_index / local root states as device tensors (mem = MP_MEM_DEVICE: the index arithmetic is one small launch)."""
    import torch
    from oracle import oracle
    n = 200
    tr, rw, tm = _tables(n, seed0=40)
    model = ctx.load_table_batch(tr, rw, tm)
    g = np.random.Generator(np.random.PCG64(8))
    s0 = g.integers(0, tr.shape[1], n).astype(np.int32)
    rng = _rng_states(n, base=9)
    dev = torch.device("cuda", ctx.device)
    d = dict(mi=torch.arange(n, dtype=torch.int32, device=dev), s0=torch.from_numpy(s0).to(dev),
             rng=torch.from_numpy(rng.view(np.int64)).to(dev), plans=torch.full((n, 8), -1, dtype=torch.int32, device=dev),
             plan_len=torch.zeros(n, dtype=torch.int32, device=dev), value=torch.zeros(n, dtype=torch.float64, device=dev),
             steps=torch.zeros(n, dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    p = np.ones(5) / 5
    ctx.uct_plan_device(model, n, d["s0"], 25, 8, 0.8, 10.0, p, p, d["rng"], 8, plans=d["plans"], plan_len=d["plan_len"],
                        root_value=d["value"], env_steps=d["steps"], model_index=d["mi"])
    ctx.synchronize()
    ref = oracle.uct_plan_each(tr, rw, tm, np.arange(n), s0, 25, 8, 0.8, 10.0, p, p, rng, max_plan_len=8)
    np.testing.assert_array_equal(d["plans"].cpu().numpy(), ref["plans"])
    assert np.array_equal(d["value"].cpu().numpy(), ref["root_value"])
    np.testing.assert_array_equal(d["rng"].cpu().numpy().view(np.uint64), ref["rng_after"])
    model.close()


def test_per_root_models_device_arrays_out_of_range_roots_are_reported(ctx):
    """device arrays cannot be validated on the host -- a root naming a model / state out of range is clamped to
    state 0 of model 0 (nothing is read out of bounds), counted, and the next synchronize() raises once; the other roots'
    results are what they would have been."""
    import torch
    from oracle import oracle
    n = 64
    tr, rw, tm = _tables(n, seed0=70)
    model = ctx.load_table_batch(tr, rw, tm)
    g = np.random.Generator(np.random.PCG64(3))
    s0 = g.integers(0, tr.shape[1], n).astype(np.int32)
    mi = np.arange(n, dtype=np.int32)
    bad_s0, bad_mi = s0.copy(), mi.copy()
    bad_mi[5], bad_mi[9], bad_s0[17], bad_s0[20] = n, -1, tr.shape[1], -3
    rng = _rng_states(n, base=4)
    dev = torch.device("cuda", ctx.device)
    d = dict(mi=torch.from_numpy(bad_mi).to(dev), s0=torch.from_numpy(bad_s0).to(dev),
             rng=torch.from_numpy(rng.view(np.int64)).to(dev), plans=torch.full((n, 8), -1, dtype=torch.int32, device=dev),
             plan_len=torch.zeros(n, dtype=torch.int32, device=dev), value=torch.zeros(n, dtype=torch.float64, device=dev),
             steps=torch.zeros(n, dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    p = np.ones(5) / 5
    ctx.uct_plan_device(model, n, d["s0"], 25, 8, 0.8, 10.0, p, p, d["rng"], 8, plans=d["plans"], plan_len=d["plan_len"],
                        root_value=d["value"], env_steps=d["steps"], model_index=d["mi"])
    torch.cuda.synchronize()
    assert ctx.device_faults() == 4
    with pytest.raises(native.NativeError, match="out of range"):
        ctx.synchronize()
    ctx.synchronize()                                   # reported once
    assert ctx.device_faults() == 0
    fixed_mi, fixed_s0 = mi.copy(), s0.copy()
    for i in (5, 9, 17, 20):
        fixed_mi[i], fixed_s0[i] = 0, 0                 # what the clamped roots planned on
    ref = oracle.uct_plan_each(tr, rw, tm, fixed_mi, fixed_s0, 25, 8, 0.8, 10.0, p, p, rng, max_plan_len=8)
    np.testing.assert_array_equal(d["plans"].cpu().numpy(), ref["plans"])
    assert np.array_equal(d["value"].cpu().numpy(), ref["root_value"])
    model.close()


@pytest.mark.parametrize("case", ["plain", "listed", "ordered"])
def test_policy_fused_on_the_device_equals_the_host_build(ctx, monkeypatch, case):
    """mp_policy_load_rows (round 6): the per-state policy -- prior rows, sampling thresholds, listed masks, fused records --
    built by kernels is the policy the host loops build: same plans, statistics and generator records on a single model
    (S = 3000) and on a batch model, where [S_each, A] rows tiled on the device equal the [N * S_each, A] tables tiled by numpy."""
    from rl_agents_amd.envs import generators
    g = np.random.Generator(np.random.PCG64(11))

    def tables(s, a):
        prior = g.random((s, a)) + 0.05
        listed, slots = None, None
        if case != "plain":
            listed = g.random((s, a)) < 0.7
            listed[np.arange(s), g.integers(0, a, s)] = True
            prior = np.where(listed, prior, 0.0)
        prior /= prior.sum(axis=1, keepdims=True)
        rollout = prior ** 2 / (prior ** 2).sum(axis=1, keepdims=True)
        if case == "ordered":       # the rollout policy lists the columns in another order (zero-probability columns last)
            slots = np.zeros((s, a), np.uint8)
            for i in range(s):
                on, off = np.flatnonzero(listed[i]), np.flatnonzero(~listed[i])
                slots[i] = np.concatenate([g.permutation(on), off])
        return prior, rollout, listed, slots

    def plan(model, roots, pol_args, mi=None):
        outs = []
        for how in ("host", "device"):
            monkeypatch.setenv("MP_POLICY_BUILD", how)
            policy = ctx.load_policy(model, *pol_args[how])
            rng = _rng_states(len(roots), base=3)
            out = ctx.uct_plan(model, roots, 20, 12, 0.8, 7.0, None, None, rng, max_plan_len=12, policy=policy, model_index=mi)
            outs.append((out, rng))
            policy.close()
        monkeypatch.delenv("MP_POLICY_BUILD")
        for key in ("plans", "plan_len", "root_value", "root_child_count", "root_child_value", "env_steps"):
            np.testing.assert_array_equal(outs[0][0][key], outs[1][0][key], err_msg=key)
        np.testing.assert_array_equal(outs[0][1], outs[1][1])

    cfg = generators.highway_shaped(5, 6, 100, seed=2)                # S = 3000
    model = ctx.load_table(cfg["transition"], cfg["reward"], cfg["terminal"])
    pr, ro, li, sl = tables(3000, 5)
    roots = g.integers(0, 3000, 500).astype(np.int32)
    plan(model, roots, dict(host=(pr, ro, li, sl), device=(pr, ro, li, sl)))
    model.close()
    n = 40
    tr, rw, tm = _tables(n, seed0=20)
    s_each = tr.shape[1]
    model = ctx.load_table_batch(tr, rw, tm)
    pr, ro, li, sl = tables(s_each, 5)
    tile = lambda x: None if x is None else np.tile(x, (n, 1))          # noqa: E731
    roots = g.integers(0, s_each, 300).astype(np.int32)
    mi = g.integers(0, n, 300).astype(np.int32)
    plan(model, roots, dict(host=(tile(pr), tile(ro), tile(li), tile(sl)), device=(pr, ro, li, sl)), mi=mi)
    with pytest.raises(native.NativeError):
        bad = pr.copy()
        bad[3, 1] = -0.5
        ctx.load_policy(model, bad, ro, li, sl)
    model.close()


# ---------------------------------------------------------------------------------------------- delta uploads
def test_update_rows_single_model_matches_reload(ctx):
    """mp_model_update_rows on a single table model: after a delta upload the model plans and solves
    exactly like a model loaded from the changed tables -- with and without terminal-flag changes, with new reward
    values (the compact LDS-resident form follows or is dropped), at S = 10 000 (LDS-resident UCT) and S = 120."""
    from oracle import oracle
    from rl_agents_amd.envs import generators
    for shape, n_roots in (((10, 10, 100), 70000), ((3, 4, 10), 256)):
        cfg = generators.highway_shaped(*shape, seed=1)
        t, r, term = cfg["transition"].copy(), cfg["reward"].copy(), cfg["terminal"].copy()
        s, a = r.shape
        model = ctx.load_table(t, r, term)
        g = np.random.Generator(np.random.PCG64(shape[0]))
        for rnd, with_term in enumerate((False, True, False)):
            rows = g.choice(s, size=max(3, s // 50), replace=False).astype(np.int32)
            t[rows] = g.integers(0, s, size=(len(rows), a))
            r[rows] = g.choice(np.unique(cfg["reward"]), size=(len(rows), a)) if rnd < 2 else g.random((len(rows), a))
            if with_term:
                term[rows] = g.random(len(rows)) < 0.3
            model.update_rows(rows, t[rows], r[rows], term[rows] if with_term else None)
            q, sweeps = ctx.vi_solve(model, 0.95, 50)
            q_ref, sw_ref = oracle.vi_solve("deterministic", t, r, term, gamma=0.95, iterations=50)
            assert sweeps == sw_ref and np.array_equal(q, q_ref), (shape, rnd)
            s0 = g.integers(0, s, n_roots).astype(np.int32)
            rng = _rng_states(n_roots, base=rnd)
            rng_ref = rng.copy()
            p = np.ones(a) / a
            out = ctx.uct_plan(model, s0, 12, 10, 0.8, 10.0, p, p, rng, max_plan_len=10)
            variant = ctx.last_kernel_variant()
            if n_roots >= 65536:
                assert variant == ("uct_ldsr" if rnd < 2 else "uct_global"), variant     # > 256 distinct rewards: gather kernel
            sample = g.choice(n_roots, 200, replace=False)
            ref = oracle.uct_plan_batch(t, r, term, s0[sample], 12, 10, 0.8, 10.0, p, p, rng_ref[sample], max_plan_len=10)
            np.testing.assert_array_equal(out["plans"][sample], ref["plans"])
            assert np.array_equal(out["root_value"][sample], ref["root_value"])
            np.testing.assert_array_equal(rng[sample], ref["rng_after"])
            rng_o = _rng_states(64, base=3)
            oo = ctx.opd_plan(model, s0[:64], 200, 0.8, 0.0, rng_o.copy(), max_plan_len=48)
            orf = oracle.opd_plan_batch(t, r, term, s0[:64], 200, 0.8, 0.0, rng_o.copy(), max_plan_len=48)
            np.testing.assert_array_equal(oo["plans"], orf["plans"])
            assert np.array_equal(oo["root_upper"], orf["root_upper"])
        model.close()


def test_update_rows_batch_model(ctx):
    from oracle import oracle
    n = 40
    tr, rw, tm = _tables(n, seed0=11)
    s, a = tr.shape[1:]
    model = ctx.load_table_batch(tr, rw, tm)
    g = np.random.Generator(np.random.PCG64(1))
    for with_term in (False, True):
        rows = np.sort(g.choice(n * s, size=90, replace=False)).astype(np.int32)
        b, loc = rows // s, rows % s
        tr[b, loc] = g.integers(0, s, size=(len(rows), a))
        rw[b, loc] = g.random((len(rows), a))
        if with_term:
            tm[b, loc] = g.random(len(rows)) < 0.4
        model.update_rows(rows, tr[b, loc], rw[b, loc], tm[b, loc] if with_term else None)
        q, sweeps = ctx.vi_solve_batch(model, 0.9, 120)
        q_ref, sw_ref = oracle.vi_solve_each(tr, rw, tm, gamma=0.9, iterations=120)
        np.testing.assert_array_equal(sweeps, sw_ref)
        assert np.array_equal(q, q_ref)
        s0 = g.integers(0, s, n).astype(np.int32)
        rng = _rng_states(n, base=21)
        rng_ref = rng.copy()
        p = np.ones(a) / a
        out = ctx.uct_plan(model, s0, 20, 12, 0.8, 10.0, p, p, rng, max_plan_len=12, model_index=np.arange(n))
        ref = oracle.uct_plan_each(tr, rw, tm, np.arange(n), s0, 20, 12, 0.8, 10.0, p, p, rng_ref, max_plan_len=12)
        np.testing.assert_array_equal(out["plans"], ref["plans"])
        np.testing.assert_array_equal(rng, ref["rng_after"])
    model.close()


def test_batch_model_argument_errors(ctx):
    from rl_agents_amd import native
    tr, rw, tm = _tables(3)
    bad = tr.copy()
    bad[1, 5, 2] = tr.shape[1]
    with pytest.raises(native.NativeError):
        ctx.load_table_batch(bad, rw, tm)                                        # a LOCAL index out of range
    model = ctx.load_table_batch(tr, rw, tm)
    assert (model.n_models, model.S_each, model.S) == (3, tr.shape[1], 3 * tr.shape[1])
    with pytest.raises(native.NativeError):
        model.update_tables(2, tr[:2], rw[:2], tm[:2])                           # MDPs [2, 4) of 3
    with pytest.raises(native.NativeError):
        model.update_tables(0, tr[:1], rw[:1], None)                             # terminal flags missing
    rng = _rng_states(2)
    p = np.ones(5) / 5
    with pytest.raises(native.NativeError):
        ctx.uct_plan(model, [0, 0], 5, 5, 0.8, 10.0, p, p, rng, model_index=[0, 3])
    with pytest.raises(native.NativeError):
        ctx.opd_plan(model, [0, tr.shape[1]], 50, 0.8, 0.0, rng, model_index=[0, 1])
    model.close()


def test_agent_delta_upload_through_versioned_tables():
    """SURVEY 8 f-2 at agent level: an MCTSAgent / ValueIterationAgent on a FiniteMDPEnv whose MDP is edited row by row
    between two act() calls -- the cached device model is PATCHED (no new upload, no hashing: MDP.tables_version) and the
    next plan equals a fresh agent's on the edited tables (= the oracle's)."""
    from oracle import oracle
    from rl_agents_amd import native
    from rl_agents_amd.agents.dynamic_programming.value_iteration import ValueIterationAgent
    from rl_agents_amd.agents.tree_search.mcts import MCTSAgent
    from rl_agents_amd.envs import FiniteMDPEnv, generators
    cfg = {k: v for k, v in generators.highway_shaped(10, 10, 100, seed=2).items() if k != "original_shape"}
    cfg["state"] = 40
    env = FiniteMDPEnv(cfg)
    env.reset()
    agent = MCTSAgent(env, dict(budget=400, gamma=0.8))
    agent.seed(5)
    agent.act(40)
    cache = agent.planner.models
    assert cache.uploads == 1
    g = np.random.Generator(np.random.PCG64(0))
    t, r, term = np.array(env.mdp.transition), np.array(env.mdp.reward), np.array(env.mdp.terminal)
    rows = g.choice(t.shape[0], size=50, replace=False)
    t[rows] = g.integers(0, t.shape[0], size=(50, 5))
    r[rows] = g.random((50, 5))
    term[rows] = g.random(50) < 0.2
    env.mdp.edit_rows(rows, transition=t[rows], reward=r[rows], terminal=term[rows])
    rng_before = native.rng_state_from_generator(agent.planner.np_random)
    plan = agent.plan(40)
    assert cache.uploads == 1 and cache.row_updates == 50
    c = agent.planner.config
    p = np.ones(5) / 5
    ref = oracle.uct_plan(t, r, term, 40, c["episodes"], c["horizon"], c["gamma"], c["temperature"], p, p, rng_before,
                          max_plan_len=c["horizon"])
    np.testing.assert_array_equal(plan, ref["plan"])
    # value iteration: same protocol, Q of the edited tables
    vi = ValueIterationAgent(env, dict(gamma=0.95, iterations=60))
    up0 = vi.models.uploads
    rows2 = g.choice(t.shape[0], size=10, replace=False)
    r[rows2] = g.random((10, 5))
    env.mdp.edit_rows(rows2, reward=r[rows2])
    q = vi.get_state_action_value()
    q_ref, _ = oracle.vi_solve("deterministic", t, r, term, gamma=0.95, iterations=60)
    assert vi.models.uploads == up0 and vi.models.row_updates == 10 and np.array_equal(q, q_ref)
