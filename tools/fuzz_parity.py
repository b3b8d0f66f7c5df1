"""Randomised parity sweep: HIP planners vs the CPU oracle on random finite MDPs and random parameters.

    MI355PLAN_NO_TORCH=1 python tools/fuzz_parity.py [n_cases] [seed]

Every case draws an MDP (states, actions, terminal rate, reward pattern), planner parameters (budget / episodes /
horizon / gamma / temperature / terminal reward / truncation / done rule) and a batch of roots, and compares every
output of the device with the oracle's, bit for bit.  tests/test_gpu_fuzz.py runs a bounded number of cases.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MI355PLAN_NO_TORCH", "1")
from oracle import oracle  # noqa: E402
from rl_agents_amd import native  # noqa: E402


def random_mdp(g):
    s = int(g.choice([1, 2, 3, 7, 40, 257, 1500]))
    a = int(g.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 70]))   # (70: more actions than lanes -- the plain OPD kernels, round 4)
    t = g.integers(0, s, size=(s, a), dtype=np.int64)
    kind = g.integers(0, 4)
    if kind == 0:
        r = g.random((s, a))
    elif kind == 1:
        r = g.integers(0, 2, size=(s, a)).astype(np.float64)          # many exact ties
    elif kind == 2:
        r = np.round(g.random((s, a)), 1)                              # few distinct values
    else:
        r = np.full((s, a), float(g.choice([0.0, 0.5, 1.0])))         # constant: everything ties
    term = g.random(s) < g.choice([0.0, 0.05, 0.4])
    return t, r, term


def random_mdp_rewards(g, s, a):
    kind = g.integers(0, 3)
    if kind == 0:
        return g.random((s, a))
    if kind == 1:
        return np.round(g.random((s, a)), 1)
    return g.integers(0, 2, size=(s, a)).astype(np.float64)


def rng_states(g, n):
    out = g.integers(1, 2 ** 62, size=(n, 6)).astype(np.uint64)
    out[:, 3] |= 1
    out[:, 4] = g.integers(0, 2, size=n)          # some streams start with a buffered 32-bit half
    out[:, 5] = g.integers(0, 2 ** 32, size=n)
    return out


def eq(a, b, what, case):
    if not np.array_equal(np.asarray(a), np.asarray(b)):
        raise AssertionError("{} differs in case {}".format(what, case))


VARIANTS = {}
HEAVY = os.environ.get("FUZZ_HEAVY") == "1"   # larger budgets / horizons / batches (slower: the oracle is the long pole)


def one_case(ctx, g, case):
    t, r, term = random_mdp(g)
    s, a = r.shape
    done_rule = "next" if g.random() < 0.3 else "source"
    max_steps = int(g.choice([0, 0, 3, 17]))
    gamma = float(g.choice([0.0, 0.3, 0.8, 0.95, 0.999]))
    n = int(g.choice([1, 3, 64, 65, 200] + ([1000, 3000] if HEAVY else [])))
    s0 = g.integers(0, s, size=n).astype(np.int32)
    rng = rng_states(g, n)
    model = ctx.load_table(t, r, term, done_rule=done_rule, max_steps=max_steps)
    kinds = ["uct", "uct_policy", "opd", "saopd", "vi", "uct_subtree", "uct_listed", "opd_masked", "ropd",
             "ropd_masked", "saopd_masked", "uct_stoch",                    # (the last three: round 3)
             "vi_batch", "per_root_models", "update_rows"]                  # round 5: one MDP per episode, delta uploads
    only = os.environ.get("FUZZ_KINDS")
    if only:
        kinds = [k for k in kinds if k in only.split(",")]
    kind = kinds[int(g.integers(0, len(kinds)))]
    # OPD kernels: let the host choose the variant, or force one ("ldsx": parent map in HBM; "global": bounds in HBM)
    variant = str(g.choice(["", "lds", "ldsx", "global"]))
    os.environ.pop("MP_OPD_MODEL", None)
    if variant and kind in ("opd", "opd_masked", "ropd", "ropd_masked"):
        os.environ["MP_OPD_MODEL"] = variant
    # high-occupancy OPD kernel: sibling layout of the bounds array (default) or the residue-class layout
    os.environ.pop("MP_OPD_WIDE", None)
    wide = str(g.choice(["", "cls"]))
    if wide:
        os.environ["MP_OPD_WIDE"] = wide
    # closing passes: opd_closing.hpp where it fits (default) or the node-array form everywhere
    os.environ.pop("MP_OPD_CLOSING", None)
    if int(g.integers(0, 3)) == 0:
        os.environ["MP_OPD_CLOSING"] = "chain"
    # main loop: the forms for bounds >= 0 where they apply (default) or the general one everywhere
    os.environ.pop("MP_OPD_LOOP", None)
    if int(g.integers(0, 3)) == 0:
        os.environ["MP_OPD_LOOP"] = "0"
    # UCT on an LDS-resident model: four lanes per root by default for 16 .. 16 384 roots; forced on / off now and then
    os.environ.pop("MP_UCT_QUAD", None)
    quad = str(g.choice(["", "", "0", "1"]))
    if quad and kind in ("uct", "uct_subtree", "per_root_models", "update_rows"):
        os.environ["MP_UCT_QUAD"] = quad
    # round 6: the row kernel on a shared model (default for 16 .. 4096 roots where uct_lone_kernel does not apply) forced on / off,
    # two / four roots per wavefront, one to eight planning wavefronts; on batch models the row / wavefront-per-root forms; batched
    # VI's cluster form (K workgroups per MDP)
    for knob in ("MP_UCT_ROWS", "MP_UCT_ROW_ROOTS", "MP_UCT_ROW_WAVES", "MP_UCT_ROW", "MP_UCT_EACH", "MP_VI_BATCH_CLUSTER", "MP_UCT_LONE_WAVES"):
        os.environ.pop(knob, None)
    rows = str(g.choice(["", "", "0", "1"]))
    if rows and not quad and kind in ("uct", "uct_subtree", "update_rows"):
        os.environ["MP_UCT_ROWS"] = rows
    rpw, rwaves = str(g.choice(["", "2", "4"])), str(g.choice(["", "1", "2", "4", "8"]))
    if rpw:
        os.environ["MP_UCT_ROW_ROOTS"] = rpw
    if rwaves:
        os.environ["MP_UCT_ROW_WAVES"] = rwaves
    each = str(g.choice(["", "", "row0", "each0"]))
    if each == "row0" and kind == "per_root_models":
        os.environ["MP_UCT_ROW"] = "0"
    if each == "each0" and kind == "per_root_models":
        os.environ["MP_UCT_EACH"] = "0"
    cluster = str(g.choice(["", "", "0", "2", "4", "8"]))
    if cluster and kind == "vi_batch":
        os.environ["MP_VI_BATCH_CLUSTER"] = cluster
    # round 6, later: several planning wavefronts per workgroup around one copy of the transitions (uct_lone_kernel<.., MW>: the
    # default for batches of up to 8 roots per CU of a model that fills the LDS) forced on with 1 / 2 / 4 / 8 wavefronts, or off
    lone_w = str(g.choice(["", "", "0", "1", "2", "4", "8"]))
    if lone_w and not quad and not rows and kind in ("uct", "uct_subtree", "update_rows"):
        os.environ["MP_UCT_LONE_WAVES"] = lone_w
    desc = dict(case=case, kind=kind, S=s, A=a, n=n, gamma=gamma, done_rule=done_rule, max_steps=max_steps, variant=variant, quad=quad, wide=wide,
                rows=rows, rpw=rpw, rwaves=rwaves, each=each, cluster=cluster, lone_w=lone_w)
    if kind in ("vi_batch", "per_root_models"):
        # N independent MDPs of this shape (mp_model_load_table_batch): N value-iteration agents in one launch, each to its own
        # allclose exit, in every kernel form; UCT and OPD with one MDP per root; a second round after mp_model_update_tables
        model.close()
        nb = int(g.choice([1, 2, 5, 33]))
        sb = min(s, int(g.choice([1, 3, 40, 130, 700])))
        tb = g.integers(0, sb, size=(nb, sb, a), dtype=np.int64)
        rb = np.stack([random_mdp_rewards(g, sb, a) for _ in range(nb)]) * (10.0 ** -g.integers(0, 4, size=(nb, 1, 1)))
        termb = g.random((nb, sb)) < g.choice([0.0, 0.1])
        with_term = bool(g.random() < 0.8)
        desc.update(n_models=nb, S=sb, with_term=with_term)
        model = ctx.load_table_batch(tb, rb, termb if with_term else None, done_rule=done_rule, max_steps=max_steps)
        tz = termb if with_term else None
        for rnd in range(2):
            if rnd == 1:                                               # the episodes' tables change: the per-step delta
                lo = int(g.integers(0, nb))
                hi = int(g.integers(lo + 1, nb + 1))
                tb[lo:hi] = g.integers(0, sb, size=(hi - lo, sb, a))
                rb[lo:hi] = np.stack([random_mdp_rewards(g, sb, a) for _ in range(hi - lo)])
                if with_term:
                    termb[lo:hi] = g.random((hi - lo, sb)) < 0.2
                model.update_tables(lo, tb[lo:hi], rb[lo:hi], termb[lo:hi] if with_term else None)
            if kind == "vi_batch":
                for knob in ("MP_VI_BATCH_NO_REG", "MP_VI_BATCH_NO_WGR", "MP_VI_BATCH_NO_VLDS"):
                    os.environ.pop(knob, None)
                form = int(g.integers(0, 4))
                for knob in (["MP_VI_BATCH_NO_REG"] if form >= 1 else []) + (["MP_VI_BATCH_NO_WGR"] if form >= 2 else []) + \
                        (["MP_VI_BATCH_NO_VLDS"] if form >= 3 else []):
                    os.environ[knob] = "1"
                iterations = int(g.choice([0, 1, 7, 100]))
                rew = rb * float(g.choice([1.0, -1.0, 10.0]))
                if form or rnd:                                        # (other rewards than the model's: reload)
                    m2 = ctx.load_table_batch(tb, rew, tz)
                else:
                    m2 = ctx.load_table_batch(tb, rew, tz)
                try:
                    q, sweeps = ctx.vi_solve_batch(m2, gamma, iterations)
                finally:
                    m2.close()
                    for knob in ("MP_VI_BATCH_NO_REG", "MP_VI_BATCH_NO_WGR", "MP_VI_BATCH_NO_VLDS"):
                        os.environ.pop(knob, None)
                q_ref, sw_ref = oracle.vi_solve_each(tb, rew, tz, gamma=gamma, iterations=iterations)
                desc.update(iterations=iterations, form=form)
                eq(sweeps, sw_ref, "sweeps per MDP", desc)
                eq(q, q_ref, "Q per MDP", desc)
            else:
                nr = min(n, 70)
                mi = g.integers(0, nb, size=nr).astype(np.int32)
                sl = g.integers(0, sb, size=nr).astype(np.int32)
                st0 = g.integers(0, 4, size=nr).astype(np.int32) if max_steps else None
                if 2 <= a <= 8 or a in (1, 9, 13):
                    episodes, horizon = int(g.choice([1, 6, 30])), int(g.choice([1, 5, 12]))
                    pp = g.random(a) + 0.05
                    pp /= pp.sum()
                    rr, rr_ref = rng[:nr].copy(), rng[:nr].copy()
                    out = ctx.uct_plan(model, sl, episodes, horizon, gamma, 5.0, pp, pp, rr, root_steps=st0, max_plan_len=max(horizon, 1),
                                       model_index=mi)
                    ref = oracle.uct_plan_each(tb, rb, tz if tz is not None else np.zeros((nb, sb), bool), mi, sl, episodes, horizon, gamma,
                                               5.0, pp, pp, rr_ref, max_plan_len=max(horizon, 1), steps0=st0, max_steps=max_steps,
                                               done_rule=done_rule)
                    for k in ("plans", "plan_len", "root_value", "root_child_count", "env_steps"):
                        eq(out[k], ref[k], "per-root-model UCT " + k, desc)
                    eq(rr, ref["rng_after"], "per-root-model UCT generator", desc)
                if float(rb.min()) >= 0.0 and float(rb.max()) <= 1.0 and gamma < 1.0:
                    budget = int(g.choice([a, 3 * a, 40 * a]))
                    rr, rr_ref = rng[:nr].copy(), rng[:nr].copy()
                    out = ctx.opd_plan(model, sl, budget, gamma, 0.0, rr, max_plan_len=16, model_index=mi)
                    ref = oracle.opd_plan_each(tb, rb, tz if tz is not None else np.zeros((nb, sb), bool), mi, sl, budget, gamma, 0.0, rr_ref,
                                               max_plan_len=16, done_rule=done_rule)
                    for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status"):
                        eq(out[k], ref[k], "per-root-model OPD " + k, desc)
                    eq(rr, ref["rng_after"], "per-root-model OPD generator", desc)
    elif kind == "update_rows":
        # delta upload (mp_model_update_rows): rows replaced in place, with and without terminal flags, rewards that stay in /
        # leave the table of distinct values -- then every planner must see exactly the edited tables
        t2, r2, term2 = t.copy(), r.copy(), term.copy()
        for rnd in range(2):
            k = int(g.integers(1, max(2, s // 3 + 1)))
            rows = g.choice(s, size=min(k, s), replace=False).astype(np.int32)
            t2[rows] = g.integers(0, s, size=(len(rows), a))
            r2[rows] = g.choice(np.unique(r), size=(len(rows), a)) if g.random() < 0.5 else g.random((len(rows), a))
            flags = bool(g.random() < 0.5)
            if flags:
                term2[rows] = g.random(len(rows)) < 0.3
            model.update_rows(rows, t2[rows], r2[rows], term2[rows] if flags else None)
            q, sweeps = ctx.vi_solve(model, gamma, 30)
            q_ref, sw_ref = oracle.vi_solve("deterministic", t2, r2, term2, gamma=gamma, iterations=30)
            eq(q, q_ref, "Q after update_rows", desc)
            if sweeps != sw_ref:
                raise AssertionError("sweeps {} vs {} in case {}".format(sweeps, sw_ref, desc))
            if 2 <= a <= 8:
                pp = np.ones(a) / a
                rr, rr_ref = rng.copy(), rng.copy()
                out = ctx.uct_plan(model, s0, 12, 6, gamma, 5.0, pp, pp, rr, max_plan_len=6)
                ref = oracle.uct_plan_batch(t2, r2, term2, s0, 12, 6, gamma, 5.0, pp, pp, rr_ref, max_steps=max_steps, done_rule=done_rule,
                                            max_plan_len=6)
                for kk in ("plans", "root_value", "env_steps"):
                    eq(out[kk], ref[kk], "UCT after update_rows " + kk, desc)
                eq(rr, ref["rng_after"], "UCT generator after update_rows", desc)
    elif kind == "vi":
        model.close()
        iterations = int(g.choice([1, 7, 100, 300]))
        rewards = r * float(g.choice([1.0, -1.0, 10.0]))             # VI takes any reward range
        which = int(g.integers(0, 4))
        desc.update(iterations=iterations, vi=["det", "robust", "sparse", "dense"][which])
        if which == 0:
            model = ctx.load_table(t, rewards, term)
            q, sweeps = ctx.vi_solve(model, gamma, iterations)
            q_ref, sweeps_ref = oracle.vi_solve("deterministic", t, rewards, term, gamma=gamma, iterations=iterations)
            eq(q, q_ref, "Q", desc)
            eq(ctx.vi_solve_v(model, gamma, iterations),
               oracle.vi_solve("deterministic", t, rewards, term, gamma=gamma, iterations=iterations, state_value=True), "V", desc)
        elif which == 1:
            m = int(g.choice([1, 2, 3]))
            tm = np.stack([g.integers(0, s, size=(s, a), dtype=np.int64) for _ in range(m)])
            rm = np.stack([rewards * float(g.uniform(0.5, 1.0)) for _ in range(m)])
            model = ctx.load_table(tm, rm)
            q, sweeps = ctx.vi_solve(model, gamma, iterations, robust=True)
            q_ref, sweeps_ref = oracle.vi_solve("deterministic", tm, rm, None, gamma=gamma, iterations=iterations, robust=True)
            eq(q, q_ref, "robust Q", desc)
        elif which == 2:
            b = int(g.choice([1, 2, 5, 8, 9, 127, 128, 129, 300]))
            nxt = g.integers(0, s, size=(s, a, b), dtype=np.int64)
            pr = g.random((s, a, b)) + 0.01
            pr /= pr.sum(-1, keepdims=True)
            model = ctx.load_sparse(pr, nxt, rewards, term)
            q, sweeps = ctx.vi_solve(model, gamma, iterations)
            q_ref, sweeps_ref = oracle.vi_solve("sparse", pr, rewards, term, gamma=gamma, iterations=iterations, next_states=nxt)
            eq(q, q_ref, "sparse Q", desc)
        else:
            # dense: any row length (every shape of numpy's pairwise recursion); a few cases beyond its 8192-element buffer
            sd = int(g.choice([1, 2, 5, 7, 8, 9, 17, 64, 100, 127, 128, 129, 130, 200, 257, 300, 500, 1029])) if g.random() < 0.85 \
                else int(g.choice([8193, 9000]))
            ad = a if sd <= 1029 else 1
            sparse_rows = g.random() < 0.3                              # exact zeros in the rows
            pr = g.random((sd, ad, sd)) ** 3 + 1e-3
            if sparse_rows:
                pr[g.random((sd, ad, sd)) < 0.5] = 0.0
                pr[..., 0] += 1e-3
            pr /= pr.sum(-1, keepdims=True)
            rd = (g.random((sd, ad)) if sd > s else np.resize(rewards, (sd, a))[:, :ad]) * 1.0
            td = g.random(sd) < 0.1
            if sd > 1029:
                iterations = min(iterations, 7)
            desc.update(dense_states=sd, dense_actions=ad)
            model = ctx.load_dense(pr, rd, td)
            q_ref, sweeps_ref = oracle.vi_solve("stochastic", pr, rd, td, gamma=gamma, iterations=iterations)
            forms = [("exact", None)]
            if sd <= 1029:
                forms += [("exact", "global"), ("exact", "pieces"), ("mfma", None)]
            form, vmode = forms[int(g.integers(0, len(forms)))]
            desc.update(dense_form=form, dense_v=vmode)
            os.environ.pop("MP_VI_EXACT_V", None)
            if vmode:
                os.environ["MP_VI_EXACT_V"] = vmode
            ctx.vi_dense_mode(form)
            try:
                q, sweeps = ctx.vi_solve(model, gamma, iterations)
                if form == "exact" and sd <= 1029:
                    eq(ctx.vi_solve_v(model, gamma, iterations),
                       oracle.vi_solve("stochastic", pr, rd, td, gamma=gamma, iterations=iterations, state_value=True), "dense V", desc)
            finally:
                ctx.vi_dense_mode("exact")
                os.environ.pop("MP_VI_EXACT_V", None)
            if form == "exact":
                eq(q, q_ref, "dense Q (numpy's order)", desc)
            else:
                if not np.allclose(q, q_ref, rtol=1e-12, atol=1e-12 * max(1.0, float(np.abs(q_ref).max()))):
                    raise AssertionError("dense Q differs beyond 1e-12 in case {}".format(desc))
                if abs(sweeps - sweeps_ref) > 1:
                    raise AssertionError("dense sweeps {} vs {} in case {}".format(sweeps, sweeps_ref, desc))
                sweeps = sweeps_ref
        if sweeps != sweeps_ref:
            raise AssertionError("sweeps {} vs {} in case {}".format(sweeps, sweeps_ref, desc))
    elif kind == "uct_subtree":     # step_strategy "subtree": plan, re-root at the executed action, plan again
        n = min(n, 6)
        s0, rng = s0[:n].copy(), rng[:n]
        episodes, horizon = int(g.choice([1, 5, 20])), int(g.choice([2, 6, 12]))
        temperature = float(g.choice([1.0, 10.0]))
        desc.update(episodes=episodes, horizon=horizon, temperature=temperature, n=n)
        prior = g.random(a) + 0.01
        prior /= prior.sum()
        trees, ref_rng = [None] * n, rng.copy()
        ctx.uct_reset_tree()
        prev = None
        for step in range(3):
            if prev is not None:
                ctx.uct_step_tree(prev)
            out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, prior, prior, rng, max_plan_len=horizon)
            for i in range(n):
                if trees[i] is not None:
                    trees[i] = oracle.uct_reroot(trees[i], int(prev[i]), a)
                o = oracle.uct_plan(t, r, term, int(s0[i]), episodes, horizon, gamma, temperature, prior, prior, ref_rng[i],
                                    max_steps=max_steps, done_rule=done_rule, max_plan_len=horizon, init_tree=trees[i])
                eq(out["plans"][i, :out["plan_len"][i]], o["plan"], "plan", desc)
                eq(rng[i], o["rng_after"], "rng", desc)
                tree = ctx.uct_tree(i)
                for k in ("count", "value", "first_child"):
                    eq(tree[k], o["tree"][k], "tree " + k, desc)
                trees[i], ref_rng[i] = o["tree"], o["rng_after"]
            prev = np.where(out["plan_len"] > 0, out["plans"][:, 0], 0).astype(np.int32)
            s0 = t[s0, prev].astype(np.int32)
        ctx.uct_reset_tree()
    elif kind in ("uct", "uct_policy"):
        episodes, horizon = int(g.choice([0, 1, 5, 33, 60] + ([150, 400] if HEAVY else []))), int(g.choice([1, 2, 9, 30] + ([60] if HEAVY else [])))
        temperature = float(g.choice([0.0, 1.0, 10.0, 3000.0]))
        steps0 = g.integers(0, 3, size=n).astype(np.int32) if max_steps else None
        desc.update(episodes=episodes, horizon=horizon, temperature=temperature)
        if kind == "uct_policy" and a in (2, 3, 4, 5, 6, 7, 8):
            w = g.random((2, s, a)) ** 2
            w[g.random((2, s, a)) < 0.2] = 0.0
            w[:, np.arange(s), g.integers(0, a, size=s)] += 0.1
            prior, rollout = w[0] / w[0].sum(1, keepdims=True), w[1] / w[1].sum(1, keepdims=True)
            policy = ctx.load_policy(model, prior, rollout)
            rng_dev = rng.copy()
            out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, None, None, rng_dev, root_steps=steps0,
                               max_plan_len=horizon, policy=policy)
            policy.close()
        else:
            prior = g.random(a) + 0.01
            prior /= prior.sum()
            rollout = g.random(a) + 0.01
            if g.random() < 0.3:
                rollout[g.integers(0, a)] = 0.0
                rollout += 1e-3 if rollout.sum() == 0 else 0.0
            rollout /= rollout.sum()
            rng_dev = rng.copy()
            out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, prior, rollout, rng_dev, root_steps=steps0,
                               max_plan_len=horizon)
        ref = oracle.uct_plan_batch(t, r, term, s0, episodes, horizon, gamma, temperature, prior, rollout, rng.copy(),
                                    steps0=steps0, max_steps=max_steps, done_rule=done_rule, max_plan_len=horizon, n_threads=8)
        for k in ("plans", "plan_len", "root_value", "root_child_count", "root_child_value", "env_steps"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
    elif kind == "uct_listed":      # policies over restricted action sets (mcts.py:59-97): listed-policy kernel variant
        from tests.helpers import reference_policy_lists
        if a < 2:
            model.close()
            return desc
        avail = g.random((s, a)) >= float(g.choice([0.2, 0.5, 0.8]))
        avail[np.arange(s), g.integers(0, a, size=s)] = True
        kinds = [{"type": "random"}, {"type": "random_available"},
                 {"type": "preference", "action": int(g.integers(0, a + 1)), "ratio": float(g.choice([2, 3, 0.5]))}]
        pl, rl = reference_policy_lists(kinds[int(g.integers(0, 3))], avail), reference_policy_lists(kinds[int(g.integers(0, 3))], avail)
        prior, rollout, listed = np.zeros((s, a)), np.zeros((s, a)), np.zeros((s, a), bool)
        for i in range(s):
            prior[i, pl["actions"][i]] = pl["p"][i]
            listed[i, pl["actions"][i]] = True
            rollout[i, rl["actions"][i]] = rl["p"][i]
        episodes, horizon = int(g.choice([0, 1, 5, 33, 60])), int(g.choice([1, 2, 9, 30]))
        temperature = float(g.choice([0.0, 1.0, 10.0, 3000.0]))
        steps0 = g.integers(0, 3, size=n).astype(np.int32) if max_steps else None
        desc.update(episodes=episodes, horizon=horizon, temperature=temperature)
        policy = ctx.load_policy(model, prior, rollout, listed=listed)
        rng_dev = rng.copy()
        if a > 8:   # more than 8 actions (round 4): the loop forms of the other kernel, on this deterministic table; open / closed loop
            closed = bool(g.integers(0, 2))
            mpl = 2 * horizon + 2
            desc.update(loop_form=True, closed=closed)
            out = ctx.uct_plan_stochastic(model, s0, episodes, horizon, gamma, temperature, None, None, rng_dev, closed_loop=closed,
                                          root_steps=steps0, max_plan_len=mpl, policy=policy)
            policy.close()
            ref = oracle.uct_plan_stoch_batch("deterministic", t, r, term, s0, episodes, horizon, gamma, temperature, pl, rl,
                                              rng.copy(), np.tile(np.array([0, 1, 0, 1, 0, 0], np.uint64), (n, 1)),
                                              closed_loop=closed, steps0=steps0, max_steps=max_steps, done_rule=done_rule,
                                              max_plan_len=mpl, n_threads=8)
            for k in ("plans", "plan_len", "root_value", "env_steps"):
                eq(out[k], ref[k], k, desc)
            eq(rng_dev, ref["rng_after"], "rng", desc)
            model.close()
            return desc
        out = ctx.uct_plan(model, s0, episodes, horizon, gamma, temperature, None, None, rng_dev, root_steps=steps0,
                           max_plan_len=horizon, policy=policy)
        policy.close()
        ref = oracle.uct_plan_batch(t, r, term, s0, episodes, horizon, gamma, temperature, pl, rl, rng.copy(), steps0=steps0,
                                    max_steps=max_steps, done_rule=done_rule, max_plan_len=horizon, n_threads=8)
        for k in ("plans", "plan_len", "root_value", "root_child_count", "root_child_value", "env_steps"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
    elif kind == "opd_masked":      # deterministic.py:32-35 on environments that restrict the available actions
        avail = g.random((s, a)) >= float(g.choice([0.2, 0.5, 0.8]))
        avail[np.arange(s), g.integers(0, a, size=s)] = True
        model.close()
        model = ctx.load_table(t, r, term, done_rule=done_rule, available=avail)
        budget = int(g.choice([0, 1, a, 3 * a + 1, 100, 700]))
        tr = float(g.choice([0.0, 0.25, 1.0]))
        if gamma >= 0.999:
            gamma = 0.95
        desc.update(budget=budget, terminal_reward=tr, gamma=gamma)
        rng_dev = rng.copy()
        out = ctx.opd_plan(model, s0, budget, gamma, tr, rng_dev, max_plan_len=budget // a + 1)
        ref = oracle.opd_plan_batch(t, r, term, s0, budget, gamma, tr, rng.copy(), done_rule=done_rule,
                                    max_plan_len=budget // a + 1, n_threads=8, available=avail)
        for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
    elif kind == "uct_stoch":       # MCTS on stochastic / sparse finite MDPs, open and closed loop (uct_stoch.hip)
        mode = str(g.choice(["stochastic", "sparse", "deterministic"]))
        closed = bool(g.random() < 0.6)
        model.close()
        nxt = None
        if mode == "stochastic":
            s = min(s, 257)                                           # [S, A, S] thresholds
            t, r, term, s0 = t[:s] % s, r[:s], term[:s], s0 % s
            p_ = g.random((s, a, s)) ** float(g.choice([1.0, 4.0, 12.0]))      # from flat to a few likely next states
            if g.random() < 0.3:
                p_[g.random((s, a, s)) < 0.5] = 0.0                   # exact zeros inside the rows
                p_[:, :, 0] += 1e-3
            p_ /= p_.sum(axis=-1, keepdims=True)
            model = ctx.load_dense(p_, r, term)
            trans = p_
        elif mode == "sparse":
            b = int(g.choice([1, 2, 3, 5]))
            nxt = g.integers(0, s, size=(s, a, b), dtype=np.int64)
            p_ = g.random((s, a, b)) + 0.05
            p_ /= p_.sum(axis=-1, keepdims=True)
            model = ctx.load_sparse(p_, nxt, r, term)
            trans = p_
        else:
            model = ctx.load_table(t, r, term)
            trans = t
        model.set_episode_rules(done_rule, max_steps)
        episodes = int(g.choice([0, 1, 7, 33, 90]))
        horizon = int(g.choice([1, 2, 6, 11, 30]))
        temperature = float(g.choice([0.0, 1.0, 10.0, 200.0]))
        prior = g.random(a) + 0.05
        prior /= prior.sum()
        roll = g.random(a) ** 2 + 1e-3
        roll /= roll.sum()
        steps0 = g.integers(0, 3, size=n).astype(np.int32)
        erng = rng_states(g, n)
        erng[:, 4:] = 0
        desc.update(mode=mode, closed=closed, episodes=episodes, horizon=horizon, temperature=temperature, S=s)
        rng_dev = rng.copy()
        mpl = 2 * horizon + 2
        out = ctx.uct_plan_stochastic(model, s0, episodes, horizon, gamma, temperature, prior, roll, rng_dev, env_rng_state=erng,
                                      closed_loop=closed, root_steps=steps0, max_plan_len=mpl)
        ref = oracle.uct_plan_stoch_batch(mode, trans, r, term, s0, episodes, horizon, gamma, temperature, prior, roll, rng.copy(),
                                          erng, next_states=nxt, closed_loop=closed, steps0=steps0, max_steps=max_steps,
                                          done_rule=done_rule, max_plan_len=mpl, n_threads=8)
        for k in ("plans", "plan_len", "env_steps", "root_value"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
        root = int(g.integers(0, n))                                  # one whole tree, node for node
        tree = ctx.uct_stoch_tree(root)
        one = oracle.uct_plan_stoch(mode, trans, r, term, int(s0[root]), episodes, horizon, gamma, temperature, prior, roll,
                                    rng[root].copy(), erng[root], next_states=nxt, closed_loop=closed, steps0=int(steps0[root]),
                                    max_steps=max_steps, done_rule=done_rule, max_plan_len=mpl)["tree"]
        for k in ("parent", "action", "is_obs", "count", "value"):
            eq(tree[k], one[k], "tree " + k, desc)
    elif kind == "ropd_masked":     # robust OPD on models that restrict their actions: union over the models (robust.py:22-25)
        m = int(g.choice([1, 2, 3, 5]))
        tm = np.stack([t] + [g.integers(0, s, size=(s, a), dtype=np.int64) for _ in range(m - 1)])
        rm = np.stack([r] + [np.clip(r * float(g.uniform(0.5, 1.0)), 0.0, 1.0) for _ in range(m - 1)])
        termm = np.stack([term] + [g.random(s) < 0.1 for _ in range(m - 1)])
        avail = g.random((m, s, a)) >= float(g.choice([0.2, 0.5, 0.8]))
        for k in range(m):
            avail[k, np.arange(s), g.integers(0, a, size=s)] = True
        if m > 1 and g.random() < 0.3:
            avail[m - 1] = True                                        # a model whose env has no get_available_actions
        model.close()
        model = ctx.load_joint(tm, rm, termm, done_rule=done_rule, available=avail)
        budget = int(g.choice([0, 1, a, 3 * a + 1, 100, 400]))
        tr = float(g.choice([0.0, 0.25, 1.0]))
        if gamma >= 0.999:
            gamma = 0.95
        n = min(n, 200)
        joint = g.integers(0, s, size=(n, m)).astype(np.int32)
        if g.random() < 0.5:
            joint[:] = joint[:, :1]
        desc.update(budget=budget, terminal_reward=tr, gamma=gamma, M=m, n=n)
        rng_dev = rng[:n].copy()
        out = ctx.ropd_plan(model, joint, budget, gamma, tr, rng_dev, max_plan_len=budget // a + 1)
        ref = oracle.ropd_plan_batch(tm, rm, termm, joint, budget, gamma, tr, rng[:n].copy(), done_rule=done_rule,
                                     max_plan_len=budget // a + 1, n_threads=8, available=avail)
        for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
        root = int(g.integers(0, n))
        if out["status"][root] == 0 and budget >= a:                  # one whole tree
            tree = ctx.ropd_tree(root, 1 + (budget // a) * a, m)
            one = oracle.ropd_plan(tm, rm, termm, joint[root], budget, gamma, tr, rng_state=rng[root].copy(), done_rule=done_rule,
                                   max_plan_len=budget // a + 1, available=avail)["tree"]
            for k in ("parent", "action", "depth", "count", "first_child", "n_children", "state", "reward"):
                eq(tree[k], one[k], "tree " + k, desc)
            eq(tree["lower"].min(axis=1), one["lower"].min(axis=1), "tree min lower", desc)
    elif kind == "ropd":            # discrete robust OPD (agents/robust/robust.py:28-50): M models, joint states
        m = int(g.choice([1, 2, 3, 7, 33, 40]))   # (more than 32 models: round 4)
        tm = np.stack([t] + [g.integers(0, s, size=(s, a), dtype=np.int64) for _ in range(m - 1)])
        rm = np.stack([r] + [np.clip(r * float(g.uniform(0.5, 1.0)), 0.0, 1.0) for _ in range(m - 1)])
        termm = np.stack([term] + [g.random(s) < 0.1 for _ in range(m - 1)])
        model.close()
        model = ctx.load_joint(tm, rm, termm, done_rule=done_rule)
        budget = int(g.choice([0, 1, a, 3 * a + 1, 100, 400]))
        tr = float(g.choice([0.0, 0.25, 1.0]))
        if gamma >= 0.999:
            gamma = 0.95
        n = min(n, 200)
        joint = g.integers(0, s, size=(n, m)).astype(np.int32)
        if g.random() < 0.5:
            joint[:] = joint[:, :1]                                   # every model in the same state (the usual root)
        desc.update(budget=budget, terminal_reward=tr, gamma=gamma, M=m, n=n)
        rng_dev = rng[:n].copy()
        out = ctx.ropd_plan(model, joint, budget, gamma, tr, rng_dev, max_plan_len=budget // a + 1)
        ref = oracle.ropd_plan_batch(tm, rm, termm, joint, budget, gamma, tr, rng[:n].copy(), done_rule=done_rule,
                                     max_plan_len=budget // a + 1, n_threads=8)
        for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
    elif kind == "opd":
        budget = int(g.choice([0, 1, a, 3 * a + 1, 100, 700] + ([3000, 6000] if HEAVY else [])))
        tr = float(g.choice([0.0, 0.25, 1.0]))
        if gamma >= 0.999:
            gamma = 0.95
        desc.update(budget=budget, terminal_reward=tr, gamma=gamma)
        rng_dev = rng.copy()
        out = ctx.opd_plan(model, s0, budget, gamma, tr, rng_dev, max_plan_len=budget // a + 1)
        ref = oracle.opd_plan_batch(t, r, term, s0, budget, gamma, tr, rng.copy(), done_rule=done_rule,
                                    max_plan_len=budget // a + 1, n_threads=8)
        for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status"):
            eq(out[k], ref[k], k, desc)
        eq(rng_dev, ref["rng_after"], "rng", desc)
        root = int(g.integers(0, n))
        if out["status"][root] == 0:                                  # one whole tree, node for node
            tree = ctx.opd_tree(root, 1 + (budget // a) * a)
            one = oracle.opd_plan(t, r, term, int(s0[root]), budget, gamma, tr, rng[root].copy(), done_rule=done_rule,
                                  max_plan_len=budget // a + 1)["tree"]
            for k in one:
                eq(tree[k], one[k], "tree " + k, desc)
    else:                           # "saopd" / "saopd_masked": state-aware OPD, two consecutive plans per planner
        avail = None
        os.environ.pop("MP_SAOPD_MODEL", None)
        os.environ.pop("MP_SAOPD_LDS", None)
        os.environ.pop("MP_SAOPD_DICT", None)
        os.environ.pop("MP_SAOPD_ORDER", None)
        # round 4: the wave kernel with the dictionaries in LDS is the default where they fit; "0" keeps them in global memory.
        # Dispatch by expected cost (only the order in which planners start) is forced on for half of the cases.
        if g.random() < 0.4:
            os.environ["MP_SAOPD_DICT"] = "0"
        if g.random() < 0.5:
            os.environ["MP_SAOPD_ORDER"] = "1"
        os.environ.pop("MP_SAOPD_CSR", None)
        if g.random() < 0.6:
            os.environ["MP_SAOPD_CSR"] = str(int(g.integers(0, 2)))   # never / from the second plan on (default: from the fourth)
        desc.update(dict_lds=os.environ.get("MP_SAOPD_DICT", "auto"), order=os.environ.get("MP_SAOPD_ORDER", "auto"))
        if kind == "saopd_masked":  # deterministic.py:32-35 under the state-aware planner: phantom rows (round 3)
            avail = g.random((s, a)) >= float(g.choice([0.2, 0.5, 0.8]))
            avail[np.arange(s), g.integers(0, a, size=s)] = True
            model.close()
            model = ctx.load_table(t, r, term, done_rule=done_rule, available=avail)
            mapping = str(g.choice(["wave", "lane", "lds"]))
            if mapping == "lane":
                os.environ["MP_SAOPD_MODEL"] = "lane"
            elif mapping == "lds":
                os.environ["MP_SAOPD_LDS"] = "1"
            desc.update(mapping=mapping)
        budget = int(g.choice([0, a, 5 * a, 120, 300] + ([1000] if HEAVY else [])))
        tr = float(g.choice([0.0, 0.5]))
        if gamma >= 0.999:
            gamma = 0.9
        n = min(n, 65)
        s0, rng = s0[:n], rng[:n]
        cfgs = dict(accuracy=float(g.choice([0.0, 0.0, 0.05])), backup_aggregated_nodes=bool(g.random() < 0.8),
                    prune_suboptimal_leaves=bool(g.random() < 0.8))
        desc.update(budget=budget, terminal_reward=tr, gamma=gamma, n=n, **cfgs)
        planners = native.StateAwarePlanners(ctx, model, n)
        ref_pl, ref_rng, states, dead = [None] * n, rng.copy(), s0.copy(), np.zeros(n, bool)
        for step in range(2):
            out = planners.plan(states, budget, gamma, tr, rng, **cfgs)
            for i in range(n):
                if dead[i]:
                    continue
                try:
                    o = oracle.saopd_plan(t, r, term, int(states[i]), budget, gamma, terminal_reward=tr, rng_state=ref_rng[i],
                                          planner=ref_pl[i], done_rule=done_rule, max_plan_len=budget + 1, available=avail, **cfgs)
                except ValueError:
                    if out["status"][i] != native.MP_ERR_ARG:
                        raise AssertionError("status {} where the reference raises, case {}".format(out["status"][i], desc))
                    dead[i] = True
                    continue
                if out["status"][i] != 0:
                    if os.environ.get("FUZZ_DUMP"):
                        np.savez(os.environ["FUZZ_DUMP"], t=t, r=r, term=term, s0=states[i], rng=ref_rng[i], step=step)
                    raise AssertionError("status {} in case {}".format(out["status"][i], desc))
                eq(out["plans"][i, :out["plan_len"][i]], o["plan"], "plan", desc)
                eq(out["updates"][i], o["updates"], "updates", desc)
                eq(out["env_steps"][i], o["env_steps"], "env steps", desc)
                eq(rng[i], o["rng_after"], "rng", desc)
                ref_pl[i], ref_rng[i] = o["planner"], o["rng_after"]
                if i % 16 == 0:
                    tree, sv = planners.export(i)
                    eq(sv, o["state_values"], "state values", desc)
                    eq(tree["alive"], o["tree"]["alive"], "leaves", desc)
                    for k in ("parent", "action", "lower", "first_child", "n_children", "count"):
                        eq(tree[k], o["tree"][k], "tree " + k, desc)
            nxt = np.where(out["plan_len"] > 0, t[states, np.maximum(out["plans"][:, 0], 0)], states)
            states = nxt.astype(np.int32)
        planners.close()
        os.environ.pop("MP_SAOPD_MODEL", None)
        os.environ.pop("MP_SAOPD_LDS", None)
        os.environ.pop("MP_SAOPD_DICT", None)
        os.environ.pop("MP_SAOPD_ORDER", None)
    model.close()
    return desc


def run(n_cases, seed, ctx=None, verbose=False):
    own = ctx is None
    ctx = ctx or native.Context(0)
    g = np.random.Generator(np.random.PCG64(seed))
    kinds = {}
    forced = os.environ.get("MP_OPD_MODEL")        # the cases force kernel variants through the environment: put it back
    try:
        for case in range(n_cases):
            try:
                d = one_case(ctx, g, case)
            except AssertionError:
                raise
            except Exception as e:     # a device error code: say which case it was
                raise RuntimeError("case {} (seed {}): {}".format(case, seed, e))
            kinds[d["kind"]] = kinds.get(d["kind"], 0) + 1
            v = ctx.last_kernel_variant()        # (the LAST launch of the case: which kernel forms the sweep reached)
            VARIANTS[v] = VARIANTS.get(v, 0) + 1
            if verbose:
                print(d)
    finally:
        os.environ.pop("MP_OPD_MODEL", None)
        os.environ.pop("MP_OPD_CLOSING", None)
        os.environ.pop("MP_OPD_LOOP", None)
        os.environ.pop("MP_OPD_WIDE", None)
        for knob in ("MP_UCT_QUAD", "MP_UCT_ROWS", "MP_UCT_ROW_ROOTS", "MP_UCT_ROW_WAVES", "MP_UCT_ROW", "MP_UCT_EACH", "MP_VI_BATCH_CLUSTER", "MP_UCT_LONE_WAVES"):
            os.environ.pop(knob, None)
        if forced is not None:
            os.environ["MP_OPD_MODEL"] = forced
    if own:
        ctx.close()
    return kinds


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print("ok", run(n, seed, verbose=os.environ.get("FUZZ_VERBOSE") == "1"))
    print("last kernel form of each case:", dict(sorted(VARIANTS.items(), key=lambda kv: -kv[1])))
