"""Multi-GPU execution of the planning path: roots shard, results gather.

One process per GPU (``python -m torch.distributed.run --nproc-per-node N``; backend ``nccl`` = RCCL over
xGMI on MI355X, ``gloo`` on CPU for tests).  Independent roots never interact, so the data path has no
collective: rank r plans the contiguous block ``shard_bounds(n, r, world)`` of the global root list with
random streams keyed by GLOBAL root index (results do not depend on the number of GPUs); the only
exchange is one all_gather of the per-root results (a few bytes per root -- latency-bound, a single
step on the fully connected xGMI mesh).  The reference's counterpart is one process per experiment
(``scripts/experiments.py:102-106``), with no result exchange at all.
"""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced block [lo, hi) of ``n_items`` for ``rank`` (first ``n % world`` ranks get one more)."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective_device():
    """Device the process group's collectives need their tensors on: the current GPU for nccl (= RCCL), CPU for gloo."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return None


def _group_active(force=False):
    """True when collectives must run: more than one rank, or ``force`` with an initialised (single-rank) group --
    the way a one-GPU box exercises RCCL's launch path on the product tensors (tests/test_gpu_distributed.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(force)


def all_gather_rows(local, n_total, device=None, force=False):
    """Gather row blocks (numpy [n_local, ...], sharded by :func:`shard_bounds`) into the full [n_total, ...] array
    on every rank.  Blocks are padded to the largest shard so that one fixed-size all_gather suffices."""
    local = np.ascontiguousarray(local)
    if not _group_active(force):
        return local
    return all_gather_packed({"x": local}, n_total, device=device, force=force)["x"]


def all_gather_packed(local, n_total, device=None, force=False):
    """Gather several per-root arrays with ONE collective: ``local`` maps names to row blocks [n_local, ...] (this
    rank's shard of ``n_total`` rows, :func:`shard_bounds`); the rows of all arrays are packed side by side into one
    ``uint8 [per, row_bytes]`` buffer (``per`` = the largest shard: fixed-size blocks), exchanged by one
    ``all_gather_into_tensor`` -- on the GPU when the group is RCCL, on the host for gloo -- and unpacked.  A planning
    call therefore costs one latency-bound exchange of a few bytes per root whatever the number of outputs."""
    import torch
    import torch.distributed as dist
    names = list(local)
    blocks = [np.ascontiguousarray(local[k]) for k in names]
    if not _group_active(force):
        return dict(zip(names, blocks))
    rank, world = rank_world()
    if device is None:
        device = collective_device()
    per = -(-n_total // world)
    n_local = blocks[0].shape[0]
    widths = [int(b.dtype.itemsize * int(np.prod(b.shape[1:], dtype=np.int64))) for b in blocks]
    packed = np.zeros((per, sum(widths)), dtype=np.uint8)
    off = 0
    for b, w in zip(blocks, widths):
        packed[:n_local, off:off + w] = b.reshape(n_local, -1).view(np.uint8).reshape(n_local, w)
        off += w
    t = torch.from_numpy(packed)
    if device is not None:
        t = t.to(device)
    out = torch.empty((world * per, packed.shape[1]), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    full = out.cpu().numpy().reshape(world, per, -1)
    keep = np.concatenate([full[r, :shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0]]
                           for r in range(world)], axis=0)
    res, off = {}, 0
    for k, b, w in zip(names, blocks, widths):
        res[k] = np.ascontiguousarray(keep[:, off:off + w]).view(b.dtype).reshape((n_total,) + b.shape[1:])
        off += w
    return res


def plan_batch_sharded(agent, root_states, root_steps=None, device=None, keys=("plans", "plan_len", "env_steps"),
                       force_collective=False):
    """Plan ``root_states`` (the same global list on every rank) with roots sharded over the process group.

    ``agent``: a tree-search agent of this package.  Returns a dict with the gathered ``keys`` (plus any of
    ``root_value`` / ``root_lower`` / ``root_upper`` the planner produced), identical on every rank and identical to the
    single-process result: random streams are keyed by the GLOBAL root index.  One collective per call."""
    rank, world = rank_world()
    root_states = np.asarray(root_states, dtype=np.int32)
    n = len(root_states)
    if n < world:       # decided identically on every rank BEFORE any collective: nobody is left waiting in one
        raise RuntimeError("fewer roots ({}) than ranks ({}): give every rank at least one root".format(n, world))
    lo, hi = shard_bounds(n, rank, world)           # (n >= world: no shard is empty)
    steps = None if root_steps is None else np.asarray(root_steps, dtype=np.int32)[lo:hi]
    rng = agent.planner.batch_rng_states(hi - lo, first_root=lo)
    if hasattr(agent, "planning_env"):
        env = agent.planning_env()
    else:
        from rl_agents_amd.agents.common.factory import preprocess_env
        env = preprocess_env(agent.env, agent.config["env_preprocessors"])
    local = agent.planner.plan_batch(env, root_states[lo:hi], steps, rng_states=rng)
    names = list(keys) + [k for k in ("root_value", "root_lower", "root_upper") if k in local]
    return all_gather_packed({k: local[k] for k in names}, n, device=device, force=force_collective)


def vi_solve_row_sharded(ctx, transition, reward, terminal=None, gamma=1.0, iterations=100, robust=False,
                         rtol=1e-5, atol=1e-8, rows=None):
    """Dense (robust) value iteration with source-state rows sharded over the process group.

    ``transition``: [S,A,S] or [M,S,A,S] -- either the full array (this rank slices its row block) or, with
    ``rows=(lo, hi)``, already this rank's block [.., hi-lo, A, S]; same for ``reward`` / ``terminal``.
    Each sweep: every rank backs up its rows on its GPU (``mp_vi_backup``), takes max_a, and the ranks
    all_gather V (8*S bytes) -- the one real exchange of the path; the ``allclose`` early exit of
    value_iteration.py:65-73 becomes an all_reduce(AND) of the per-rank tests.  Returns (Q [S,A], sweeps) on every rank.
    """
    rank, world = rank_world()
    t = np.asarray(transition)
    r = np.asarray(reward)
    n_states = t.shape[-1]
    if rows is None:
        lo, hi = shard_bounds(n_states, rank, world)
        t, r = t[..., lo:hi, :, :], r[..., lo:hi, :]
        term = None if (terminal is None or robust) else np.asarray(terminal).reshape(n_states)[lo:hi]
    else:
        lo, hi = rows
        term = None if (terminal is None or robust) else np.asarray(terminal).reshape(hi - lo)
    n_actions = r.shape[-1]
    model = ctx.load_dense_rows(t, r, term)
    v = np.zeros(n_states)
    q_local = np.zeros((hi - lo, n_actions))
    sweeps = 0
    for _ in range(int(iterations)):
        q_next = ctx.vi_backup(model, gamma, v, robust=robust)
        sweeps += 1
        close = bool(np.allclose(q_local, q_next, rtol=rtol, atol=atol))
        if world > 1:
            import torch
            import torch.distributed as dist
            flag = torch.tensor([1 if close else 0], dtype=torch.int32, device=collective_device() or "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            close = bool(flag.item())
        if close:
            break
        q_local = q_next
        v = all_gather_rows(q_local.max(axis=-1), n_states)
    model.close()
    return all_gather_rows(q_local, n_states), sweeps


def vi_solve_row_sharded_device(ctx, transition_rows, reward_rows, terminal_rows, n_states, rows, gamma=1.0,
                                iterations=100, robust=False, rtol=1e-5, atol=1e-8, check_every=8,
                                force_collective=False):
    """Device-resident form of :func:`vi_solve_row_sharded`: this rank's row block is already on the GPU.

    ``transition_rows`` / ``reward_rows``: torch CUDA tensors [.., hi-lo, A, S] / [.., hi-lo, A] (borrowed, not
    copied: at C5 size a block is 12.5 GB per model); ``terminal_rows``: torch uint8 [hi-lo] or None; ``rows`` =
    (lo, hi).  Every sweep stays on the device and NOTHING returns to the host per sweep: ``mp_vi_backup`` enqueues on
    the ctx stream, max_a / isclose run as torch ops, V is exchanged with one ``all_gather_into_tensor`` (RCCL; shards
    padded to equal length, re-ordered by one index op) and the ``allclose`` verdict with one 4-byte ``all_reduce``;
    once the verdict is "close" a device-side ``done`` flag freezes the iterate (the reference returns the PREVIOUS
    one, value_iteration.py:69-71) and the host only looks at that flag every ``check_every`` sweeps to leave the loop.
    ``ctx`` must enqueue on torch's current stream (``native.Context(device, torch.cuda.current_stream().cuda_stream)``).
    ``force_collective``: run the collectives also on a single-rank group (exercises RCCL on a one-GPU box).
    Returns (Q [S, A] tensor, sweeps)."""
    import torch
    import torch.distributed as dist
    rank, world = rank_world()
    grouped = _group_active(force_collective)   # world > 1, or a single-rank group whose collectives are to run
    lo, hi = rows
    dev = transition_rows.device
    n_actions = reward_rows.shape[-1]
    model = ctx.load_dense_rows(transition_rows, reward_rows, None if robust else terminal_rows)
    per = -(-n_states // world)                                # padded shard length
    v = torch.zeros(n_states, dtype=torch.float64, device=dev)
    v_pad = torch.zeros(world * per, dtype=torch.float64, device=dev)
    v_loc = torch.zeros(per, dtype=torch.float64, device=dev)
    even = world * per == n_states
    if grouped and not even:                                 # state s lives at v_pad[order[s]]
        order = torch.cat([torch.arange(r * per, r * per + (shard_bounds(n_states, r, world)[1] -
                                                           shard_bounds(n_states, r, world)[0]), device=dev)
                           for r in range(world)])
    q_local = torch.zeros((hi - lo, n_actions), dtype=torch.float64, device=dev)
    q_next = torch.empty_like(q_local)
    done = torch.zeros(1, dtype=torch.int32, device=dev)      # 1 once allclose held: the iterate is frozen
    sweeps_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    for it in range(int(iterations)):
        ctx.vi_backup(model, gamma, v, q_out=q_next, robust=robust)
        close = torch.isclose(q_local, q_next, rtol=rtol, atol=atol).all().to(torch.int32).reshape(1)
        if grouped:
            dist.all_reduce(close, op=dist.ReduceOp.MIN)
        active = 1 - done
        sweeps_dev += active                                   # a sweep counts until (and including) the close one
        advance = (active * (1 - close)).to(torch.bool)        # value = next only if not close and not frozen
        q_local = torch.where(advance, q_next, q_local)
        done = torch.maximum(done, close)
        v_loc.zero_()
        v_loc[:hi - lo] = q_local.max(dim=-1).values           # (unchanged once frozen)
        if grouped:
            dist.all_gather_into_tensor(v_pad, v_loc)
            v = v_pad if even else v_pad[order]
        else:
            v[lo:hi] = v_loc[:hi - lo]
        if (it + 1) % int(check_every) == 0 and bool(done.item()):
            break
    sweeps = int(sweeps_dev.item())
    model.close()
    if grouped:
        q_pad = torch.zeros((per, n_actions), dtype=torch.float64, device=dev)
        q_pad[:hi - lo] = q_local
        q_all = torch.empty((world * per, n_actions), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(q_all, q_pad)
        return (q_all if even else q_all[order]), sweeps
    return q_local, sweeps


class ShardedDevicePlan(object):
    """Device-resident form of :func:`plan_batch_sharded`: the same sharding and the same random streams (keyed by the
    GLOBAL root index), but roots, generator records and results are device buffers and the exchange never touches the
    host -- per ``plan()``: the planner's asynchronous batched launch on this rank's shard, ``mp_pack_rows`` (the shard's
    per-root rows into ONE byte matrix), ONE ``all_gather_into_tensor`` (RCCL over xGMI) and ``mp_unpack_rows`` (back into
    full per-root arrays on every rank).  The collective and the unpack run on a side stream, double-buffered, so a
    caller that plans again straight away overlaps them with its next launch; the returned ``ready`` event orders any
    consumer after them.  A ``gloo`` group (CPU tests, same-device dry runs) exchanges the same packed bytes through host
    memory.  The generator records stay resident and continue from call to call, as ``planner.np_random`` does for a
    single root (tree_search/abstract.py:124-131).

    ``payload``: what a root's row carries (per-root {plan[0..k], value, env_steps}).
      ``"compact"`` (default): ``{plans[:plan_entries] int32, value f64, env_steps i64}`` -- 20 B per root with
      ``plan_entries = 1`` (the action every caller of ``plan()`` executes, trainer/evaluation.py:168-180); the planner's
      per-root status rides in the top byte of the env-step word (env steps stay far below 2^56), so no row is spent on it;
      ``plan_len`` is not exchanged (the gathered ``plan_len`` is ``min(plan_len, plan_entries)`` of what was sent: 1 or 0);
      ``"full"``: ``{plans[max_plan_len], plan_len, value, env_steps, status}`` (56 B at ``max_plan_len`` 8).
    At 8 x 262 144 roots a step's exchange is 42 MB instead of 117 MB per rank.

    ``agent``: a tree-search agent whose planner has a device loop (``plan_batch_device``: MCTS, OPD)."""

    def __init__(self, agent, n_total, max_plan_len=None, force_collective=False, overlap=True, payload="compact", plan_entries=1,
                 time_exchange=False):
        import torch
        planner = agent.planner
        if getattr(planner, "plan_batch_device", None) is None or not planner.supports_device_loop():
            raise NotImplementedError("this agent's planner has no device-resident batched plan")
        if payload not in ("compact", "full"):
            raise ValueError("payload must be 'compact' or 'full'")
        self.agent, self.planner = agent, planner
        self.rank, self.world = rank_world()
        self.n = int(n_total)
        if self.n < self.world:
            raise RuntimeError("fewer roots ({}) than ranks ({}): give every rank at least one root".format(self.n, self.world))
        self.lo, self.hi = shard_bounds(self.n, self.rank, self.world)
        self.env = agent.planning_env()
        self.model = planner.model_for(self.env)
        self.ctx = ctx = planner.models.ctx
        self.dev = dev = torch.device("cuda", ctx.device)
        self.grouped = _group_active(force_collective)
        self.on_device = self.grouped and collective_device() is not None          # RCCL: tensors stay on the GPU
        self.mpl = mpl = int(max_plan_len or planner.device_plan_len(self.model))
        self.payload = payload
        self.sent = sent = mpl if payload == "full" else max(1, min(int(plan_entries), mpl))   # plan entries per exchanged row
        self.time_exchange = bool(time_exchange)
        nl = self.hi - self.lo
        self.per = per = -(-self.n // self.world)
        with torch.cuda.device(dev):
            self.ctx_stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev)
            self.comm = torch.cuda.Stream(device=dev) if (overlap and self.grouped) else self.ctx_stream
            self.overlapped = self.comm is not self.ctx_stream
            self.d_rng = torch.from_numpy(planner.batch_rng_states(nl, first_root=self.lo).view(np.int64)).to(dev)

            def local():
                return dict(plans=torch.full((nl, mpl), -1, dtype=torch.int32, device=dev),
                            plan_len=torch.zeros(nl, dtype=torch.int32, device=dev),
                            value=torch.zeros(nl, dtype=torch.float64, device=dev),
                            env_steps=torch.zeros(nl, dtype=torch.int64, device=dev),
                            status=torch.zeros(nl, dtype=torch.int32, device=dev),
                            head=torch.zeros((nl, sent), dtype=torch.int32, device=dev),       # compact rows: plans[:, :sent]
                            word=torch.zeros(nl, dtype=torch.int64, device=dev))              # compact rows: env_steps | status << 56

            def full():
                return dict(plans=torch.empty((self.n, sent), dtype=torch.int32, device=dev),
                            plan_len=torch.empty(self.n, dtype=torch.int32, device=dev),
                            value=torch.empty(self.n, dtype=torch.float64, device=dev),
                            env_steps=torch.empty(self.n, dtype=torch.int64, device=dev),
                            status=torch.empty(self.n, dtype=torch.int32, device=dev))
            self.keys = ("plans", "plan_len", "value", "env_steps", "status") if payload == "full" else ("head", "value", "word")
            # two buffer sets in every configuration: the buffers a plan() returns stay valid through the NEXT plan() and are
            # overwritten by the one after it (the ungrouped case used to hold a single set)
            nbuf = 2
            self.local = [local() for _ in range(nbuf)]
            self.row_bytes = 4 * mpl + 4 + 8 + 8 + 4 if payload == "full" else 4 * sent + 8 + 8
            if self.grouped:
                self.full = [full() for _ in range(nbuf)]
                self.packed = [torch.empty((per, self.row_bytes), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
                self.gathered = [torch.empty((self.world * per, self.row_bytes), dtype=torch.uint8, device=dev)
                                 for _ in range(nbuf)]
            self.kernel_done = [torch.cuda.Event(enable_timing=self.time_exchange) for _ in range(nbuf)]
            self.exchange_done = [torch.cuda.Event(enable_timing=self.time_exchange) for _ in range(nbuf)]
            self.gather_done = [None] * nbuf
            order = getattr(self.model, "action_order", None)
            self.order = None if order is None else torch.from_numpy(np.asarray(order, dtype=np.int32)).to(dev)
            torch.cuda.synchronize(dev)                   # the buffers exist before the ctx stream touches them
        self.turn = 0
        self._last_timed = None

    def shard(self, d_roots_all):
        """This rank's block of a global per-root device tensor."""
        return d_roots_all[self.lo:self.hi]

    def plan(self, d_root_states, d_root_steps=None):
        """``d_root_states``: int32 device tensor, either the GLOBAL root list [n_total] (this rank plans its block) or
        already this rank's block [hi - lo].  Only enqueues.  Returns the full per-root device tensors (identical on every
        rank) ``plans [n, plan entries exchanged]`` (environment action ids), ``plan_len``, ``value`` (root value / lower
        bound), ``env_steps``, ``status`` and ``ready``, the event after which they hold this call's results (``None`` when
        they are ready in ctx-stream order); they stay valid through the next ``plan()`` and are overwritten by the one
        after it.  Without a process group the tensors are this rank's (= all) roots' own result buffers."""
        import torch
        import torch.distributed as dist
        nl = self.hi - self.lo
        if d_root_states.shape[0] == self.n and self.n != nl:
            d_root_states = d_root_states[self.lo:self.hi]
            if d_root_steps is not None:
                d_root_steps = d_root_steps[self.lo:self.hi]
        if d_root_states.shape[0] != nl:
            raise ValueError("expected {} (global) or {} (this rank's) root states, got {}".format(self.n, nl, d_root_states.shape[0]))
        b = self.turn % len(self.local)
        self.turn += 1
        loc = self.local[b]
        if self.gather_done[b] is not None and self.comm is not self.ctx_stream:
            self.ctx_stream.wait_event(self.gather_done[b])        # buffer set b is free again
        self.planner.plan_batch_device(self.env, self.model, nl, d_root_states, d_root_steps, self.d_rng, loc["plans"],
                                       loc["plan_len"], loc["env_steps"], loc["status"], d_value=loc["value"])
        if not self.grouped:
            out = {k: loc[k] for k in ("plans", "plan_len", "value", "env_steps", "status")}
            if self.order is not None:
                with torch.cuda.stream(self.ctx_stream):
                    out["plans"] = torch.where(loc["plans"] >= 0, self.order[loc["plans"].clamp(min=0).long()], loc["plans"])
            out["ready"] = None
            return out
        if self.time_exchange:
            self.kernel_done[b].record(self.ctx_stream)            # the planner's kernel is done here: the exchange starts
        if self.payload == "compact":
            with torch.cuda.stream(self.ctx_stream):               # two small elementwise launches: the row's plan entries and
                loc["head"].copy_(loc["plans"][:, :self.sent])     # the env-step word with the status in its top byte
                torch.bitwise_or(loc["env_steps"], (-loc["status"]).to(torch.int64) << 56, out=loc["word"])
        arrays = [loc[k] for k in self.keys]
        self.ctx.pack_rows(arrays, nl, self.packed[b])
        full = self.full[b]
        outs = [full[k] for k in (self.keys if self.payload == "full" else ("plans", "value", "env_steps"))]
        if self.on_device:
            if self.comm is not self.ctx_stream:
                if not self.time_exchange:
                    self.kernel_done[b].record(self.ctx_stream)
                    self.comm.wait_event(self.kernel_done[b])
                else:                                           # (kernel_done was recorded BEFORE the pack: order after the pack)
                    ev = torch.cuda.Event()
                    ev.record(self.ctx_stream)
                    self.comm.wait_event(ev)
            with torch.cuda.stream(self.comm):
                dist.all_gather_into_tensor(self.gathered[b], self.packed[b])
            self.ctx.unpack_rows(self.gathered[b], self.n, self.world, outs, stream=self.comm.cuda_stream)
        else:                                                   # gloo: the same packed bytes, through host memory
            self.ctx.synchronize()
            host = self.packed[b].cpu()
            got = torch.empty((self.world * self.per, self.row_bytes), dtype=torch.uint8)
            dist.all_gather_into_tensor(got, host)
            with torch.cuda.stream(self.comm):
                self.gathered[b].copy_(got)
            self.ctx.unpack_rows(self.gathered[b], self.n, self.world, outs, stream=self.comm.cuda_stream)
        out = dict(full)
        with torch.cuda.stream(self.comm):
            if self.payload == "compact":                       # status back out of the env-step word; plan_len of what was sent
                word = full["env_steps"]
                full["status"].copy_((-(word >> 56)).to(torch.int32))
                word.bitwise_and_((1 << 56) - 1)
                full["plan_len"].copy_((full["plans"] >= 0).sum(dim=1).to(torch.int32))
            if self.order is not None:
                out["plans"] = torch.where(full["plans"] >= 0, self.order[full["plans"].clamp(min=0).long()], full["plans"])
            ev = self.exchange_done[b] if self.time_exchange else torch.cuda.Event()
            ev.record(self.comm)
        self.gather_done[b] = ev
        if self.time_exchange:
            self._last_timed = b
        out["ready"] = ev
        return out

    def last_exchange_ms(self):
        """Milliseconds from the end of the planner's kernel to the end of the unpack of the LAST plan() (pack + collective +
        unpack, as HIP events saw them; needs ``time_exchange=True``; synchronises).  None without a process group."""
        if not self.time_exchange or self._last_timed is None:
            return None
        b = self._last_timed
        self.exchange_done[b].synchronize()
        return float(self.kernel_done[b].elapsed_time(self.exchange_done[b]))

    def wait(self, out):
        """Block the host until ``out`` (a result of :meth:`plan`) is complete."""
        if out.get("ready") is not None:
            out["ready"].synchronize()
        else:
            self.ctx.synchronize()
        return out


def plan_batch_sharded_device(agent, d_root_states, d_root_steps=None, max_plan_len=None, force_collective=False):
    """One-shot form of :class:`ShardedDevicePlan` (fresh generator records keyed by global root index, as
    :func:`plan_batch_sharded`): plans the global device root list sharded over the group and returns the gathered
    device tensors after the exchange completed."""
    sp = ShardedDevicePlan(agent, int(d_root_states.shape[0]), max_plan_len=max_plan_len, force_collective=force_collective,
                           overlap=False, payload="full")
    return sp.wait(sp.plan(d_root_states, d_root_steps))
