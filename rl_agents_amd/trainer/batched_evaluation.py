"""Batched evaluation: N episodes advanced in lock-step, one batched ``plan`` call per step.

The reference evaluates one environment per process (``Evaluation.run_episodes``,
``rl_agents/trainer/evaluation.py:139-194``; ``scripts/experiments.py:102-106`` fans processes out).  With a
device planner the natural caller is the opposite: many episodes of one finite MDP side by side, their current
states being the roots of a single launch.  Per step, for every live episode i:

    actions = agent.plan(observation)        (evaluation.py:168)   ->  one row of planner.plan_batch(...)
    env.step(actions[0])                     (evaluation.py:180)   ->  a table lookup, vectorised over episodes

Episode i draws from its own PCG64 stream -- the one a sequential ``Evaluation`` would give a fresh agent
seeded with ``seed + i`` (evaluation.py:375 seeds the agent with ``sim_seed + episode``) -- and the stream
continues from step to step exactly as ``planner.np_random`` does, so a batched run reproduces N sequential
runs action for action (tests/test_gpu_agents.py).

Planners that carry state from one ``plan()`` to the next -- MCTS with ``step_strategy="subtree"`` (the kept trees),
the state-aware planner (its state values and state-node lists) -- keep that state per batch slot on the device, so
for them every step plans the FULL batch (finished episodes stay where they ended and their results are ignored):
slot i then sees exactly the sequence of plans a sequential agent i would make.
"""
import time

import numpy as np

from rl_agents_amd import device_model, native
from rl_agents_amd.agents.common.factory import preprocess_env
from rl_agents_amd.agents.tree_search.abstract import np_random


class BatchedEvaluation(object):
    def __init__(self, env, agent, num_episodes=64, sim_seed=0, max_steps=None, device_resident="auto", sharded=False,
                 check_every=8, env_seed=0):
        """``env``: a finite-MDP environment (template of every episode); ``agent``: a tree-search or value-iteration
        agent of this package built on it.  ``max_steps``: episode length cap (defaults to the env's ``max_steps``, else
        100).

        ``device_resident``: keep the whole loop on the GPU -- root states, step counters, generator records, returns and
        the action log live in device buffers, the planner runs in its asynchronous device mode and the environments are
        stepped by ``mp_env_step`` on the planner's own root-state buffer: NO host round trip per step (the host looks at
        one 4-byte counter every ``check_every`` steps to leave the loop).  ``"auto"``: when the planner supports it
        (MCTS without tree re-use / closed loop, OPD, value-iteration agents), else the host-stepped loop; ``True``
        raises if it does not.  ``sharded``: the episodes are split over the ranks of the process group
        (:func:`rl_agents_amd.distributed.shard_bounds`; episode i keeps seed ``sim_seed + i`` whatever the number of
        ranks) and the per-episode results are gathered with ONE collective at the end.

        Stochastic environments (``stochastic`` / ``sparse`` finite MDPs; MCTS and value-iteration agents): episode i steps
        with its OWN env generator ``Generator(PCG64(SeedSequence([env_seed, i])))`` -- the reference's ``Evaluation`` never
        seeds the env (evaluation.py:372-376), so the convention is this class's -- sampled exactly as
        ``FiniteMDPEnv.step`` does, and every plan's clones start from that generator as it is at that step."""
        self.env, self.agent = env, agent
        self.num_episodes = int(num_episodes)
        self.sim_seed = sim_seed
        self.device_resident, self.sharded, self.check_every = device_resident, bool(sharded), int(check_every)
        mdp = device_model.finite_mdp_of(env)
        if mdp.mode not in ("deterministic", "stochastic", "sparse"):
            raise TypeError("batched evaluation steps a finite MDP")
        self.mdp, self.stochastic, self.env_seed = mdp, mdp.mode != "deterministic", int(env_seed)
        self.transition = np.asarray(mdp.transition)
        self.reward = np.asarray(mdp.reward)
        self.terminal = np.asarray(mdp.terminal, dtype=bool)
        self.done_rule = getattr(mdp, "done_rule", "source")
        self.initial_state = int(getattr(getattr(env, "unwrapped", env), "config", {}).get("state", mdp.state))
        self.max_steps = int(max_steps or device_model.env_max_steps(env) or 100)

    def run(self, initial_states=None):
        """Run all episodes to termination / truncation. Returns dict(returns, discounted_returns, lengths, actions, fps,
        plan_seconds, planner_env_steps) -- with ``sharded`` the per-episode arrays cover ALL episodes on every rank."""
        first, n = 0, self.num_episodes
        starts = (np.full(n, self.initial_state, dtype=np.int32) if initial_states is None
                  else np.asarray(initial_states, dtype=np.int32).copy())
        if self.sharded:
            from rl_agents_amd.distributed import rank_world, shard_bounds
            rank, world = rank_world()
            if n < world:
                raise RuntimeError("fewer episodes ({}) than ranks ({})".format(n, world))
            first, hi = shard_bounds(n, rank, world)
            out = self._run_local(starts[first:hi], first)
            return self._gather(out, n)
        return self._run_local(starts, 0)

    def _gather(self, out, n_total):
        """One packed all_gather of the per-episode results; rates are summed over the ranks' own clocks."""
        from rl_agents_amd.distributed import all_gather_packed, rank_world
        keys = ("returns", "discounted_returns", "lengths", "actions")
        full = all_gather_packed({k: out[k] for k in keys}, n_total)
        rank, world = rank_world()
        scal = np.array([[out["fps"], out["plan_seconds"], float(out["planner_env_steps"])]])
        scal = all_gather_packed({"s": scal}, world)["s"] if world > 1 else scal
        return dict(full, fps=float(scal[:, 0].sum()), plan_seconds=float(scal[:, 1].max()),
                    planner_env_steps=int(scal[:, 2].sum()), device_resident=out.get("device_resident", False))

    def _env_generators(self, first, n):
        """Episode i's env generator (stochastic models): Generator(PCG64(SeedSequence([env_seed, first + i])))."""
        return [np.random.Generator(np.random.PCG64(np.random.SeedSequence([self.env_seed, first + i]))) for i in range(n)]

    def _env_step(self, idx, s, act, gens):
        """reward, next state, done of env.step(act) for the live episodes `idx` in states `s` (FiniteMDPEnv.step)."""
        r = self.reward[s, act]
        if self.stochastic:
            s_next = np.array([self.mdp.next_state(int(si), int(ai), np_random=gens[int(i)]) for i, si, ai in zip(idx, s, act)],
                              dtype=np.int32).reshape(len(idx))
        else:
            s_next = self.transition[s, act].astype(np.int32)
        done = self.terminal[s] if self.done_rule == "source" else self.terminal[s_next]
        return r, s_next, done

    def _device_capable(self):
        agent = self.agent
        if hasattr(agent, "get_state_action_value") and not hasattr(agent, "planner"):
            return True
        planner = getattr(agent, "planner", None)
        return planner is not None and getattr(planner, "plan_batch_device", None) is not None and \
            planner.supports_device_loop()

    def _run_local(self, starts, first):
        use_device = self.device_resident is True or (self.device_resident == "auto" and self._device_capable())
        if use_device:
            if not self._device_capable():
                raise NotImplementedError("this agent's planner has no device-resident evaluation loop")
            return self._run_device(starts, first)
        return self._run_host(starts, first)

    # ---------------------------------------------------------------------------------------- device-resident loop
    def _run_device(self, starts, first):
        import torch
        agent = self.agent
        vi = not hasattr(agent, "planner")
        n, T = len(starts), self.max_steps
        env = agent.planning_env() if hasattr(agent, "planning_env") else self.env
        if vi:
            models = device_model.ModelCache()
            model = models.get(device_model.spec_from_mdp(device_model.finite_mdp_of(self.env),
                                                          max_steps=device_model.env_max_steps(self.env)))
            if self.stochastic:
                model.set_episode_rules(self.done_rule, device_model.env_max_steps(self.env))
            ctx = models.ctx
        else:
            planner = agent.planner
            # (a planner of the discrete robust agent plans on its JointEnv of candidate models; what the loop steps is the true env)
            joint = getattr(planner, "plans_on_joint_env", False)
            model = planner.model_for(preprocess_env(self.env, agent.config["env_preprocessors"]) if joint else env)
            ctx = planner.models.ctx
            if hasattr(planner, "begin_device_loop"):
                planner.begin_device_loop()
            if joint and getattr(model, "action_order", None) is not None:
                # the plans come from the JOINT model in the env's action ids; re-labelling them through the stepped
                # model's listing order would map ids that are already ids
                raise NotImplementedError("the device-resident loop of the discrete robust planner steps a true environment "
                                          "that lists its actions in ascending order; use device_resident=False")
        dev = torch.device("cuda", ctx.device)
        order = getattr(model, "action_order", None)
        mpl = 1 if vi else planner.device_plan_len(model)
        d_state = torch.from_numpy(np.ascontiguousarray(starts, dtype=np.int32)).to(dev)
        d_steps = torch.zeros(n, dtype=torch.int32, device=dev)
        d_alive = torch.ones(n, dtype=torch.uint8, device=dev)
        d_ret = torch.zeros(n, dtype=torch.float64, device=dev)
        d_disc = torch.zeros(n, dtype=torch.float64, device=dev)
        d_log = torch.full((n, T), -1, dtype=torch.int32, device=dev)
        d_plans = torch.full((n, mpl), -1, dtype=torch.int32, device=dev)
        d_len = torch.zeros(n, dtype=torch.int32, device=dev)
        d_es = torch.zeros((T, n), dtype=torch.int64, device=dev)          # planner env steps, per step and episode
        d_status = torch.zeros((T, n), dtype=torch.int32, device=dev)
        d_nalive = torch.zeros(1, dtype=torch.int32, device=dev)
        gamma = float(agent.config.get("gamma", 1))
        d_gpow = torch.from_numpy(gamma ** np.arange(T, dtype=np.float64) if T else np.zeros(0)).to(dev)
        # the stream a sequential Evaluation gives agent i: np_random(sim_seed + i) (evaluation.py:375)
        d_rng = torch.from_numpy(native.seed_sequence_states((), self.sim_seed + first, n).view(np.int64)).to(dev)
        d_q = None
        if vi:
            d_q = torch.from_numpy(np.ascontiguousarray(agent.get_state_action_value(), dtype=np.float64)).to(dev)
        # planners that carry state from plan to plan (kept UCT trees, the state-aware planner's values and lists): every
        # episode starts with a new planner object, every step plans the FULL batch (finished slots are ignored)
        subtree = (not vi) and planner.config.get("step_strategy") == "subtree" and hasattr(planner, "step_by_subtree")
        if not vi and (subtree or getattr(planner, "carries_state", False)):
            planner.step_by_reset()
            if hasattr(planner, "forget"):
                planner.forget()
        d_prev = torch.zeros(n, dtype=torch.int32, device=dev)
        d_erng = None
        if self.stochastic:                                    # the episodes' own env generators, resident on the device
            d_erng = torch.from_numpy(np.stack([native.rng_state_from_generator(g) for g in self._env_generators(first, n)])
                                      .view(np.int64)).to(dev)
        skw = dict(d_env_rng=d_erng) if self.stochastic else {}
        torch.cuda.synchronize(dev)                            # the buffers exist before the ctx stream touches them
        t0 = time.perf_counter()
        t = 0
        while t < T:
            if vi:
                ctx.greedy_actions_device(d_q, d_state, d_plans)
            elif subtree:                                      # AbstractPlanner.step_tree -> step_by_subtree(actions[0])
                planner.plan_batch_device(env, model, n, d_state, d_steps, d_rng, d_plans, d_len, d_es[t], d_status[t],
                                          keep_actions=d_prev if t else None, **skw)
            else:
                planner.plan_batch_device(env, model, n, d_state, d_steps, d_rng, d_plans, d_len, d_es[t], d_status[t], **skw)
            if self.stochastic:
                ctx.env_step_stochastic_device(model, d_state, d_steps, d_alive, d_plans, T, d_gpow, d_ret, d_disc, d_log,
                                               d_nalive, d_erng)
            else:
                ctx.env_step_device(model, d_state, d_steps, d_alive, d_plans, T, d_gpow, d_ret, d_disc, d_log, d_nalive)
            if subtree:
                self._copy_first_actions(ctx, d_plans, d_prev)
            t += 1
            if t % self.check_every == 0 or t == T:
                ctx.synchronize()
                if int(d_nalive.item()) == 0:
                    break
        ctx.synchronize()
        wall = time.perf_counter() - t0
        lengths = d_steps.cpu().numpy()
        live = torch.arange(T, device=dev)[:, None] < d_steps[None, :].to(torch.int64)      # step t of episode i was played
        if not vi:
            planner.raise_for_device_status(d_status, live)
        if (d_log.cpu().numpy()[live.cpu().numpy().T] < 0).any():   # a live episode was handed an empty plan (budget < |A|, no
            raise Exception("The agent did not plan any action")     # episodes): Evaluation.step raises (evaluation.py:168-170)
        if not vi:
            if subtree or getattr(planner, "carries_state", False):
                planner.env_steps += int(d_es[:t].sum().item())   # (stateful planners plan every slot at every step, as the
            else:                                                #  host-stepped loop does: finished slots count there too)
                planner.env_steps += int((d_es * live).sum().item())
        actions = d_log.cpu().numpy()
        if order is not None:                                  # the device planned in the env's listing order
            actions = np.where(actions >= 0, np.asarray(order)[np.maximum(actions, 0)], -1).astype(np.int32)
        return dict(returns=d_ret.cpu().numpy(), discounted_returns=d_disc.cpu().numpy(), lengths=lengths, actions=actions,
                    fps=float(lengths.sum()) / wall, plan_seconds=wall,
                    planner_env_steps=0 if vi else int(planner.env_steps), device_resident=True)

    @staticmethod
    def _copy_first_actions(ctx, d_plans, d_prev):
        """d_prev <- d_plans[:, 0] on the planner's stream (the actions the kept trees are re-rooted under next step)."""
        import torch
        with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_ptr(), device=d_plans.device)):
            d_prev.copy_(d_plans[:, 0])

    # ---------------------------------------------------------------------------------------- host-stepped loop
    def _run_host(self, starts, first):
        n = len(starts)
        if not hasattr(self.agent, "planner"):
            return self._run_host_vi(starts, first)
        planner = self.agent.planner
        states = np.asarray(starts, dtype=np.int32).copy()
        steps = np.zeros(n, dtype=np.int32)
        alive = np.ones(n, dtype=bool)
        returns = np.zeros(n)
        gamma_returns = np.zeros(n)
        gamma = float(self.agent.config.get("gamma", 1))
        rng = native.seed_sequence_states((), self.sim_seed + first, n)    # np_random(sim_seed + i), evaluation.py:375
        actions_log = np.full((n, self.max_steps), -1, dtype=np.int32)
        env = preprocess_env(self.env, self.agent.config["env_preprocessors"])
        if getattr(planner, "plans_on_joint_env", False):      # the discrete robust agent: plans on its candidate models
            env = self.agent.planning_env()
        self._gens = self._env_generators(first, n) if self.stochastic else None
        subtree = planner.config.get("step_strategy") == "subtree" and hasattr(planner, "step_by_subtree")
        stateful = subtree or getattr(planner, "carries_state", False)
        if stateful:
            planner.step_by_reset()
            if hasattr(planner, "forget"):
                planner.forget()                               # every episode starts with a new planner object
        previous = None                                        # first actions of the previous step's plans (all slots)
        planner.defer_errors = stateful                        # errors of finished slots must not end the run
        t0, plan_seconds, env_steps = time.perf_counter(), 0.0, 0
        try:
            return self._loop(planner, env, n, stateful, subtree, previous, states, steps, alive, returns, gamma_returns, gamma,
                              rng, actions_log, t0)
        finally:
            planner.defer_errors = False

    def _loop(self, planner, env, n, stateful, subtree, previous, states, steps, alive, returns, gamma_returns, gamma, rng,
              actions_log, t0):
        plan_seconds, env_steps = 0.0, 0
        while alive.any():
            idx = np.arange(n) if stateful else np.flatnonzero(alive)
            sub_rng = np.ascontiguousarray(rng[idx])
            t1 = time.perf_counter()
            kw = {}
            if self.stochastic:                                # the clones of every plan copy the env's generator (factory.py:119-134)
                kw["env_rng_states"] = np.stack([native.rng_state_from_generator(self._gens[int(i)]) for i in idx])
            if subtree:                                        # AbstractPlanner.step_tree -> step_by_subtree(actions[0])
                out = planner.plan_batch(env, states[idx], steps[idx], rng_states=sub_rng, keep_actions=previous, **kw)
            else:
                out = planner.plan_batch(env, states[idx], steps[idx], rng_states=sub_rng, **kw)
            plan_seconds += time.perf_counter() - t1
            rng[idx] = sub_rng
            act = out["plans"][:, 0].astype(np.int64)
            if stateful and "status" in out and hasattr(planner, "raise_for_status"):
                planner.raise_for_status(np.asarray(out["status"])[alive[idx]])   # what a sequential agent would raise
            if (act[alive[idx]] < 0).any():                    # Evaluation.step (evaluation.py:168-170) raises on an empty plan
                raise Exception("The agent did not plan any action")
            act[act < 0] = 0                                   # (episodes that are over: their plans are ignored)
            if stateful:
                previous = act.astype(np.int32)
                live = alive[idx]
                idx, act = idx[live], act[live]
            s = states[idx]
            r, s_next, done = self._env_step(idx, s, act, self._gens)
            actions_log[idx, steps[idx]] = act
            gamma_returns[idx] += r * gamma ** steps[idx]
            returns[idx] += r
            states[idx] = s_next
            steps[idx] += 1
            env_steps += len(idx)
            alive[idx] = ~(done | (steps[idx] >= self.max_steps))
        wall = time.perf_counter() - t0
        return dict(returns=returns, discounted_returns=gamma_returns, lengths=steps.copy(), actions=actions_log,
                    fps=env_steps / wall, plan_seconds=plan_seconds, planner_env_steps=planner.env_steps,
                    device_resident=False)

    def _run_host_vi(self, starts, first=0):
        """Value-iteration agents on the host-stepped loop: act = argmax Q[state] (value_iteration.py:35), vectorised."""
        q = np.asarray(self.agent.get_state_action_value())
        n, T = len(starts), self.max_steps
        states, steps, alive = np.asarray(starts, dtype=np.int32).copy(), np.zeros(n, np.int32), np.ones(n, bool)
        returns, gamma_returns = np.zeros(n), np.zeros(n)
        gamma = float(self.agent.config.get("gamma", 1))
        actions_log = np.full((n, T), -1, dtype=np.int32)
        gens = self._env_generators(first, n) if self.stochastic else None
        t0, env_steps = time.perf_counter(), 0
        while alive.any():
            idx = np.flatnonzero(alive)
            s = states[idx]
            act = np.argmax(q[s], axis=1)
            r, s_next, done = self._env_step(idx, s, act, gens)
            actions_log[idx, steps[idx]] = act
            gamma_returns[idx] += r * gamma ** steps[idx]
            returns[idx] += r
            states[idx] = s_next
            steps[idx] += 1
            env_steps += len(idx)
            alive[idx] = ~(done | (steps[idx] >= T))
        wall = time.perf_counter() - t0
        return dict(returns=returns, discounted_returns=gamma_returns, lengths=steps.copy(), actions=actions_log,
                    fps=env_steps / wall, plan_seconds=0.0, planner_env_steps=0, device_resident=False)


# --------------------------------------------------------------------------------------------- benchmark mode
def load_env(env_config):
    """An environment from a reference-style env config: a dict or the path of a JSON file holding the finite-MDP
    tables (``scripts/configs/FiniteMDPEnv/**/env_*.json``: mode / transition / reward / terminal / max_steps, next to
    the ``id`` / ``import_module`` keys that name the absent ``finite_mdp`` package)."""
    import json
    from rl_agents_amd.envs import FiniteMDPEnv, MaskedFiniteMDPEnv
    if not isinstance(env_config, dict):
        with open(env_config) as f:
            env_config = json.load(f)
    cfg = {k: v for k, v in env_config.items() if k in ("mode", "transition", "reward", "terminal", "next", "max_steps",
                                                         "state", "done_rule", "available")}
    if "transition" not in cfg:
        raise ValueError("batched benchmark: the environment config must hold finite-MDP tables")
    env = (MaskedFiniteMDPEnv if "available" in cfg else FiniteMDPEnv)(cfg)
    env.reset()
    return env


def generate_agent_configs(benchmark_config):
    """``scripts/experiments.py:118-143`` without the temporary files: a ``base_agent`` config varied over ``values`` of
    ``key`` becomes the ``agents`` list (dicts)."""
    from rl_agents_amd.agents.common.factory import load_agent_config
    agents = list(benchmark_config.get("agents", []))
    if "base_agent" in benchmark_config:
        base = benchmark_config["base_agent"]
        base = dict(base) if isinstance(base, dict) else load_agent_config(base)
        agents += [dict(base, **{benchmark_config["key"]: value}) for value in benchmark_config["values"]]
    return agents


def batched_benchmark(benchmark_config, episodes=64, seed=0, max_steps=None):
    """The reference's benchmark mode (``scripts/experiments.py:85-116``: the product of ``environments`` x ``agents``,
    one process per experiment, ``--episodes`` sequential episodes each) on the device planners: every experiment
    becomes ONE :class:`BatchedEvaluation` whose ``episodes`` episodes -- seeded ``seed + i`` like
    ``Evaluation(sim_seed=seed)`` seeds episode i -- share each step's batched ``plan`` launch.  Returns one summary dict per
    experiment, in product order (the reference writes the run directories of its evaluations to a summary file)."""
    from itertools import product
    from rl_agents_amd.agents.common.factory import load_agent
    if not isinstance(benchmark_config, dict):
        import json
        with open(benchmark_config) as f:
            benchmark_config = json.load(f)
    results = []
    for env_config, agent_config in product(benchmark_config["environments"], generate_agent_configs(benchmark_config)):
        env = load_env(env_config)
        agent = load_agent(agent_config, env)
        out = BatchedEvaluation(env, agent, num_episodes=episodes, sim_seed=seed, max_steps=max_steps).run()
        results.append(dict(environment=env_config if not isinstance(env_config, dict) else "<dict>",
                            agent=agent_config if not isinstance(agent_config, dict) else
                            {k: v for k, v in agent_config.items() if not isinstance(v, (list, dict)) or k == "__class__"},
                            episodes=int(episodes), mean_return=float(out["returns"].mean()),
                            mean_discounted_return=float(out["discounted_returns"].mean()),
                            mean_length=float(out["lengths"].mean()), fps=float(out["fps"]),
                            plan_seconds=float(out["plan_seconds"]), returns=out["returns"], lengths=out["lengths"],
                            actions=out["actions"]))
    return results
