"""Batched evaluation of N environments THAT EACH OWN THEIR FINITE MDP -- and change it at every step.

The north-star environment is "highway-v0 as a finite MDP": ``to_finite_mdp()`` rebuilds the environment's own
time-to-collision table from the current traffic at EVERY step, and every agent of the reference re-reads it there
(``ValueIterationAgent.act`` re-converts and re-solves, dynamic_programming/value_iteration.py:29-35; the tree-search agents
step deep copies of the live env, tree_search/abstract.py:59-62).  The reference evaluates one such environment per process
(``Evaluation.run_episodes``, trainer/evaluation.py:139-194; ``scripts/experiments.py:102-106`` fans processes out).

Here N environments advance in lock-step and share ONE launch per step:

    per step, for the live episodes:   mdp_i = env_i.to_finite_mdp()                 (host: the reference's own extraction)
                                       batch model <- the tables that changed        (mp_model_update_tables: the delta of
                                                                                      a batch is each episode's own table)
                                       ONE batched plan, one MDP per root            (mp_vi_solve_batch + argmax /
                                                                                      mp_uct_plan_models / mp_opd_plan_models)
                                       env_i.step(action_i)                          (host: the environments are host objects)

Episode i draws from the generator a sequential ``Evaluation`` would give agent i (``np_random(sim_seed + i)``,
evaluation.py:375), continued from step to step, so the batch reproduces N sequential (environment, agent) loops action for
action (tests/test_gpu_per_episode_eval.py) -- and the unmodified reference's, on tests/golden/per_episode.npz.

Supported agents of this package: ``ValueIterationAgent``, ``MCTSAgent`` (open loop, ``step_strategy="reset"``),
``DeterministicPlannerAgent``; environments: deterministic finite MDPs of one (S, A) shape, with or without restricted /
re-ordered action sets (``get_available_actions``: the restriction must be the same table for every episode, as it is for
grids of one shape -- device_model.availability_of).
"""
import time

import numpy as np

from rl_agents_amd import device_model, native


class PerEpisodeEvaluation(object):
    def __init__(self, envs, agent, sim_seed=0, max_steps=None):
        """``envs``: N environments exposing a deterministic finite MDP (``.mdp`` or ``to_finite_mdp()``) of one shape;
        ``agent``: an agent of this package built on ``envs[0]`` -- the template: its class and configuration are what every
        episode's agent would be.  ``max_steps``: episode length cap (default: the env's ``max_steps``, else 100)."""
        self.envs, self.agent = list(envs), agent
        if not self.envs:
            raise ValueError("at least one environment")
        self.n = len(self.envs)
        self.sim_seed = sim_seed
        self.max_steps = int(max_steps or device_model.env_max_steps(self.envs[0]) or 100)
        self.vi = not hasattr(agent, "planner")
        planner = None if self.vi else agent.planner
        self.kind = "vi" if self.vi else ("uct" if hasattr(planner, "prior_policy") else "opd")
        if not self.vi:
            cfg = planner.config
            if cfg.get("step_strategy", "reset") != "reset":
                raise NotImplementedError("per-episode tables: the tree of the previous step was built on the previous table; "
                                          "step_strategy must be 'reset'")
            if self.kind == "uct" and (cfg.get("closed_loop") or getattr(planner, "policy_source", None) is not None):
                raise NotImplementedError("per-episode tables: open-loop MCTS with the planner's own policies")
        self.ctx = (device_model.ModelCache().ctx if self.vi else planner.models.ctx)
        self.model = None
        self.uploads = 0            # MDP tables sent to the device (initial load + per-step deltas)
        self._tables = None

    # ------------------------------------------------------------------------------------------------ model extraction
    def _extract(self, i):
        """(transition [S,A] in device column order, reward, terminal, availability or None, listing order or None, state,
        steps) of environment i as it is now."""
        env = self.envs[i]
        mdp = device_model.finite_mdp_of(env)
        if mdp.mode != "deterministic":
            raise TypeError("per-episode evaluation plans on deterministic finite MDPs, got mode '{}'".format(mdp.mode))
        available, order = (None, None) if self.vi else device_model.availability_of(env, mdp)
        if self.kind == "uct":
            self.agent.planner._env_order = order           # (what MCTS.model_for notes for restricted_policy_tables)
            if order is not None and self.agent.planner.prior_policy["type"] == "random":
                order = None                                # policy type `random` lists np.arange(n) (mcts.py:46-57)
        spec = device_model.spec_from_mdp(mdp, max_steps=device_model.env_max_steps(env), available=available, action_order=order)
        return spec, int(mdp.state), int(getattr(getattr(env, "unwrapped", env), "steps", 0) or 0)

    def _sync_model(self, live):
        """Bring the batch model up to date with the live episodes' environments; returns (states, steps)."""
        n = self.n
        states, steps = np.zeros(n, np.int32), np.zeros(n, np.int32)
        specs = [None] * n
        for i in live:
            specs[i], states[i], steps[i] = self._extract(i)
        first_spec = specs[live[0]]
        if self.model is None:
            # every slot needs a table: finished episodes cannot be among them at the first step
            s, a = first_spec.n_states, first_spec.n_actions
            self._tables = dict(t=np.zeros((n, s, a), np.int64), r=np.zeros((n, s, a), np.float64), term=np.zeros((n, s), np.uint8))
            for i in live:
                self._check_shape(specs[i], first_spec)
                self._tables["t"][i], self._tables["r"][i], self._tables["term"][i] = specs[i].transition, specs[i].reward, specs[i].terminal
            self.model = self.ctx.load_table_batch(self._tables["t"], self._tables["r"], self._tables["term"],
                                                   done_rule=first_spec.done_rule, max_steps=first_spec.max_steps)
            self.model.action_order = first_spec.action_order
            self._available = None if first_spec.available is None else first_spec.available.astype(bool)
            if self._available is not None and self.kind == "opd":
                # (MCTS reads availability through its policies, mcts.py:59-97: see _uct_policy)
                self.model.set_available(np.tile(self._available, (n, 1)))
            self.uploads += len(live)
            return states, steps
        changed = []
        tb = self._tables
        for i in live:
            sp = specs[i]
            self._check_shape(sp, first_spec)
            if not (np.array_equal(tb["t"][i], sp.transition) and np.array_equal(tb["r"][i].view(np.uint64), sp.reward.view(np.uint64))
                    and np.array_equal(tb["term"][i], sp.terminal)):
                tb["t"][i], tb["r"][i], tb["term"][i] = sp.transition, sp.reward, sp.terminal
                changed.append(i)
        # contiguous runs of changed episodes: one mp_model_update_tables each
        k = 0
        while k < len(changed):
            j = k
            while j + 1 < len(changed) and changed[j + 1] == changed[j] + 1:
                j += 1
            lo, hi = changed[k], changed[j] + 1
            self.model.update_tables(lo, tb["t"][lo:hi], tb["r"][lo:hi], tb["term"][lo:hi])
            k = j + 1
        self.uploads += len(changed)
        if changed and getattr(self, "_policy", None) is not None:      # fused policy records hold the old transitions
            self._policy.close()
            self._policy = None
        return states, steps

    def _check_shape(self, spec, first):
        if spec.reward.shape != first.reward.shape:
            raise ValueError("every environment of the batch must have the same number of states and actions")
        same_order = (spec.action_order is None) == (first.action_order is None) and \
            (spec.action_order is None or np.array_equal(spec.action_order, first.action_order))
        same_avail = (spec.available is None) == (first.available is None) and \
            (spec.available is None or np.array_equal(spec.available, first.available))
        if not (same_order and same_avail and spec.done_rule == first.done_rule and spec.max_steps == first.max_steps):
            raise NotImplementedError("per-episode evaluation: the episodes' environments must list / restrict their actions "
                                      "identically and share done_rule / max_steps")

    # ------------------------------------------------------------------------------------------------ one batched plan
    def _uct_policy(self):
        """Per-state policy tables over the GLOBAL states for environments that restrict their actions (the same [S, A]
        rows for every episode), re-fused whenever a table changed."""
        planner = self.agent.planner
        if getattr(self, "_policy", None) is None:
            self.model.available = self._available
            prior, rollout, listed, slots = planner.restricted_policy_tables(self.model, self._available)
            n = self.n
            tile = lambda x: None if x is None else np.tile(x, (n, 1))       # noqa: E731
            self._policy = self.ctx.load_policy(self.model, tile(prior), tile(rollout), listed=tile(listed), rollout_slots=tile(slots))
        return self._policy

    def _plan(self, live, states, steps):
        """First actions (environment action ids) of the live episodes + planner env steps of this plan."""
        from rl_agents_amd.agents.tree_search.mcts import policy_probabilities
        agent, model = self.agent, self.model
        idx = np.asarray(live, dtype=np.int32)
        if self.vi:
            q, _ = self.ctx.vi_solve_batch(model, agent.config["gamma"], agent.config["iterations"])
            return np.array([int(np.argmax(q[i, states[i]])) for i in live], dtype=np.int64), 0   # value_iteration.py:35
        planner = agent.planner
        cfg = planner.config
        rng = np.ascontiguousarray(self.rng[idx])
        if self.kind == "uct":
            if self._available is not None:
                out = self.ctx.uct_plan(model, idx * model.S_each + states[idx], cfg["episodes"], cfg["horizon"], cfg["gamma"],
                                        cfg["temperature"], None, None, rng, root_steps=steps[idx], max_plan_len=1,
                                        policy=self._uct_policy())
            else:
                pp, rp = policy_probabilities(planner.prior_policy, model.A), policy_probabilities(planner.rollout_policy, model.A)
                out = self.ctx.uct_plan(model, states[idx], cfg["episodes"], cfg["horizon"], cfg["gamma"], cfg["temperature"], pp, rp,
                                        rng, root_steps=steps[idx], max_plan_len=1, model_index=idx)
        else:
            budget = int(cfg["budget"])
            if cfg["gamma"] == 1 and budget >= model.A:
                raise ZeroDivisionError("float division by zero")           # gamma ** depth / (1 - gamma), deterministic.py:53
            out = self.ctx.opd_plan(model, states[idx], budget, cfg["gamma"], cfg.get("terminal_reward", 0), rng, max_plan_len=1,
                                    model_index=idx)
            if (out["status"] == native.ERR_REWARD_RANGE).any():
                raise ValueError("This planner assumes that all rewards are normalized in [0, 1]")     # deterministic.py:46-47
        self.rng[idx] = rng
        if (out["plan_len"] < 1).any():
            raise Exception("The agent did not plan any action")             # evaluation.py:168-170
        first = out["plans"][:, 0].astype(np.int64)
        order = getattr(model, "action_order", None)
        if order is not None:
            first = np.asarray(order, dtype=np.int64)[first]
        return first, int(out["env_steps"].sum())

    # ------------------------------------------------------------------------------------------------ the loop
    def run(self):
        """Run every episode to termination / truncation (each environment is reset first).  Returns dict(returns,
        discounted_returns, lengths, actions [N, max_steps] (-1 padded), wall / extract / update / plan seconds,
        planner_env_steps, uploads)."""
        n, T = self.n, self.max_steps
        for env in self.envs:
            env.reset()
        self.rng = native.seed_sequence_states((), self.sim_seed, n)        # np_random(sim_seed + i), evaluation.py:375
        alive = np.ones(n, dtype=bool)
        returns, disc, lengths = np.zeros(n), np.zeros(n), np.zeros(n, np.int64)
        actions = np.full((n, T), -1, dtype=np.int32)
        gamma = float(self.agent.config.get("gamma", 1))
        t_extract = t_plan = 0.0
        planner_steps = 0
        t0 = time.perf_counter()
        for t in range(T):
            live = [int(i) for i in np.flatnonzero(alive)]
            if not live:
                break
            c0 = time.perf_counter()
            states, steps = self._sync_model(live)
            c1 = time.perf_counter()
            acts, es = self._plan(live, states, steps)
            self.ctx.synchronize()
            c2 = time.perf_counter()
            t_extract += c1 - c0
            t_plan += c2 - c1
            planner_steps += es
            for i, a in zip(live, acts):
                _, reward, terminated, truncated, _ = self.envs[i].step(int(a))
                returns[i] += reward
                disc[i] += reward * gamma ** t
                actions[i, t] = int(a)
                lengths[i] += 1
                if terminated or truncated or lengths[i] >= T:
                    alive[i] = False
        wall = time.perf_counter() - t0
        if not self.vi:
            self.agent.planner.env_steps += planner_steps
        return dict(returns=returns, discounted_returns=disc, lengths=lengths, actions=actions, wall_seconds=wall,
                    extract_update_seconds=t_extract, plan_seconds=t_plan, planner_env_steps=planner_steps, uploads=self.uploads,
                    fps=float(lengths.sum()) / wall if wall > 0 else 0.0)

    def close(self):
        if getattr(self, "_policy", None) is not None:
            self._policy.close()
            self._policy = None
        if self.model is not None:
            self.model.close()
            self.model = None
