"""Discrete robust planning on the MI355X planning core (reference ``rl_agents/agents/robust/robust.py:9-71``).

``DiscreteRobustPlannerAgent`` plans over a finite set of candidate models of the environment: ``config["models"]`` is
a list of preprocessor lists, each turning the true environment into one model (robust.py:67-70); the planner maximises
the worst case over the models (``RobustNode``: ``np.min`` of per-model bounds, robust.py:42-49).  Here every model must
be a deterministic finite MDP over the same states and actions: their tables go to the device as one joint model
(``mp_model_load_joint``) and the expansions run in ``mp_ropd_plan`` (rl_agents_amd/csrc/ropd.hip).

Note on the reference: its ``JointEnv.step`` (robust.py:13-16) still returns the old 4-tuple, which
``DeterministicNode.expand`` (deterministic.py:41) cannot unpack, so ``DiscreteRobustPlannerAgent`` raises on current
environments.  The planner classes themselves run unmodified over a joint environment with a 5-tuple step; that is how
the golden vectors this agent reproduces were made (tests/golden/gen/make_golden_robust.py).
"""
import hashlib
import logging

import numpy as np

from rl_agents_amd import device_model, native
from rl_agents_amd.agents.common.factory import preprocess_env
from rl_agents_amd.agents.tree_search.abstract import build_tree
from rl_agents_amd.agents.tree_search.deterministic import DeterministicPlannerAgent, OptimisticDeterministicPlanner

logger = logging.getLogger(__name__)


class JointEnv(object):
    """The environments of all models, stepped together (robust.py:9-26) -- with the 5-tuple ``step`` of the
    environments it wraps.  The device planner never steps it: it reads the models' tables and current states."""

    def __init__(self, envs):
        self.joint_state = envs

    def step(self, action):
        transitions = [state.step(action) for state in self.joint_state]
        observations, rewards, terminals, truncated, info = zip(*transitions)
        return observations, np.array(rewards), np.array(terminals), np.array(truncated), info

    @property
    def action_space(self):
        return self.joint_state[0].action_space

    def get_available_actions(self):
        return list(set().union(*[s.get_available_actions() if hasattr(s, "get_available_actions")
                                  else range(s.action_space.n)
                                  for s in self.joint_state]))


class DiscreteRobustPlanner(OptimisticDeterministicPlanner):
    """robust.py:28-40 for one or many roots of one set of models."""

    def __init__(self, env, config=None):
        super(DiscreteRobustPlanner, self).__init__(env, config)
        self._joint = {}

    def joint_model(self, state):
        """Device model of a :class:`JointEnv` (or any object with ``joint_state``: a list of finite-MDP envs)."""
        envs = getattr(state, "joint_state", None)
        if not envs:
            raise TypeError("the discrete robust planner plans on a JointEnv of at least one model")
        mdps = [device_model.finite_mdp_of(e) for e in envs]
        masks = []
        for e, mdp in zip(envs, mdps):
            if mdp.mode != "deterministic":
                raise TypeError("every model must be a deterministic finite MDP, got mode '{}'".format(mdp.mode))
            masks.append(device_model.available_actions_of(e, mdp))
        shapes = {np.asarray(m.transition).shape for m in mdps}
        if len(shapes) != 1:
            raise ValueError("all models must share the state and action spaces, got tables of shapes {}".format(shapes))
        t = np.ascontiguousarray(np.stack([np.asarray(m.transition, dtype=np.int64) for m in mdps]))
        r = np.ascontiguousarray(np.stack([np.asarray(m.reward, dtype=np.float64) for m in mdps]))
        term = np.ascontiguousarray(np.stack([np.asarray(m.terminal).reshape(-1).astype(np.uint8) for m in mdps]))
        rules = {getattr(m, "done_rule", "source") for m in mdps}
        if len(rules) != 1:     # one terminal convention per joint model (the first model's used to win silently)
            raise ValueError("all models of a joint environment must share one done_rule, got {}".format(sorted(rules)))
        rule = rules.pop()
        # JointEnv.get_available_actions (robust.py:22-25): the union over the models of what each lists in its own
        # state; a model whose env has no get_available_actions lists everything
        available = None
        if any(m is not None for m in masks):
            available = np.ascontiguousarray(np.stack([np.ones(t.shape[1:], dtype=bool) if m is None else m for m in masks]))
        h = hashlib.blake2b(digest_size=16)
        for arr in (t, r, term) + (() if available is None else (available,)):
            h.update(arr.view(np.uint8).reshape(-1))
        key = (t.shape, rule, h.hexdigest())
        model = self._joint.get(key)
        if model is None:
            if len(self._joint) >= 4:
                for old in self._joint.values():
                    old.close()
                self._joint.clear()
            model = self._joint[key] = self.models.ctx.load_joint(t, r, term, done_rule=rule, available=available)
        return model, [int(m.state) for m in mdps]

    def plan(self, state, observation):
        model, joint_state = self.joint_model(state)
        rng = native.rng_state_from_generator(self.np_random).reshape(1, 6)
        out = self.plan_batch(state, [joint_state], rng_states=rng, model=model)
        native.generator_set_state(self.np_random, rng[0])
        n = int(out["plan_len"][0])
        return [int(a) for a in out["plans"][0, :n]]

    def plan_batch(self, state, root_states, root_steps=None, rng_states=None, model=None):
        """root_states: [n] (all models start in the same state) or [n, M] joint states."""
        if model is None:
            model, _ = self.joint_model(state)
        rs = np.asarray(root_states, dtype=np.int32)
        n = rs.shape[0]
        if rng_states is None:
            rng_states = self.batch_rng_states(n)
        cfg = self.config
        budget = int(cfg["budget"])
        if cfg["gamma"] == 1 and budget >= model.A:
            raise ZeroDivisionError("float division by zero")       # gamma ** depth / (1 - gamma), deterministic.py:53
        self.about_to_plan()
        out = self.models.ctx.ropd_plan(model, rs, budget, cfg["gamma"], cfg.get("terminal_reward", 0), rng_states,
                                        max_plan_len=budget // model.A + 1)
        if (out["status"] == native.ERR_REWARD_RANGE).any():
            raise ValueError("This planner assumes that all rewards are normalized in [0, 1]")  # deterministic.py:46-47
        out["rng_states"] = rng_states
        self.last, self._root, self._last_actions, self._last_models = out, None, model.A, model.M
        self.claim_device_tree()
        self.env_steps += int(out["env_steps"].sum())
        return out

    # -- device-resident evaluation loop (BatchedEvaluation): the planner plans on the JOINT model, the loop steps the TRUE env
    plans_on_joint_env = True

    def supports_device_loop(self):
        return type(self).plan_batch is DiscreteRobustPlanner.plan_batch

    def begin_device_loop(self):
        """A new device-resident evaluation run: forget the joint model held for the previous one."""
        self._device_joint = None

    def plan_batch_device(self, state, model, n, d_state, d_steps, d_rng, d_plans, d_len, d_env_steps, d_status, d_value=None):
        """One asynchronous batched plan (mp_ropd_plan, MP_MEM_DEVICE): every model starts in the episode's state (the agent
        rebuilds its JointEnv from the true env before every plan, robust.py:67-70).  ``model`` (the true env's, which the loop
        steps) is not used; ``state`` is the JointEnv."""
        import torch
        cfg = self.config
        # the joint model of this loop's JointEnv, resolved ONCE: joint_model() stacks and hashes every candidate model's
        # tables on the host, O(M S A) work that would make every step of an asynchronous loop host-bound;
        # the evaluation loop announces itself with begin_device_loop()
        held = getattr(self, "_device_joint", None)
        if held is None or held[0] is not state:
            held = self._device_joint = (state, self.joint_model(state)[0])
        jm = held[1]
        budget = int(cfg["budget"])
        if cfg["gamma"] == 1 and budget >= jm.A:
            raise ZeroDivisionError("float division by zero")
        ctx = self.models.ctx
        if getattr(self, "_d_joint", None) is None or self._d_joint.shape != (n, jm.M) or self._d_joint.device != d_state.device:
            self._d_joint = torch.empty((n, jm.M), dtype=torch.int32, device=d_state.device)
        with torch.cuda.stream(torch.cuda.ExternalStream(ctx.stream_ptr(), device=d_state.device)):
            self._d_joint.copy_(d_state[:, None].expand(n, jm.M))          # (on the planner's stream, after the env step)
        self.about_to_plan()
        ctx.ropd_plan_device(jm, n, self._d_joint, budget, cfg["gamma"], cfg.get("terminal_reward", 0), d_rng,
                             int(d_plans.shape[1]), plans=d_plans, plan_len=d_len, root_lower=d_value, env_steps=d_env_steps,
                             status=d_status)
        self._last_actions, self._last_models = jm.A, jm.M
        self.claim_device_tree()
        self.last, self._root = None, None

    def export_tree(self, root=0):
        self.require_device_tree()
        a, m = self._last_actions, self._last_models
        arrays = self.models.ctx.ropd_tree(root, 1 + (int(self.config["budget"]) // a) * a, m)
        lower, upper = arrays["lower"], arrays["upper"]
        arrays["value_lower_min"], arrays["value_upper_min"] = lower.min(axis=1), upper.min(axis=1)
        tree = build_tree(arrays, "value_upper_min", extra=("value_lower_min", "value_upper_min"), planner=self)
        # per-model vectors as the reference's nodes hold them: ndarrays on leaves, the backed-up scalars once expanded
        by_id = [tree] + [None] * (len(arrays["parent"]) - 1)
        for i in range(1, len(by_id)):       # creation order: parents come first
            by_id[i] = by_id[int(arrays["parent"][i])].children[int(arrays["action"][i])]
        for i, node in enumerate(by_id):
            expanded = arrays["first_child"][i] >= 0 or i == 0
            node.value_lower = float(lower[i, 0]) if expanded else lower[i].copy()
            node.value_upper = float(upper[i, 0]) if expanded else upper[i].copy()
            node.reward, node.done = arrays["reward"][i].copy(), arrays["done"][i].astype(bool)
            node.observation = tuple(int(s) for s in arrays["state"][i])
        return tree


class DiscreteRobustPlannerAgent(DeterministicPlannerAgent):
    """Drop-in for ``rl_agents.agents.robust.robust.DiscreteRobustPlannerAgent``."""
    PLANNER_TYPE = DiscreteRobustPlanner

    def __init__(self, env, config=None):
        self.true_env = env
        super(DiscreteRobustPlannerAgent, self).__init__(env, config)

    @classmethod
    def default_config(cls):
        config = super(DiscreteRobustPlannerAgent, cls).default_config()
        config.update(dict(models=[]))
        return config

    def joint_env(self):
        """The candidate models of the true environment, stepped together (robust.py:67-70)."""
        return JointEnv([preprocess_env(self.true_env, preprocessors) for preprocessors in self.config["models"]])

    def planning_env(self):
        self.env = self.joint_env()
        return preprocess_env(self.env, self.config["env_preprocessors"])

    def plan(self, observation):
        return super(DiscreteRobustPlannerAgent, self).plan(observation)
