"""Tree-search agent and planner bases (reference ``rl_agents/agents/tree_search/abstract.py``).

``AbstractTreeSearchAgent`` keeps the reference's receding-horizon logic (abstract.py:49-96) on the host;
``AbstractPlanner`` replaces the Python tree with a device call: ``plan(state, observation)`` extracts
the finite MDP behind the environment object, uploads it once (cached by content) and runs the batched
HIP planner for one root -- or ``plan_batch`` for many roots of the same model.
"""
import logging

import numpy as np

from rl_agents_amd import device_model, native
from rl_agents_amd.agents.common.abstract import AbstractAgent
from rl_agents_amd.agents.common.factory import preprocess_env
from rl_agents_amd.configuration import Configurable

logger = logging.getLogger(__name__)


def np_random(seed=None):
    """gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed))) and the entropy used."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


class AbstractTreeSearchAgent(AbstractAgent):
    PLANNER_TYPE = None

    def __init__(self, env, config=None):
        super(AbstractTreeSearchAgent, self).__init__(config)
        self.env = env
        self.planner = self.make_planner()
        self.previous_actions = []
        self.remaining_horizon = 0
        self.steps = 0

    @classmethod
    def default_config(cls):
        return {"env_preprocessors": [], "display_tree": False, "receding_horizon": 1, "terminal_reward": 0}

    def make_planner(self):
        if not self.PLANNER_TYPE:
            raise NotImplementedError()
        return self.PLANNER_TYPE(self.env, self.config)

    def plan(self, observation):
        """The planned action sequence from the environment's current state (abstract.py:49-68)."""
        self.steps += 1
        if self.step(self.previous_actions):
            actions = self.planner.plan(state=self.planning_env(), observation=observation)
        else:
            actions = self.previous_actions[1:]
        self.write_tree()
        self.previous_actions = actions
        return actions

    def write_tree(self):
        """With ``display_tree`` and a summary writer set (``set_writer``): the expanded tree of the last plan as an
        image (abstract.py:104-106; the plot is tree_search/graphics.py:115-166 restated in this package's graphics)."""
        if self.config["display_tree"] and self.writer:
            from rl_agents_amd.agents.tree_search.graphics import TreePlot
            TreePlot(self.planner, max_depth=6).plot_to_writer(self.writer, epoch=self.steps, show=True)

    def step(self, actions):
        """Receding-horizon bookkeeping; True when a new plan is required (abstract.py:70-82)."""
        replan = self.remaining_horizon == 0 or len(actions) <= 1
        if replan:
            self.remaining_horizon = self.config["receding_horizon"] - 1
        else:
            self.remaining_horizon -= 1
        self.planner.step_tree(actions)
        return replan

    def plan_batch(self, root_states, root_steps=None):
        """Plans for many independent roots of this agent's environment model in one launch.

        ``root_states``: state indices. Returns the planner's batch result (dict of arrays)."""
        return self.planner.plan_batch(self.planning_env(), root_states, root_steps)

    def planning_env(self):
        """The environment object the planner plans on (abstract.py:59-61: the preprocessed copy of ``self.env``)."""
        return preprocess_env(self.env, self.config["env_preprocessors"])

    def reset(self):
        self.planner.step_by_reset()
        self.remaining_horizon = 0
        self.steps = 0

    def seed(self, seed=None):
        return self.planner.seed(seed)

    def record(self, state, action, reward, next_state, done, info):
        pass

    def act(self, state):
        return self.plan(state)[0]

    def save(self, filename):
        return False

    def load(self, filename):
        return False


class Node(object):
    """One node of an exported device tree, with what the reference's tree consumers use (tree_search/graphics.py:19-37,
    115-166; abstract.py:212-358): ``children`` (dict keyed by action label, creation order), ``parent``, ``count``,
    ``get_value()``, ``depth``, ``planner``, ``path()`` / ``sequence()``, ``breadth_first_search``, ``get_trajectories``,
    ``get_obs_visits``.  Read-only: the tree itself lives on the device; this is its picture after a plan."""

    def __init__(self, parent, action, count, value, depth, planner=None):
        self.parent = parent
        self.action = action
        self.children = {}
        self.count = count
        self.value = value
        self.depth = depth
        self.planner = planner

    def get_value(self):
        return self.value

    def is_leaf(self):
        return not self.children

    def selection_rule(self):
        """Greedy child (no exploration) as the viewers highlight it (tree_search/graphics.py:71): for UCT nodes the
        most visited child, ties to the larger value (mcts.py:212-218); for OPD nodes the largest lower bound
        (deterministic.py:21-26, first maximum: a viewer must not consume the planner's random stream)."""
        if not self.children:
            return None
        actions = list(self.children.keys())
        if hasattr(self, "value_lower"):
            return max(actions, key=lambda a: self.children[a].value_lower)
        top = max(self.children[a].count for a in actions)
        return max((a for a in actions if self.children[a].count == top), key=lambda a: self.children[a].get_value())

    def selection_strategy(self, temperature):
        """UCT exploration score of this node under its parent (mcts.py:275-286)."""
        if self.parent is None:
            return self.get_value()
        return self.get_value() + temperature * len(self.parent.children) * getattr(self, "prior", 1.0) / (self.count + 1)

    def path(self):
        """Action labels from the root to this node (abstract.py:269-281; a list -- the reference returns an iterator)."""
        node, actions = self, []
        while node.parent is not None:
            actions.append(node.action)
            node = node.parent
        return actions[::-1]

    def sequence(self):
        """Nodes from the root to this node (abstract.py:283-293)."""
        node, nodes = self, [self]
        while node.parent is not None:
            node = node.parent
            nodes.append(node)
        return nodes[::-1]

    @staticmethod
    def all_argmax(x):
        x = np.asarray(x)
        return np.flatnonzero(x == x.max())

    @staticmethod
    def breadth_first_search(root, operator=None, condition=None, condition_blocking=True):
        """Level-order traversal (abstract.py:246-265): yields ``operator(node, path)`` -- or ``(node, path)`` -- for every
        node that meets ``condition`` (all of them without one); with ``condition_blocking`` the subtree below a node that
        met the condition is not explored.  ``path`` = the action labels from ``root``."""
        from collections import deque
        todo = deque([(root, [])])
        while todo:
            node, path = todo.popleft()
            met = condition is None or bool(condition(node))
            if met:
                yield operator(node, path) if operator else (node, path)
            if condition is None or not condition_blocking or not met:
                for key, child in node.children.items():
                    todo.append((child, path + [key]))

    def get_trajectories(self, full_trajectories=True, include_leaves=True):
        """The nodes of this subtree as the reference lists them (abstract.py:319-339): with ``full_trajectories`` one node
        sequence per root-to-leaf path (this node first); otherwise the single nodes, every child's subtree before the
        node itself; leaves only with ``include_leaves``."""
        if full_trajectories:
            out = []
            stack = [(self, [self])]
            while stack:                                    # depth first, children in dict order
                node, seq = stack.pop()
                if node.children:
                    for child in reversed(list(node.children.values())):
                        stack.append((child, seq + [child]))
                elif include_leaves:                        # (without leaves every path ends in nothing: [])
                    out.append(seq)
            return out
        out = []

        def visit(node):                                    # post-order: children first, then the node
            if node.children:
                for child in node.children.values():
                    visit(child)
                out.append(node)
            elif include_leaves:
                out.append(node)
        import sys
        limit = sys.getrecursionlimit()
        sys.setrecursionlimit(max(limit, 10000))
        try:
            visit(self)
        finally:
            sys.setrecursionlimit(limit)
        return out

    def get_obs_visits(self, state=None):
        """How often each observation is the one of an EXPANDED node of this subtree (abstract.py:341-358) ->
        (visits, updates), dicts keyed by ``str(observation)``.  Nodes that carry their ``observation`` (the optimistic
        planners' nodes: the state reached) are counted directly.  Otherwise the reference replays every node's action
        path on a copy of the environment ``state``; here the state a node is reached in is known from the device model
        (``node.state``: root state + transition table, no stepping), and only trees exported without it (CartPole) replay
        on copies of ``state``.  As in the reference a node with an EMPTY path -- the root -- is booked under the
        observation of the node listed before it (its loop variable is simply not reassigned)."""
        from collections import defaultdict
        visits, updates = defaultdict(int), defaultdict(int)
        nodes = self.get_trajectories(full_trajectories=False, include_leaves=False)
        if hasattr(self, "observation"):
            for node in nodes:
                if hasattr(node, "observation"):
                    visits[str(node.observation)] += 1
                    if hasattr(node, "updates_count"):
                        updates[str(node.observation)] += node.updates_count
            return visits, updates
        observation, bound = None, False
        for node in nodes:
            path = node.path()
            if path:
                if hasattr(node, "state"):
                    observation = node.state
                else:
                    import copy
                    env = copy.deepcopy(getattr(state, "unwrapped", state))
                    for action in path:
                        observation = env.step(action)[0]
                bound = True
            if not bound:
                raise NameError("the root is the only expanded node: the reference's replay has no observation to book it under")
            visits[str(observation)] += 1
        return visits, updates

    def __str__(self):
        return "{} (n:{}, v:{:.2f})".format(self.path(), self.count, self.get_value())


def build_tree(arrays, value_key, extra=(), prior=None, planner=None, transition=None, root_state=None):
    """Creation-order arrays (parent, action, count, <value_key>, ...) -> linked :class:`Node` objects.
    ``prior``: per-action prior probabilities, attached to the nodes as ``node.prior`` (mcts.py:237-246).
    ``transition`` [S, A] (environment action ids) + ``root_state``: every node also gets ``node.state``, the state its
    action sequence reaches (the observation a replay of ``node.path()`` would end on)."""
    parent, action = arrays["parent"], arrays["action"]
    nodes = []
    for i in range(len(parent)):
        par = nodes[parent[i]] if parent[i] >= 0 else None
        node = Node(par, int(action[i]), int(arrays["count"][i]), float(arrays[value_key][i]),
                    0 if par is None else par.depth + 1, planner)
        if transition is not None and root_state is not None:
            node.state = int(root_state) if par is None else int(transition[par.state, int(action[i])])
        for name in extra:
            setattr(node, name, arrays[name][i].item())
        if prior is not None:
            node.prior = 1.0 if par is None else float(prior[int(action[i])])
        if par is not None:
            par.children[int(action[i])] = node
        nodes.append(node)
    return nodes[0]


class AbstractPlanner(Configurable):
    """Device-backed planner: holds the numpy Generator (source of truth of the random stream), the
    model cache and the last batch result."""

    def __init__(self, config=None):
        super(AbstractPlanner, self).__init__(config)
        self.np_random = None
        self.models = device_model.ModelCache()
        self.last = None          # dict of arrays returned by the last device call
        self.env_steps = 0        # counterpart of len(planner.observations) (abstract.py:158-161)
        self._root = None
        self.reset()
        self.seed()

    @classmethod
    def default_config(cls):
        return dict(budget=500, gamma=0.8, step_strategy="reset")

    def seed(self, seed=None):
        self.np_random, seed = np_random(seed)
        self._entropy = seed
        return [seed]

    # -- to be provided by subclasses ---------------------------------------------------------------
    def plan_batch(self, state, root_states, root_steps=None, rng_states=None):
        raise NotImplementedError()

    def export_tree(self, root=0):
        raise NotImplementedError()

    # -------------------------------------------------------------------------------------------------
    supports_cartpole = False
    supports_restricted_actions = True
    # device-resident evaluation loop (trainer/batched_evaluation.py): planners that can plan a batch whose roots,
    # generator records and results all live in device buffers provide plan_batch_device(...)
    plan_batch_device = None

    def supports_device_loop(self):
        return False

    def raise_for_device_status(self, d_status, live):
        pass

    def model_for(self, state):
        if device_model.is_cartpole(state):
            if not self.supports_cartpole:
                raise TypeError("this planner needs a finite-MDP environment")
            return self.models.get_cartpole(getattr(state, "unwrapped", state).cartpole_params())
        mdp = device_model.finite_mdp_of(state)
        if mdp.mode != "deterministic":
            raise TypeError("tree search on the device needs a deterministic finite MDP, got mode '{}'".format(mdp.mode))
        available, order = device_model.availability_of(state, mdp)
        if available is not None and not self.supports_restricted_actions:
            raise NotImplementedError("this planner does not handle environments that restrict the available actions")
        return self.models.get(device_model.spec_from_mdp(mdp, max_steps=device_model.env_max_steps(state),
                                                          available=available, action_order=order))

    # An environment that lists its actions in a non-ascending order is planned on in the permuted action space (see
    # device_model.TableSpec.action_order): labels are mapped back here, at the planner's boundary.
    @staticmethod
    def action_order(model):
        return getattr(model, "action_order", None)

    @classmethod
    def relabel(cls, out, model):
        """Device labels -> the environment's action ids in a batch result (plans, per-action root statistics)."""
        order = cls.action_order(model)
        if order is None:
            return out
        plans = out["plans"]
        out["plans"] = np.where(plans >= 0, order[np.maximum(plans, 0)], -1).astype(plans.dtype)
        for k in ("root_child_count", "root_child_value"):
            if k in out:
                back = np.empty_like(out[k])
                back[:, order] = out[k]
                out[k] = back
        return out

    @classmethod
    def device_actions(cls, actions, model):
        """Environment action ids -> device labels (for re-rooting kept trees)."""
        order = cls.action_order(model)
        if order is None:
            return actions
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))
        return inv[np.asarray(actions, dtype=np.int64)]

    @classmethod
    def relabel_tree(cls, arrays, model):
        order = cls.action_order(model)
        if order is not None:
            act = arrays["action"]
            arrays["action"] = np.where(act >= 0, order[np.maximum(act, 0)], act).astype(act.dtype)
        return arrays

    def plan(self, state, observation):
        """Plan from the current state of the environment object ``state`` (never stepped, never copied)."""
        s0, steps0 = device_model.env_root_state(state)
        rng = native.rng_state_from_generator(self.np_random).reshape(1, 6)
        out = self.plan_batch(state, [s0], [steps0], rng_states=rng)
        native.generator_set_state(self.np_random, rng[0])
        n = int(out["plan_len"][0])
        return [int(a) for a in out["plans"][0, :n]]

    def get_plan(self):
        if self.last is None:
            return []
        n = int(self.last["plan_len"][0])
        return [int(a) for a in self.last["plans"][0, :n]]

    @property
    def root(self):
        """Root :class:`Node` of the last plan's tree (root 0 of the batch), exported on demand -- or already, when
        another planner of the process was about to reuse the device workspaces (:meth:`claim_device_tree`)."""
        if self._root is None and self.last is not None:
            self._root = self.export_tree(0)
        return self._root

    def get_visits(self):
        """Observations the planner stepped through (abstract.py:163-167).  The device logs no observations; planners whose
        tree holds every state they stepped into (the optimistic planners) derive it from the last plan's tree."""
        raise NotImplementedError("this planner's rollout steps leave no trace on the device: get_visits is not available "
                                  "(use planner.root.get_obs_visits for the states of the expanded nodes)")

    def get_updates(self):
        from collections import defaultdict
        return defaultdict(int)

    # The trees of the last plan live in workspaces of the process-wide device context, shared by every planner of the
    # process: a planner claims them when it plans and may only read them back while the claim still stands.
    def about_to_plan(self):
        """Called before a device plan overwrites the context's tree workspaces: the planner that owns them exports the
        tree of its last plan first (root 0, once), so its ``planner.root`` keeps answering after another agent of the
        process has planned -- e.g. benchmark mode with ``display_tree``."""
        ctx = self.models.ctx
        prev = getattr(ctx, "_tree_owner", None)
        if prev is not None and prev is not self:
            prev.preserve_tree()
            ctx._tree_owner = None

    def claim_device_tree(self):
        self.models.ctx._tree_owner = self

    def preserve_tree(self):
        if self.last is not None and self._root is None:
            try:
                self._root = self.export_tree(0)
            except Exception as e:          # an export must never break the planner that is about to plan
                logger.warning("could not keep the tree of the previous planner: %s", e)

    def owns_device_tree(self):
        return getattr(self.models.ctx, "_tree_owner", None) is self

    def require_device_tree(self):
        if not self.owns_device_tree():
            raise RuntimeError("the tree of this planner's last plan is no longer on the device: another planner of the "
                               "process has planned since (export `planner.root` before planning with another agent)")

    def step_tree(self, actions):
        strategy = self.config["step_strategy"]
        if strategy == "reset":
            self.step_by_reset()
        elif strategy == "subtree":
            if actions:
                self.step_by_subtree(actions[0])
            else:
                self.step_by_reset()
        else:
            logger.warning("Unknown step strategy: %s", strategy)
            self.step_by_reset()

    def step_by_reset(self):
        self.reset()

    def step_by_subtree(self, action):
        """Keep the subtree under the root's child ``action`` for the next plan (abstract.py:195-206)."""
        raise NotImplementedError("step_strategy 'subtree' is not available for this planner on the device")

    def reset(self):
        self.last = None
        self._root = None

    def batch_rng_states(self, n_roots, first_root=0):
        """PCG64 records for a batch: root i draws from Generator(PCG64(SeedSequence([entropy, i])))."""
        # numpy's SeedSequence hashing + PCG64 seeding restated in C on the host (mp_seed_sequence_states; compared with
        # numpy in tests/test_host_logic.py): 262 144 records in 30 ms instead of seconds of Python constructors
        return native.seed_sequence_states([int(self._entropy) % (1 << 63)], first_root, n_roots)
