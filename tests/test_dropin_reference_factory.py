"""The drop-in claim, executed: the REFERENCE's own agent_factory / load_agent (rl_agents/agents/common/factory.py:12-56)
builds this package's agents from JSON that differs from a reference agent config in the ``__class__`` string only, and
the configuration surfaces match (default_config keys and default values of the agent and of its planner:
tree_search/abstract.py:35-41,119-122, mcts.py:22-31,121-127, value_iteration.py:24-27, robust_value_iteration.py:15-19,
state_aware.py:85-91, robust/robust.py:60-63).

Build-container only (the reference lives at /root/reference and is imported through the generator's stub modules, in a
subprocess so that the stubs never leak into this test session); CPU only: nothing here reaches a native call.
"""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = r'''
import json, os, sys, tempfile
sys.dont_write_bytecode = True
repo, ref = sys.argv[1], sys.argv[2]
sys.path[:0] = [os.path.join(repo, "tests", "golden", "gen", "stubs"), ref, repo]
from rl_agents.agents.common.factory import agent_factory, load_agent          # the REFERENCE's factory
from rl_agents_amd.envs import FiniteMDPEnv, generators

def env():
    e = FiniteMDPEnv(generators.gridworld()); e.reset(); return e

def cls(path):
    return "<class '{}'>".format(path)

PAIRS = [  # reference class, this package's class, keys whose DEFAULT VALUE differs by documented design
    ("rl_agents.agents.tree_search.mcts.MCTSAgent", "rl_agents_amd.agents.tree_search.mcts.MCTSAgent", set()),
    ("rl_agents.agents.tree_search.deterministic.DeterministicPlannerAgent",
     "rl_agents_amd.agents.tree_search.deterministic.DeterministicPlannerAgent", set()),
    ("rl_agents.agents.tree_search.state_aware.StateAwarePlannerAgent",
     "rl_agents_amd.agents.tree_search.state_aware.StateAwarePlannerAgent", set()),
    ("rl_agents.agents.robust.robust.DiscreteRobustPlannerAgent",
     "rl_agents_amd.agents.robust.robust.DiscreteRobustPlannerAgent", set()),
    ("rl_agents.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent",
     "rl_agents_amd.agents.tree_search.mcts_with_prior.MCTSWithPriorPolicyAgent", {"prior_agent"}),  # DQN -> VI prior
]
report = {}
for ref_path, our_path, differs in PAIRS:
    theirs_cls = getattr(__import__(ref_path.rsplit(".", 1)[0], fromlist=["x"]), ref_path.rsplit(".", 1)[1])
    ours_cls = getattr(__import__(our_path.rsplit(".", 1)[0], fromlist=["x"]), our_path.rsplit(".", 1)[1])
    a, b = theirs_cls.default_config(), ours_cls.default_config()
    assert set(a) == set(b), (ref_path, set(a) ^ set(b))
    for k in a:
        if k not in differs:
            assert a[k] == b[k], (ref_path, k, a[k], b[k])
    if "MCTSWithPrior" in ref_path:
        continue        # both need a constructible prior agent (the reference's default is its torch DQN)
    # instantiate through the reference factory: dict config and JSON file (load_agent), only __class__ changed
    ref_agent = agent_factory(env(), {"__class__": cls(ref_path), "budget": 60, "gamma": 0.75})
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump({"__class__": cls(our_path), "budget": 60, "gamma": 0.75}, f)
    ours = load_agent(f.name, env())
    os.unlink(f.name)
    assert type(ours) is ours_cls
    assert set(ours.config) == set(ref_agent.config), (ref_path, set(ours.config) ^ set(ref_agent.config))
    # the agent surface Evaluation / the tests call (trainer/evaluation.py:89-90,133,168,190,301-302,375-376)
    for name in ("act", "plan", "reset", "seed", "record", "save", "load", "eval", "set_writer", "set_directory", "set_time"):
        assert callable(getattr(ours, name)), (our_path, name)
    assert ours.seed(3) == ref_agent.seed(3) == [3]
    pc_ref, pc_ours = ref_agent.planner.config, ours.planner.config
    assert set(pc_ref) == set(pc_ours), (ref_path, set(pc_ref) ^ set(pc_ours))
    for k in pc_ref:
        if k != "__class__":            # the one line of the JSON that differs
            assert pc_ref[k] == pc_ours[k], (ref_path, k, pc_ref[k], pc_ours[k])
    report[our_path.rsplit(".", 1)[1]] = sorted(pc_ours)
# value-iteration agents solve at construction (device call): configuration surface only
for ref_path, our_path in (("rl_agents.agents.dynamic_programming.value_iteration.ValueIterationAgent",
                            "rl_agents_amd.agents.dynamic_programming.value_iteration.ValueIterationAgent"),
                           ("rl_agents.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent",
                            "rl_agents_amd.agents.dynamic_programming.robust_value_iteration.RobustValueIterationAgent")):
    theirs_cls = getattr(__import__(ref_path.rsplit(".", 1)[0], fromlist=["x"]), ref_path.rsplit(".", 1)[1])
    ours_cls = getattr(__import__(our_path.rsplit(".", 1)[0], fromlist=["x"]), our_path.rsplit(".", 1)[1])
    a, b = theirs_cls.default_config(), ours_cls.default_config()
    assert set(a) <= set(b) and all(a[k] == b[k] for k in a), (ref_path, a, b)
    report[our_path.rsplit(".", 1)[1]] = sorted(set(b) - set(a))     # extra keys (documented): e.g. the prior temperature
print(json.dumps(report))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference only exists in the build container")
def test_reference_factory_builds_this_packages_agents():
    out = subprocess.run([sys.executable, "-c", SCRIPT, REPO, REF], capture_output=True, text=True,
                         env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    report = json.loads(out.stdout.strip().splitlines()[-1])
    assert {"MCTSAgent", "DeterministicPlannerAgent", "StateAwarePlannerAgent", "DiscreteRobustPlannerAgent"} <= set(report)
