"""Agent construction from JSON configs (reference ``rl_agents/agents/common/factory.py:12-56,97-116``).

A reference agent config becomes an MI355X one by changing only its ``"__class__"`` string, e.g.
``"<class 'rl_agents.agents.tree_search.mcts.MCTSAgent'>"`` ->
``"<class 'rl_agents_amd.agents.tree_search.mcts.MCTSAgent'>"``.
"""
import importlib
import json
import logging

from rl_agents_amd.configuration import merge_config

logger = logging.getLogger(__name__)


def agent_factory(environment, config):
    """Instantiate ``config["__class__"]`` (format ``"<class 'package.module.Class'>"``) as ``Class(environment, config)``."""
    if "__class__" not in config:
        raise ValueError("The configuration should specify the agent __class__")
    dotted = config["__class__"].split("'")[1]
    module_name, class_name = dotted.rsplit(".", 1)
    agent_class = getattr(importlib.import_module(module_name), class_name)
    return agent_class(environment, config)


def load_agent_config(config_path):
    """Read an agent JSON file, resolving ``"base_config"`` inheritance (child keys override the base)."""
    with open(config_path) as f:
        config = json.load(f)
    base_path = config.pop("base_config", None)
    if base_path is not None:
        config = merge_config(load_agent_config(base_path), config)
    return config


def load_agent(agent_config, env):
    """``agent_config``: a dict or the path of a JSON file."""
    if not isinstance(agent_config, dict):
        agent_config = load_agent_config(agent_config)
    return agent_factory(env, agent_config)


def preprocess_env(env, preprocessor_configs):
    """Apply ``[{"method": name, "args": ...}, ...]`` to ``env.unwrapped`` in turn (e.g. highway ``simplify``)."""
    for entry in preprocessor_configs:
        if "method" not in entry:
            logger.error("The method is not specified in %s", entry)
            continue
        try:
            method = getattr(env.unwrapped, entry["method"])
        except AttributeError:
            logger.warning("The environment does not have a %s method", entry["method"])
            continue
        env = method(entry["args"]) if "args" in entry else method()
    return env
