"""Configuration container of agents and planners.

Mirrors the behaviour of the reference's ``rl_agents/configuration.py:5-44`` (``Configurable``): a
class-level ``default_config()`` is overridden by the user's dict, and the user's dict is then
completed in place with the resulting values (the reference's JSON round-trip relies on that:
``serialize(agent)`` dumps ``agent.config``, configuration.py:93-99).
"""
from collections.abc import Mapping


def merge_config(target, source):
    """Recursively write ``source`` into ``target`` (mappings are merged, everything else replaced)."""
    for key, value in source.items():
        if isinstance(value, Mapping):
            target[key] = merge_config(target.get(key, {}), value)
        else:
            target[key] = value
    return target


class Configurable(object):
    def __init__(self, config=None):
        self.config = self.default_config()
        if config:
            merge_config(self.config, config)   # user values win over defaults
            merge_config(config, self.config)   # and the user's dict is back-filled with the defaults

    def update_config(self, config):
        merge_config(self.config, config)

    @classmethod
    def default_config(cls):
        return {}

    # the reference exposes the merge as a static method; keep the name for callers that use it
    rec_update = staticmethod(merge_config)
