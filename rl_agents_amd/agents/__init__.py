"""Agents of the planning hot path, same class names and module layout as the reference
(``rl_agents/agents/{tree_search,dynamic_programming}``), computing on libmi355plan.so."""
