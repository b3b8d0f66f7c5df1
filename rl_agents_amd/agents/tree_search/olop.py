"""Budget allocation used by MCTS when no horizon is configured (reference ``tree_search/olop.py:42-62``).
Only these two functions of the reference's OLOP class are on the planning path."""
from rl_agents_amd import native


class OLOP(object):
    @staticmethod
    def allocation(budget, gamma):
        """Largest number of episodes e with e * horizon(e) <= budget, and that horizon."""
        return native.olop_allocation(budget, gamma)
