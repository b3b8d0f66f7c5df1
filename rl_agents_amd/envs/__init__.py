from .finite_mdp import FiniteMDPEnv, MaskedFiniteMDPEnv, OrderedMaskedFiniteMDPEnv, MDP, DeterministicMDP, StochasticMDP, SparseMDP, Discrete  # noqa: F401
from . import generators  # noqa: F401
from .cartpole import CartPoleEnv  # noqa: F401
from .highway_like import HighwayLikeEnv  # noqa: F401
from .changing import ScheduledTableEnv, MaskedScheduledTableEnv, ChangingHighwayEnv  # noqa: F401
