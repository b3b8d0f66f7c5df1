import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# A hung kernel or a rank blocked in a collective must cost a test, not the whole run: ten minutes per test where the
# pytest-timeout plugin is installed (it reads this at configure time; --timeout on the command line still wins).
os.environ.setdefault("PYTEST_TIMEOUT", "600")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(REPO, "tests", "golden")
    return {k: np.load(os.path.join(d, k + ".npz")) for k in ("vi", "opd", "uct", "misc", "uct_cartpole", "uct_prior", "state_aware", "tree_tools", "per_episode", "per_episode_prior")}
