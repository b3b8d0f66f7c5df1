"""Randomised parity sweep (tools/fuzz_parity.py): random finite MDPs x random planner parameters x random batches,
HIP planners vs the CPU oracle, bit for bit.  A bounded number of cases here; the tool runs thousands."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_cases_match_the_oracle(seed):
    import fuzz_parity
    kinds = fuzz_parity.run(200, seed)
    assert sum(kinds.values()) == 200 and len(kinds) == 9
