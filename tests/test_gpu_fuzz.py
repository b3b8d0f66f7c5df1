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
    assert sum(kinds.values()) == 200 and len(kinds) == 15        # every kind, the three of round 3 and of round 5 included


def test_random_cases_of_the_round3_kinds(monkeypatch):
    """Only the kinds added in round 3: robust OPD and state-aware OPD on restricted action sets (all kernel mappings),
    MCTS on stochastic / sparse / deterministic models through the literal kernel (open and closed loop, whole trees)."""
    import fuzz_parity
    monkeypatch.setenv("FUZZ_KINDS", "ropd_masked,saopd_masked,uct_stoch")
    kinds = fuzz_parity.run(150, 31)
    assert sum(kinds.values()) == 150 and set(kinds) == {"ropd_masked", "saopd_masked", "uct_stoch"}


def test_random_cases_of_the_round5_kinds(monkeypatch):
    """Only the kinds added in round 5: N MDPs per launch (batched VI in every kernel form, tables replaced between rounds),
    UCT / OPD with one MDP per root, delta uploads of single models (mp_model_update_rows)."""
    import fuzz_parity
    monkeypatch.setenv("FUZZ_KINDS", "vi_batch,per_root_models,update_rows")
    kinds = fuzz_parity.run(150, 51)
    assert sum(kinds.values()) == 150 and set(kinds) == {"vi_batch", "per_root_models", "update_rows"}
