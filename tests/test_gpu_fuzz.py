"""A fixed slice of the parity fuzz campaign (tools/fuzz_parity.py) as a regression test: random robots x
RMP sets (every leaf kind, random order, parameters jittered per leaf) x obstacle interfaces (shared / ragged spheres and
capsules, explicit pairs, attached-point records, fused link geometry) x mappings x resolves x fleet sizes, poisoned robots
and fused rollouts -- the HIP engine against the oracle, every robot bounded by oracle.accuracy_gate.  (What the campaign
exposed in round 4 has tests of its own: test_gpu_parity.py::test_structural_zero_columns_*, ::test_all_empty_ragged_lists,
test_oracle_pins.py::test_pinv_of_a_non_finite_system_is_nan; a seed draws the same case only as long as the generator is
unchanged, so the slices below are coverage, not replays.)"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK_HEAVY = list(range(300000, 300060))      # (the range the link-geometry interface was first run on)


@pytest.mark.parametrize("seeds", [LINK_HEAVY, list(range(0, 150)), list(range(200000, 200090))],
                         ids=["300000..300059", "0..149", "200000..200089"])
def test_fuzz_slice(hip_lib, seeds):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import fuzz_parity as F
    finally:
        sys.path.pop(0)
    outcomes = {"passed": 0, "declined": 0, "failed": 0}
    failures = []
    for seed in seeds:
        outcome, what = F.run_case(seed, torch)
        outcomes[outcome] += 1
        if outcome == "failed":
            failures.append((seed, what.get("why", "")[:300]))
    assert not failures, failures
    assert outcomes["passed"] >= 0.9 * len(seeds), outcomes


def test_fuzz_slice_pair_grid(hip_lib):
    """The one-grid step of two engines (rmp2_step_pair: a TwoJoint and a Panda shard of one mixed rank, BASELINE config 5) on
    random RMP sets, shared / ragged sphere tables, fleets just over the fused grid's threshold -- each part against its oracle."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import fuzz_parity as F
    finally:
        sys.path.pop(0)
    ran, failures = [], []
    for seed in range(12):
        outcome, what = F.run_pair_case(seed, torch)
        ran.append(what.get("ran", ""))
        if outcome == "failed":
            failures.append((seed, what.get("why", "")[:300]))
    assert not failures, failures
    assert any("pair" in r for r in ran), ran          # the fused grid did run in some of them
