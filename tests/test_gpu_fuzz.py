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
    weaker = {"undetermined_at_fp32_backward_error_only": 0, "tiny_projection_alone_on_a_dof_componentwise_limit": 0,
              "poisoned_input_answered_nan_where_the_oracle_stays_finite": 0, "oracle_nan_answered_finite": 0}
    robots = 0
    for seed in seeds:
        outcome, what = F.run_case(seed, torch)
        outcomes[outcome] += 1
        if outcome == "failed":
            failures.append((seed, what.get("why", "")[:300]))
        for k in weaker:
            weaker[k] += int(what.get("gate", {}).get(k, 0))
        robots += int(what.get("gate", {}).get("robots", 0))
    assert not failures, failures
    assert outcomes["passed"] >= 0.9 * len(seeds), outcomes
    # the classes that get a weaker bound than the gate's stay what they were when the slices were recorded (round 4's campaign:
    # ~4e-5 of the robots undetermined at fp32, ~3e-7 in the componentwise-limit class, ~1e-5 poisoned-and-NaN): a kernel
    # regression that hides in one of them moves these counts by orders of magnitude
    assert weaker["oracle_nan_answered_finite"] == 0, weaker
    assert weaker["undetermined_at_fp32_backward_error_only"] <= max(20, 2e-3 * robots), (weaker, robots)
    assert weaker["tiny_projection_alone_on_a_dof_componentwise_limit"] <= 2, (weaker, robots)
    assert weaker["poisoned_input_answered_nan_where_the_oracle_stays_finite"] <= 3 * len(seeds), (weaker, robots)


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
