"""Default dispatch at awkward fleet sizes: block tails (R not a multiple of the robots per wave / block), the cuts between
the three mappings (dispatch_solve, csrc/rmp2_hip.hip: hex up to 8 192 robots, quad beyond, lane beyond 49 152 for sets
without distance leaves under solve = auto) and between the quad kernel's register caps (two / three waves per SIMD by fleet size).  Against
the oracle on the first and last robots of the fleet and a random sample in between; the kernel that ran is asserted, so a
moved cut shows up here and not only in a benchmark."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = {
    # R: (kernel of config 2, kernel of config 3)
    1: ("hex", "hex"), 17: ("hex", "hex"), 4097: ("hex", "hex"), 8192: ("hex", "hex"), 8193: ("quad", "quad"),
    20481: ("quad", "quad"), 32769: ("quad", "quad"), 49152: ("quad", "quad"), 49153: ("one lane", "quad"),
    70001: ("one lane", "quad"),
}


@pytest.mark.parametrize("name", ["config2", "config3"])
def test_awkward_fleet_sizes_default_dispatch(hip_lib, name):
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    _, desc = getattr(Cf, name)()
    eng = Engine(desc, 0)
    for R, kernels in SIZES.items():
        rng = np.random.default_rng(R)
        s = Cf.sample_panda_states(rng, R)
        sph = Cf.sample_spheres(rng)
        sph[:, 2] += 1.2                       # above the arms: clearances stay out of the near-contact regime
        obs = eng.obstacles(spheres=torch.from_numpy(sph)) if name == "config3" else None
        st = torch.zeros(R, dtype=torch.int32, device="cuda")
        out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), obstacles=obs, status=st)
        torch.cuda.synchronize()
        assert kernels[name == "config3"] in eng.last_kernel(), f"{name} R={R}: {eng.last_kernel()}"
        sub = np.unique(np.concatenate([np.arange(min(R, 40)), np.arange(max(0, R - 40), R), rng.integers(0, R, 200)]))
        kw = dict(spheres=sph) if name == "config3" else {}
        ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw)["qdd64"]
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), f"{name} R={R}: non-finite output"
        err = np.abs(got[sub] - ref).max(axis=1)
        tol = 1e-5 * np.maximum(1.0, np.abs(ref).max(axis=1))     # the tolerance of tests/test_gpu_parity.py
        assert (err <= tol).all(), f"{name} R={R}: worst {err.max():.3e} ({eng.last_kernel()})"
        assert int((st != 0).sum()) == 0


def test_a_pinv_handle_never_takes_the_plain_elimination_of_the_lane_kernel(hip_lib):
    """solve = pinv without distance leaves beyond the lane cut: the quad mapping's certifying step, not the lane-per-robot kernel's
    AUTO resolve (which fleets beyond 32 768 robots silently took until round 5: the same numbers on full-rank robots, but not the
    resolve the handle asks for)."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    _, desc = Cf.config2("pinv")
    eng = Engine(desc, 0)
    R = 70001
    s = Cf.sample_panda_states(np.random.default_rng(R), R)
    out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]))
    torch.cuda.synchronize()
    assert "quad" in eng.last_kernel() and "certified" in eng.last_kernel(), eng.last_kernel()
    sub = np.concatenate([np.arange(64), np.arange(R - 64, R)])
    ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub])["qdd64"]
    err = np.abs(out.cpu().numpy()[sub] - ref).max(axis=1)
    assert (err <= 1e-5 * np.maximum(1.0, np.abs(ref).max(axis=1))).all()


@pytest.mark.parametrize("rank", [0, 7])
def test_eight_rank_shard_shapes_of_the_mixed_fleet(rank):
    """BASELINE config 5 at its full size: the shards rank 0 and rank 7 of the 262 144-robot fleet would own on 8 GPUs
    (fleet.MixedFleetShard.synthetic: tens of thousands of ragged TwoJoint robots on one GPU / ~22 k ragged Pandas),
    built and stepped on this GPU exactly as bench.py --emulate-world 8 does, against the oracle on a sample of each robot
    type present; every robot finite (or flagged)."""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle as O
    from riemannian_motion_policies_amd.fleet import MixedFleetShard
    shard = MixedFleetShard.synthetic(262144, 8, rank, 0)
    assert shard.n_two_joint + shard.n_panda > 15000
    for _ in range(2):
        shard.step()
    torch.cuda.synchronize()
    n_ref = 160
    for key, part in shard.parts.items():
        q, qd, goal, _ = part["keep"]
        h = part["host"]
        n = min(n_ref, part["n"])
        off = h["csr_offset"][: n + 1]
        ref = O.step(part["desc"], q[:n].cpu().numpy(), qd[:n].cpu().numpy(), goal[:n].cpu().numpy(),
                     spheres=h["spheres"], csr_offset=off, csr_index=h["csr_index"][: off[-1]])
        got = part["out"][:n].cpu().numpy()
        # perf inputs are unrestricted (near-contact robots, and the TwoJoint set has no inertia leaf): EVERY robot of the sample
        # must pass a bound -- the north-star 1e-5, else the backward error against the oracle's system (no loosening with
        # cond(M)), else the robot's own fp32 resolution (oracle.accuracy_gate); none is exempted
        kw = dict(spheres=h["spheres"], csr_offset=off, csr_index=h["csr_index"][: off[-1]])
        res = O.fp32_resolution(part["desc"], q[:n].cpu().numpy(), qd[:n].cpu().numpy(), goal[:n].cpu().numpy(), **kw)
        verdict = O.accuracy_gate(got, ref, spread=res)
        assert verdict["ok"].all(), f"rank {rank} {key}: {O.gate_summary(verdict)}"
        print(f"rank {rank} {key}: {O.gate_summary(verdict)}")
        allout = part["out"].cpu().numpy()
        assert np.isfinite(allout).mean() > 0.999, f"rank {rank} {key}: {np.isnan(allout).any(axis=1).sum()} non-finite robots"


@pytest.mark.parametrize("solve", ["auto", "pinv"])
def test_two_robot_types_in_one_grid(solve):
    """(solve = "pinv", round 4: the 2-dof part's closed form IS the pseudo-inverse, the Panda part certifies full rank per
    robot -- a strict mixed shard steps as one grid too.)
    rmp2_step_pair: the TwoJoint and the Panda part of a mixed shard as ONE grid (the first blocks run the TwoJoint
    program, the rest the Panda's).  Same template bodies as the two separate launches: the results must be identical bit
    for bit, and right against the oracle; shards whose parts are small keep two launches."""
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle as O
    from riemannian_motion_policies_amd.fleet import MixedFleetShard
    fused = MixedFleetShard.synthetic(40000, 1, 0, 0, solve=solve)               # 20 000 + 20 000 robots
    plain = MixedFleetShard.synthetic(40000, 1, 0, 0, fused=False, solve=solve)
    assert fused._fused and "pair" in fused.parts["panda"]["engine"].last_kernel()
    assert not plain._fused
    for _ in range(2):
        fused.step()
        plain.step()
    torch.cuda.synchronize()
    for key in ("two_joint", "panda"):
        a, b = fused.parts[key]["out"], plain.parts[key]["out"]
        assert torch.equal(a, b) or torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)), f"{key}: fused grid differs from the two launches"
        part = fused.parts[key]
        q, qd, goal, _ = part["keep"]
        h = part["host"]
        n = 128
        off = h["csr_offset"][: n + 1]
        ref = O.step(part["desc"], q[:n].cpu().numpy(), qd[:n].cpu().numpy(), goal[:n].cpu().numpy(),
                     spheres=h["spheres"], csr_offset=off, csr_index=h["csr_index"][: off[-1]])
        kw = dict(spheres=h["spheres"], csr_offset=off, csr_index=h["csr_index"][: off[-1]])
        res = O.fp32_resolution(part["desc"], q[:n].cpu().numpy(), qd[:n].cpu().numpy(), goal[:n].cpu().numpy(), **kw)
        verdict = O.accuracy_gate(a[:n].cpu().numpy(), ref, spread=res)     # every robot bounded, none exempted
        assert verdict["ok"].all(), f"{key}: {O.gate_summary(verdict)}"
    small = MixedFleetShard.synthetic(4096, 1, 0, 0)
    assert not small._fused                                          # 2 048 + 2 048 robots: two launches (hex mapping)
