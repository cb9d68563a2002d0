"""The accuracy gate on the BASELINE perf inputs and on committed near-contact fixtures (round-4 review, item 1): error the kernel
ADDS, separated from error ANY fp32 evaluation of the reference's formulae has.

north_star asks for 1e-5 against the reference's fp32 TensorFlow path.  On the unrestricted perf inputs (seed 1) 3 % of
the config-3 robots and a fifth of the capsule fleet sit within millimetres of an obstacle -- or inside one --, where the reference's
own formulae amplify an fp32 rounding of a position into 1e-4 .. 1e-2 of q-double-dot: the autograd restatement of the reference's
graph (tests/golden/perf_envelope.npz, `err_ref32`) and the C oracle's fp32-leaf build miss the fp64 evaluation of the same
formulae by that much themselves, and so would TensorFlow.  The test therefore holds EVERY robot to one of
    A   |engine - fp64|_inf <= 1e-5 max(1, |fp64|_inf)                            (north star, against the exact value)
    E   |engine - fp64|_inf <= 2 x max(fp32 envelope of the robot, err_ref32)      (oracle.fp32_envelope: 17 fp32 evaluations)
        -- all but 3 per thousand of a fleet (chance: the engine's error is one more draw of the robot's fp32 noise), and 8 x for all --
and the fleet as a whole to error quantiles near the fp32-leaf oracle's (90th percentile within 1.25 x, 99th within 2 x).  B (backward error <= oracle.ETA = 2e-5)
is counted and asserted for all but a handful.  tools/accuracy_survey.py prints the same quantities as a table
(profiles/r05_accuracy_survey.txt), with the control that justifies the factor 2."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.fixture(scope="module")
def torch_mod(hip_lib):
    import torch
    assert torch.cuda.is_available()
    return torch


def _judge(name, got, desc, fleet, kw, err_ref32, max_beyond_b):
    """The assertions shared by the perf fleets and the near-contact fixtures; returns the counts (for the message)."""
    import oracle as O
    q, qd, goal = fleet["q"], fleet["qd"], fleet["goal"]
    exact = O.step(desc, q, qd, goal, precision="f64", **kw)
    truth = exact["qdd64"]
    c32 = O.step(desc, q, qd, goal, precision="f32", **kw)
    env = O.fp32_envelope(desc, q, qd, goal, **kw)
    if err_ref32 is not None:
        env = np.maximum(env, err_ref32)
    # every clause against the EXACT evaluation of the reference's formulae (its q-double-dot for A and E, its system (M, f) for the
    # backward error B): an fp32 evaluation's own system is no yardstick near contact -- `1. - tf.sigmoid(z)` (rmp2.py:189-194) alone
    # leaves 1e-7 / (1 - sigmoid) of relative noise in a pair's weight, percents for a point that moves away fast, which the kernels'
    # cancellation-free form does not share
    v = O.accuracy_gate(got, exact, truth=truth, envelope=env)
    scale = np.maximum(1.0, np.abs(truth).max(axis=1))
    err = np.abs(got.astype(np.float64) - truth).max(axis=1)
    a64 = err <= 1e-5 * scale
    e = v["each"]["e"]
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(env > 0, err / env, np.inf)
    beyond2 = ~(a64 | e)                       # outside the north star AND beyond twice the envelope
    beyond8 = ~a64 & ~(ratio <= 8.0)
    counts = dict(robots=len(got), A_vs_fp64=int(a64.sum()), B=int(v["each"]["b"].sum()), E_x2=int(e.sum()), beyond_x2=int(beyond2.sum()),
                  beyond_x8=int(beyond8.sum()), worst_ratio=float(np.max(ratio[~a64])) if (~a64).any() else 0.0)
    # The bulk bound: twice the envelope.  A robot's envelope is the largest of seventeen draws of its fp32 noise and the engine's
    # error one more draw, so a robot lands beyond the factor 2 by chance now and then -- and the kernel is not a seventeen-sample
    # maximum's equal either (measured, profiles/r05_accuracy_survey.txt: its ratio to the envelope has a 90th percentile of
    # 1.35-1.9 over the robots outside A where a further oracle evaluation's is 1.0: the frame-summed metric J^T (sum m n n^T) J, the
    # 1-ulp reciprocals / exponentials).  Allowed: 3 per thousand beyond 2 x, none beyond 8 x.
    bad = np.nonzero(beyond8)[0]
    assert bad.size == 0, (f"{name}: {bad.size} robot(s) further from the fp64 evaluation than 1e-5 AND than 8 x their fp32 envelope, first "
                           f"{bad[:5].tolist()}: err {err[bad[:5]].tolist()}, envelope {env[bad[:5]].tolist()}, |qdd| {scale[bad[:5]].tolist()}; {counts}")
    assert beyond2.sum() <= max(1, int(np.ceil(3e-3 * len(got)))), (
        f"{name}: {int(beyond2.sum())} robots beyond twice their fp32 envelope (ratios {np.sort(ratio[beyond2])[::-1][:6].round(2).tolist()}); {counts}")
    # backward error: oracle.ETA (2e-5) holds all but a handful (capsule fleet: control points inside a capsule)
    assert (~v["each"]["b"]).sum() <= max_beyond_b, f"{name}: {int((~v['each']['b']).sum())} robots beyond omega <= {O.ETA}; worst {np.nanmax(v['omega']):.2e}"
    # the fleet as a whole: no worse than an fp32 evaluation with correctly rounded square roots and divisions
    e_c32 = np.abs(c32["qdd64"] - truth).max(axis=1)
    # (measured: the 90th percentiles agree to 5 %, the engine's 99th is 1.5 x the oracle's on the sphere fleets)
    for pq, factor in ((90, 1.25), (99, 2.0)) if len(got) >= 1000 else ():   # (quantiles of 64 robots are single robots: the per-robot bounds above)
        pe, pc = np.percentile(err / scale, pq), np.percentile(e_c32 / scale, pq)
        assert pe <= factor * pc + 2e-7, f"{name}: {pq}th percentile of the relative error {pe:.2e} against the fp32-leaf oracle's {pc:.2e}"
    return counts


def test_perf_fleets_inside_the_fp32_envelope(torch_mod):
    """First 2 048 robots of every perf fleet, stepped at FULL fleet size (so that the kernels checked are the ones bench.py times),
    solve = pinv (the reference's resolve)."""
    torch = torch_mod
    import make_perf_envelope as E
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import MixedFleetShard
    envf = np.load(os.path.join(ROOT, "tests", "golden", "perf_envelope.npz"))
    fleets = E.perf_fleets()
    n = E.N_PERF
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for name, fl in fleets.items():   # the stored vectors belong to the inputs the seeds regenerate
        chk = np.array([fl["q"].astype(np.float64).sum(), fl["qd"].astype(np.float64).sum(), fl["goal"].astype(np.float64).sum()])
        assert np.allclose(chk, envf[name + "_input_sum"], rtol=0, atol=1e-9), name
    report = {}
    # config 2
    _, desc = Cf.config2("pinv")
    s = Cf.sample_panda_states(np.random.default_rng(1), 4096)
    eng = Engine(desc, 0)
    out = eng.step(t(s["q"]), t(s["qd"]), t(s["goal"]))
    report["config2"] = _judge("config2", out[:n].cpu().numpy(), desc, fleets["config2"], {}, envf["config2_err_ref32"], 0)
    # config 3 / 3c on the shared table, 3b through explicit pairs
    s = Cf.sample_panda_states(np.random.default_rng(1), 65536)
    _, desc = Cf.config3("pinv")
    eng = Engine(desc, 0)
    q, qd, goal = t(s["q"]), t(s["qd"]), t(s["goal"])
    for name, beyond_b in (("config3", 0), ("config3c", 6)):
        fl = fleets[name]
        out = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=t(fl["table"])))
        assert "quad" in eng.last_kernel()
        report[name] = _judge(name, out[:n].cpu().numpy(), desc, fl, dict(spheres=fl["table"]), envf[name + "_err_ref32"], beyond_b)
    fl = fleets["config3"]
    pl, po = E.pair_arrays(fl)
    reps = 65536 // n
    out = eng.step(t(np.tile(fl["q"], (reps, 1))), t(np.tile(fl["qd"], (reps, 1))), t(np.tile(fl["goal"], (reps, 1))),
                   obstacles=eng.obstacles(p_link=t(np.tile(pl, (reps, 1, 1))), p_obs=t(np.tile(po, (reps, 1, 1)))))
    report["config3b"] = _judge("config3b", out[:n].cpu().numpy(), desc, fl, dict(p_link=pl, p_obs=po), envf["config3_err_ref32"], 0)
    assert np.array_equal(out[:n].cpu().numpy(), out[n:2 * n].cpu().numpy(), equal_nan=True)      # (the tiles are copies of one another)
    del eng
    # config 5: rank 0 (131 072 TwoJoint robots) and rank 7 (~18.7 k Pandas) of the 8-rank cut, ragged lists
    for rank, key, name in ((0, "two_joint", "config5_two_joint"), (7, "panda", "config5_panda")):
        shard = MixedFleetShard.synthetic(262144, 8, rank, 0, solve="pinv", cost=E._fixture_cut())
        shard.step()
        torch.cuda.synchronize()
        part, fl = shard.parts[key], fleets[name]
        m = len(fl["q"])
        assert np.array_equal(part["keep"][0][:m].cpu().numpy(), fl["q"])
        report[name] = _judge(name, part["out"][:m].cpu().numpy(), part["desc"], fl, E.obstacle_kwargs(fl), envf[name + "_err_ref32"], 0)
        del shard
    # the statement is not vacuous: robots beyond the north star exist in every obstacle fleet and are held by E
    assert all(report[k]["A_vs_fp64"] < report[k]["robots"] for k in ("config3", "config3c", "config5_two_joint", "config5_panda")), report


@pytest.mark.parametrize("kernel", ["", "hex", "quad", "lane"])
def test_near_contact_fixtures(torch_mod, kernel):
    """tests/golden/near_contact.npz: 64 robots per fleet with their smallest clearance in [0.005, 0.05] m (the band the other
    fixtures reject), committed with the expected values of three evaluations -- autograd fp32 restatement, C oracle fp32 leaves, C
    oracle fp64.  Every mapping; every robot within 1e-5 of the fp64 value or within twice its fp32 envelope, the stored
    evaluations' own errors included in that envelope."""
    torch = torch_mod
    import oracle as O
    import make_perf_envelope as E
    from riemannian_motion_policies_amd.engine import Engine
    g = np.load(os.path.join(ROOT, "tests", "golden", "near_contact.npz"))
    base = E.perf_fleets(4)
    for name in ("config3", "config3c", "config5_two_joint", "config5_panda"):
        fl = dict(base[name], q=g[f"{name}_q"], qd=g[f"{name}_qd"], goal=g[f"{name}_goal"], table=g[f"{name}_table"])
        if f"{name}_csr_offset" in g.files:
            fl.update(csr_offset=g[f"{name}_csr_offset"], csr_index=g[f"{name}_csr_index"])
        _, desc = E.desc_of(fl)
        kw = E.obstacle_kwargs(fl)
        # the stored expectations are what this checkout's oracle computes (the fixture has not gone stale)
        r64 = O.step(desc, fl["q"], fl["qd"], fl["goal"], precision="f64", **kw)["qdd64"]
        assert np.abs(r64 - g[f"{name}_qdd_f64"]).max() <= 1e-9 * max(1.0, np.abs(r64).max())
        assert (g[f"{name}_clearance"] >= 0.005).all() and (g[f"{name}_clearance"] <= 0.05).all()
        old = os.environ.get("RMP2_KERNEL")
        if kernel:
            os.environ["RMP2_KERNEL"] = kernel
        try:
            eng = Engine(desc, 0)
        finally:
            if old is None:
                os.environ.pop("RMP2_KERNEL", None)
            else:
                os.environ["RMP2_KERNEL"] = old
        dev_kw = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in kw.items()}
        out = eng.step(torch.from_numpy(fl["q"]), torch.from_numpy(fl["qd"]), torch.from_numpy(fl["goal"]), obstacles=eng.obstacles(**dev_kw))
        torch.cuda.synchronize()
        if kernel:
            assert kernel in eng.last_kernel(), eng.last_kernel()
        stored = np.maximum(np.abs(g[f"{name}_qdd_ref32"] - g[f"{name}_qdd_pairs_f64"]).max(axis=1),
                            np.abs(g[f"{name}_qdd_c32"] - g[f"{name}_qdd_f64"]).max(axis=1))
        counts = _judge(f"near-contact {name} [{kernel or 'default'}]", out.cpu().numpy(), desc, fl, kw, stored, 2)
        # (clearances of 5 .. 50 mm: |qdd| reaches 1e2 .. 1e3, yet the engine stays within 1e-5 RELATIVE of the fp64 value on nearly
        #  every robot of these fixtures -- what breaks the north star on the perf fleets is contact and penetration, below 5 mm)
        assert counts["A_vs_fp64"] >= 0.9 * counts["robots"], counts
