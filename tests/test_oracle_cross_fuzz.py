"""The two oracles against each other on RANDOM sets (CPU): the plain-C analytic restatement (oracle/rmp2_oracle.c: geometric
Jacobians, closed-form curvature terms, its own pseudo-inverse) and the autodiff restatement of the reference's graph
(oracle/torch_autodiff_oracle.py: the reference's tensor ops, derivatives by nested autograd, numpy's pinv with TensorFlow's
cutoff) share no derivation -- the fixtures pin one to the other on the reference's own experiment sets; this does it on sets
nobody wrote down: both reference robots, a random subset of the leaf kinds in random order with parameters jittered per leaf
on random frames (tools/fuzz_parity.draw_specs), sphere / capsule tables (shared and ragged) against the equivalent explicit
pairs, explicit closest-point pairs / attached-point records with uneven counts,
both resolves.  An inertia leaf is forced so that the systems are the kind fp32 determines.  (120 seeds in
the suite; `pytest tests/test_oracle_cross_fuzz.py -n 6` with the range raised to 1 500 ran clean in round 4: 4 500 robots.)"""
import json
import os
import sys

import numpy as np
import pytest

import oracle as O
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd import descriptor as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", range(120))
def test_c_oracle_against_autodiff_oracle_on_random_sets(golden_dir, seed):
    import torch_autodiff_oracle as TA
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import fuzz_parity as F
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(7000 + seed)
    gold = json.load(open(os.path.join(golden_dir, "kinematic_tables.json")))
    name = "panda" if seed % 2 else "two_joint"
    t = Cf.panda_table() if name == "panda" else Cf.two_joint_table()
    lo, hi = (Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH) if name == "panda" else (Cf.TWO_JOINT_Q_LOW, Cf.TWO_JOINT_Q_HIGH)
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    specs, obstacle_kind = F.draw_specs(rng, t, lo, hi)
    if not any(s.kind == D.LEAF_JOINT_DAMPING for s in specs):
        specs.append(D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, F.jitter(rng, Cf.JOINT_DAMPING_PARAMS)))
    solve = "pinv" if seed % 3 == 0 else "auto"
    desc = D.build_desc(t, specs, solve)
    n, R = t.n_dof, 3
    span = hi - lo
    q = rng.uniform(lo + 0.1 * span, hi - 0.1 * span, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.8, 0.8, (R, max(desc.goal_floats, 1))).astype(np.float32)
    dl = D.distance_leaf_indices(desc)
    kw, pairs_of = {}, [dict() for _ in range(R)]
    if dl:
        counts = [int(rng.integers(1, 4)) for _ in dl]
        begin = np.concatenate([[0], np.cumsum(counts)])
        P = int(begin[-1])
        if obstacle_kind == "point":
            rel = rng.uniform(-0.15, 0.15, (R, P, 3)).astype(np.float32)
            nv = rng.normal(size=(R, P, 3))
            nv = (nv / np.linalg.norm(nv, axis=-1, keepdims=True)).astype(np.float32)
            dist = rng.uniform(0.05, 1.3, (R, P)).astype(np.float32)
            kw = dict(p_link=rel, p_obs=nv, dist=dist, pair_counts=counts)
            for r in range(R):
                for k, li in enumerate(dl):
                    sl = slice(begin[k], begin[k + 1])
                    pairs_of[r][li] = (rel[r, sl], nv[r, sl], dist[r, sl])
        elif rng.random() < 0.5:
            # a primitive TABLE for the C oracle (shared, or a ragged list per robot: its own closest-point code for spheres and
            # capsules) against the autodiff restatement fed the equivalent explicit pairs, formed here in numpy
            K = int(rng.integers(1, 6))
            T = O.forward_kinematics(desc, q, "f64")
            org = T[:, [desc.leaves[li].frame for li in dl]][:, :, :3, 3]
            ctr = org.reshape(-1, 3)[rng.integers(0, R * len(dl), K)] + rng.normal(size=(K, 3)) * 0.35     # near the control points
            rad = rng.uniform(0.03, 0.08, (K, 1))
            capsules = rng.random() < 0.5
            tab = (np.concatenate([ctr, rad, ctr + rng.normal(size=(K, 3)) * 0.2, np.zeros((K, 1))], axis=1) if capsules
                   else np.concatenate([ctr, rad], axis=1)).astype(np.float32)
            pl, po = (Cf.pairs_from_capsules if capsules else Cf.pairs_from_spheres)(org.astype(np.float32), tab)
            # keep the fixture rule (clear of contact): drop tables that come within 0.06 m of a control point
            if (np.linalg.norm(pl.astype(np.float64) - po, axis=-1) < 0.06).any():
                tab[:, :3] += 5.0
                if capsules:
                    tab[:, 4:7] += 5.0
                pl, po = (Cf.pairs_from_capsules if capsules else Cf.pairs_from_spheres)(org.astype(np.float32), tab)
            kw = dict(spheres=tab)
            lists = [np.arange(K)] * R
            if rng.random() < 0.5:
                lists = [rng.permutation(K)[: int(rng.integers(0, K + 1))] for _ in range(R)]
                kw.update(csr_offset=np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32),
                          csr_index=np.concatenate(lists + [np.zeros(0, np.int64)]).astype(np.int32))
            for r in range(R):
                for k, li in enumerate(dl):
                    cols = k * K + lists[r]
                    pairs_of[r][li] = (pl[r, cols], po[r, cols])
        else:
            T = O.forward_kinematics(desc, q, "f64")
            pl = np.zeros((R, P, 3), np.float32)
            po = np.zeros((R, P, 3), np.float32)
            for k, li in enumerate(dl):
                org = T[:, desc.leaves[li].frame, :3, 3]
                c = counts[k]
                a = org[:, None, :] + rng.uniform(-0.08, 0.08, (R, c, 3))
                dirs = rng.normal(size=(R, c, 3))
                dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
                pl[:, begin[k]:begin[k + 1]] = a
                po[:, begin[k]:begin[k + 1]] = a + dirs * rng.uniform(0.06, 0.6, (R, c, 1))    # clear of contact (fixture rule)
            kw = dict(p_link=pl, p_obs=po, pair_counts=counts)
            for r in range(R):
                for k, li in enumerate(dl):
                    sl = slice(begin[k], begin[k + 1])
                    pairs_of[r][li] = (pl[r, sl], po[r, sl])
    ref = O.step(desc, q, qd, goal[:, : desc.goal_floats] if desc.goal_floats else None, **kw)
    fk = TA.UrdfForwardKinematicTorch(gold[name])
    leaves = TA.leaves_from_desc(desc, t.frame_names)
    for r in range(R):
        qdd, M, f = TA.evaluate_one(fk, leaves, q[r], qd[r], goal[r], pairs=pairs_of[r])
        scale_M = np.abs(M).max()
        assert np.abs(ref["M"][r] - M).max() <= 2e-5 * scale_M, f"seed {seed} robot {r}: M differs by {np.abs(ref['M'][r] - M).max() / scale_M:.2e} ({[(s.kind, s.taskmap, s.frame) for s in specs]})"
        assert np.abs(ref["f"][r] - f).max() <= 2e-5 * max(np.abs(f).max(), 1e-6 * scale_M), f"seed {seed} robot {r}: f differs"
        # both resolve their own system the same way (TensorFlow's cutoff); the systems are full rank by the inertia leaf
        cond = np.linalg.cond(M)
        # (two fp32 evaluations of the leaves in different operation orders: ~2e-7 relative in (M, f), times the condition number)
        tol = max(1e-5 * max(1.0, np.abs(qdd).max()), 5e-7 * cond * np.abs(qdd).max())
        assert np.abs(ref["qdd64"][r] - qdd).max() <= tol, f"seed {seed} robot {r}: qdd differs by {np.abs(ref['qdd64'][r] - qdd).max():.2e} (cond {cond:.1e})"
