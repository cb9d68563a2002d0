#!/usr/bin/env python3
"""Parity fuzz campaign: random robots x random RMP sets x obstacle interfaces x kernel mappings x resolves, the HIP engine
against the CPU oracle through oracle.accuracy_gate (clauses A-D; a case with a robot outside them is judged with clause E too: within
twice the fp32 envelope of the fp64 evaluation).  Every robot gets a bound; three classes get a WEAKER one than the gate's,
each counted in the summary and capped by tests/test_gpu_fuzz.py: robots whose system is undetermined at fp32 (backward error
only), single dofs whose only metric is a tiny projection (that dof excused, the robot's other dofs gated), robots this harness
fed a non-finite state (NaN + status bit accepted where the reference's graph never reaches the value).

Test infrastructure (it calls oracle/): run on the GPU box, e.g.

    python tools/fuzz_parity.py --seeds 0 2000 --minutes 8 --log gpurun_out/r04/fuzz_parity.log

One case = one seed: a robot (TwoJoint, Panda, or a random URDF tree: revolute / prismatic / fixed joints, arbitrary axes,
branches), a random subset of the thirteen leaf kinds in random ORDER (the order fixes the fp64 summation order) with
jittered parameters on random frames, solve = auto | pinv, one obstacle interface (shared spheres / capsules, ragged lists,
explicit pairs with uneven counts per leaf, attached-point records), a fleet size with awkward tails, and one of the
mappings (default dispatch, hex, quad, lane).  A case the engine declines with RMP2_ERR_UNSUPPORTED is counted, not failed;
anything else that differs from the oracle -- a robot outside the gate, a status word that disagrees about non-finite
results, an exception -- is a failure and is logged with its seed for replay (`--seeds S S+1 --verbose`).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FLEET_SIZES = [1, 3, 16, 17, 64, 65, 257, 1000, 2049]
BIG_FLEET_SIZES = [8192, 8193, 20481, 32769]      # around the cuts between the mappings and the register-cap builds (5 % of the cases)


def jitter(rng, params):
    return [float(p) * float(rng.uniform(0.8, 1.25)) for p in params]


def draw_robot(rng, tmpdir):
    from riemannian_motion_policies_amd import configs as Cf, urdf
    kind = rng.choice(["two_joint", "panda", "random"], p=[0.2, 0.4, 0.4])
    if kind == "two_joint":
        t = Cf.two_joint_table()
        lo, hi = Cf.TWO_JOINT_Q_LOW, Cf.TWO_JOINT_Q_HIGH
    elif kind == "panda":
        t = Cf.panda_table()
        lo, hi = Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH
    else:
        from test_gpu_random_robots import _write_urdf
        path = os.path.join(tmpdir, "rnd.urdf")
        t = None
        for _ in range(50):
            movable = _write_urdf(path, rng, int(rng.integers(2, 20 if rng.random() < 0.2 else 13)), branch_prob=0.25)
            order = [m for m in movable if rng.random() < 0.9][:(16 if rng.random() < 0.15 else 9)]   # (10 .. 16 dofs: hex mapping only)
            if not order:
                continue
            t = urdf.compile_urdf(path, order)
            if t.depth_first_schedule()[3] <= 2:
                break
            t = None
        if t is None:
            return draw_robot(rng, tmpdir)
        lo, hi = -np.ones(t.n_dof) * 1.2, np.ones(t.n_dof) * 1.2
    return kind, t, np.asarray(lo, np.float64), np.asarray(hi, np.float64)


def draw_specs(rng, t, lo, hi):
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    n, F = t.n_dof, t.n_frames
    fr = lambda: int(rng.integers(0, F))
    specs = []
    if rng.random() < 0.6:
        specs.append(D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, fr(), jitter(rng, Cf.TARGET_ATTRACTOR_PARAMS), goal_len=3))
    if rng.random() < 0.35:
        specs.append(D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, fr(), jitter(rng, Cf.TARGET_POLICY_PARAMS), goal_len=3))
    if rng.random() < 0.2:
        specs.append(D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.EXP04_TARGET_POLICY_PARAMS), goal_len=n))
    if rng.random() < 0.4:
        specs.append(D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.JOINT_VELOCITY_CAP_PARAMS)))
    if rng.random() < 0.6:
        specs.append(D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.JOINT_DAMPING_PARAMS)))
    if rng.random() < 0.4:
        specs.append(D.LeafSpec(D.LEAF_CSPACE_BIASING, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.CSPACE_BIASING_PARAMS),
                                vec_a=rng.uniform(lo, hi) * 0.5))
    if rng.random() < 0.4:
        specs.append(D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.JOINT_LIMIT_PARAMS), vec_a=lo, vec_b=hi))
    if rng.random() < 0.2:
        specs.append(D.LeafSpec(D.LEAF_CONFIG_SPACE_BIASING, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.PANDA04_CONFIG_SPACE_BIASING_PARAMS),
                                vec_a=rng.uniform(lo, hi) * 0.5))
    obstacle_kind = rng.choice(["none", "distance", "point"], p=[0.25, 0.55, 0.2])
    if obstacle_kind != "none":
        count = int(rng.integers(1, min(F, 8) + 1))
        frames = list(rng.choice(F, size=count, replace=False))
        if rng.random() < 0.15 and count < 8:
            frames.append(frames[0])          # two leaves on one frame
        for f in frames:
            if obstacle_kind == "distance":
                specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, int(f), jitter(rng, Cf.OBSTACLE_AVOIDANCE_PARAMS)))
            else:
                specs.append(D.LeafSpec(D.LEAF_COLLISION_AVOIDANCE, D.TASKMAP_FK_POINT, int(f), jitter(rng, Cf.COLLISION_AVOIDANCE_PARAMS)))
    if not specs:
        specs.append(D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.JOINT_DAMPING_PARAMS)))
    order = rng.permutation(len(specs))
    return [specs[i] for i in order], obstacle_kind


def draw_obstacles(rng, O, desc, q, obstacle_kind):
    """(kwargs for oracle.step / Engine.obstacles as numpy arrays, label)."""
    from riemannian_motion_policies_amd import descriptor as D
    R = q.shape[0]
    dl = D.distance_leaf_indices(desc)
    if obstacle_kind == "none" or not dl:
        return {}, "none"
    T = O.forward_kinematics(desc, q, "f64")
    org = T[:, [desc.leaves[i].frame for i in dl]][:, :, :3, 3]          # [R, L, 3]
    if obstacle_kind == "point":
        counts = [int(rng.integers(1, 5)) for _ in dl] if rng.random() < 0.5 else [int(rng.integers(1, 5))] * len(dl)
        P = sum(counts)
        rel = rng.uniform(-0.15, 0.15, (R, P, 3))
        nv = rng.normal(size=(R, P, 3))
        nv /= np.linalg.norm(nv, axis=-1, keepdims=True)
        d = rng.uniform(0.05, 1.3, (R, P))
        return dict(p_link=rel.astype(np.float32), p_obs=nv.astype(np.float32), dist=d.astype(np.float32), pair_counts=counts), f"point records {counts}"
    mode = rng.choice(["spheres", "capsules", "ragged", "ragged_capsules", "pairs", "link"], p=[0.27, 0.13, 0.18, 0.09, 0.2, 0.13])
    lo, hi = org.reshape(-1, 3).min(axis=0) - 0.4, org.reshape(-1, 3).max(axis=0) + 0.4
    if mode == "link":
        # link geometry: a capsule per distance leaf in its frame's coordinates; the engine forms the closest points of link capsule
        # and table primitive inside the step (fp32), the oracle reads them as explicit pairs formed here in fp64 (closed form of
        # configs.pairs_from_link_capsules on the oracle's fp64 forward kinematics; taskmap.py:124-129: the Jacobian stays the
        # frame origin's)
        from riemannian_motion_policies_amd import configs as Cf
        L = len(dl)
        K = int(rng.choice([1, 5, 32, 100]))
        la = rng.uniform(-0.05, 0.05, (L, 3))
        lb = la + rng.uniform(-0.15, 0.15, (L, 3))
        lc = np.concatenate([la, rng.uniform(0.02, 0.05, (L, 1)), lb, np.zeros((L, 1))], axis=1).astype(np.float32)
        c = rng.uniform(lo, hi, (K, 3))
        rad = rng.uniform(0.03, 0.1, (K, 1))
        if rng.random() < 0.5:
            tab = np.concatenate([c, rad, c + rng.normal(size=(K, 3)) * 0.25, np.zeros((K, 1))], axis=1).astype(np.float32)
        else:
            tab = np.concatenate([c, rad], axis=1).astype(np.float32)
        frames = [desc.leaves[i].frame for i in dl]
        pl, po = Cf.pairs_from_link_capsules(T[:, frames], lc, tab)
        pl32, po32 = Cf.pairs_from_link_capsules(T[:, frames], lc, tab, dtype=np.float32)
        eng_kw = dict(spheres=tab, link_capsules=lc)
        label = f"link geometry K={K} ({'capsules' if tab.shape[1] == 8 else 'spheres'})"
        if rng.random() < 0.4:        # ragged lists over the table (no duplicates: a list entry is a pair)
            counts = rng.integers(0, K + 1, size=R)
            off = np.zeros(R + 1, np.int32)
            off[1:] = np.cumsum(counts)
            idx = np.concatenate([rng.permutation(K)[: int(k)] for k in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
            kmax = max(int(counts.max()), 1)
            pl2 = np.zeros((R, L * kmax, 3), np.float32)
            po2 = np.full((R, L * kmax, 3), 1.0e3, np.float32)
            for r in range(R):
                lst = idx[off[r]:off[r + 1]]
                for l in range(L):
                    pl2[r, l * kmax:l * kmax + len(lst)] = pl[r, l * K + lst]
                    po2[r, l * kmax:l * kmax + len(lst)] = po[r, l * K + lst]
            pl3 = np.zeros((R, L * kmax, 3), np.float32)
            po3 = np.full((R, L * kmax, 3), 1.0e3, np.float32)
            for r in range(R):
                lst = idx[off[r]:off[r + 1]]
                for l in range(L):
                    pl3[r, l * kmax:l * kmax + len(lst)] = pl32[r, l * K + lst]
                    po3[r, l * kmax:l * kmax + len(lst)] = po32[r, l * K + lst]
            pl, po, pl32, po32 = pl2, po2, pl3, po3
            eng_kw.update(csr_offset=off, csr_index=idx)
            label += " ragged"
        return dict(p_link=pl, p_obs=po, _engine=eng_kw, _pairs_fp32=(pl32, po32)), label
    if mode == "pairs":
        counts = [int(rng.integers(1, 40)) for _ in dl] if rng.random() < 0.6 else [int(rng.choice([1, 4, 32]))] * len(dl)
        pl, po = [], []
        for k, c in enumerate(counts):
            a = org[:, k, None, :] + rng.uniform(-0.08, 0.08, (R, c, 3))
            dirs = rng.normal(size=(R, c, 3))
            dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
            gap = np.where(rng.random((R, c)) < 0.3, rng.uniform(0.01, 0.06, (R, c)), rng.uniform(0.06, 0.8, (R, c)))
            pl.append(a)
            po.append(a + dirs * gap[..., None])
        return dict(p_link=np.concatenate(pl, axis=1).astype(np.float32), p_obs=np.concatenate(po, axis=1).astype(np.float32),
                    pair_counts=counts), f"explicit pairs {counts}"
    K = int(rng.choice([1, 5, 32, 100, 300]))
    c = rng.uniform(lo, hi, (K, 3))
    rad = rng.uniform(0.03, 0.1, (K, 1))
    if mode in ("capsules", "ragged_capsules"):
        b = c + rng.normal(size=(K, 3)) * 0.25
        tab = np.concatenate([c, rad, b, np.zeros((K, 1))], axis=1).astype(np.float32)
        if K > 1:
            tab[0, 4:7] = tab[0, 0:3]      # a degenerate capsule
    else:
        tab = np.concatenate([c, rad], axis=1).astype(np.float32)
    kw = dict(spheres=tab)
    if mode.startswith("ragged"):
        counts = rng.integers(0, min(K, 40) + 1, size=R)
        off = np.zeros(R + 1, np.int32)
        off[1:] = np.cumsum(counts)
        idx = np.concatenate([rng.integers(0, K, size=int(k)) for k in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
        kw.update(csr_offset=off, csr_index=idx)
    return kw, f"{mode} K={K}"


def known_classes(O, desc, specs, n, q, qd, goal, kw, ref):
    """(undetermined [R], tiny_alone [R, n]) -- the two classes no fp32 evaluation of the reference's formulae pins down (see run_case
    for the reasoning): `undetermined` = a singular value of the oracle's M (fp32 or fp64 evaluation) inside (1e-18, 1e-6] sigma_max,
    or rank deficiency beyond the all-zero rows; `tiny_alone[r, j]` = dof j of robot r has a positive diagonal entry of M below 1e-6
    of the largest obstacle-leaf metric scale the descriptor can produce AND no larger entry in its row or column (the documented
    componentwise limit of J^T S J: the dof's whole answer is that one entry's).  Only THAT DOF is then excused; the robot's other
    dofs go through the gate like everybody's."""
    from riemannian_motion_policies_amd import descriptor as D
    ref64 = O.step(desc, q, qd, goal, precision="f64", **kw)
    with np.errstate(invalid="ignore"):
        sv = np.linalg.svd(np.where(np.isfinite(ref["M"]), ref["M"], 0.0), compute_uv=False)
        sv64 = np.linalg.svd(np.where(np.isfinite(ref64["M"]), ref64["M"], 0.0), compute_uv=False)
    rel = sv / np.maximum(sv[:, :1], 1e-300)
    rel64 = sv64 / np.maximum(sv64[:, :1], 1e-300)
    und = ((rel > 1e-18) & (rel <= 1e-6)).any(axis=1) | ((rel64 > 1e-18) & (rel64 <= 1e-6)).any(axis=1)
    cut = 10.0 * n * np.finfo(np.float64).eps
    zero_rows = ((ref["M"] == 0).all(axis=2) & (ref["M"] == 0).all(axis=1)).sum(axis=1)
    und |= (rel <= cut).sum(axis=1) > zero_rows
    # a pair within an fp32 rounding of a leaf's cutoff radius: metric exactly 0 in one evaluation, +-1e-12 in another -- when it is
    # the ONLY metric of the system, the oracle's M is all zero while its fp64 evaluation is not (or the other way round)
    und |= (sv[:, 0] == 0) != (sv64[:, 0] == 0)
    # ... or when it is the only metric of ONE dof: the rank of M then flips under a one-ulp jiggle of the inputs (oracle.rank_flips;
    # seed 510845: the engine had such a pair in range -- metric 1e-12, q-double-dot -23.8 on that dof -- both oracle builds out of range)
    und |= O.rank_flips(desc, q, qd, goal, **kw)
    Mz = np.where(np.isfinite(ref["M"]), ref["M"], 0.0)
    diag = np.einsum("rii->ri", Mz)
    scale_m = max([float(sp.params[8]) / max(float(sp.params[10]), 1e-30) for sp in specs if sp.kind == D.LEAF_OBSTACLE_AVOIDANCE] + [0.0])
    # PER DOF, and only where the tiny entry is ALONE: it dominates its row and its column (no coupling larger than itself to any
    # other dof), so that the dof's answer is f_j / M_jj to within the entry's own relative error and the other dofs do not feel it
    off = np.abs(Mz) * (1.0 - np.eye(n))[None]
    alone = (off.max(axis=2) <= diag) & (off.max(axis=1) <= diag)
    tiny = (diag > 0) & (diag < 1e-6 * max(scale_m, 1e-30)) & alone & (scale_m > 0)
    return und, tiny, rel


def excused_by_tiny_dofs(O, got, ref, tiny_alone, candidates, res=None, sys_res=None):
    """Robots among `candidates` that pass the gate once the dofs of `tiny_alone` [R, n] (known_classes) are taken from the oracle:
    the excused dof is out of the comparison, every other dof of the robot is held to the gate's bounds."""
    ok = np.zeros(len(got), bool)
    cand = candidates & tiny_alone.any(axis=1) & np.isfinite(got).all(axis=1)
    if cand.any():
        sub = {k: ref[k][cand] for k in ("qdd64", "M", "f")}
        v = O.accuracy_gate(np.where(tiny_alone, ref["qdd64"], got)[cand], sub, spread=None if res is None else res[cand],
                            system_spread=None if sys_res is None else sys_res[cand])
        ok[np.nonzero(cand)[0]] = v["ok"]
    return ok


def draw_case(seed):
    """Everything one seed fixes, drawn in ONE order (tools/diag_fuzz_*.py replay through this too): dict with table, specs,
    solve, kernel, R, desc (None: the descriptor compiler declined, `why`), q, qd, goal, oracle kwargs, engine kwargs, the
    poisoned robots, the label, and the generator -- the later draws (debug outputs, rollout) continue from it."""
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D
    rng = np.random.default_rng(seed)
    with tempfile.TemporaryDirectory() as tmp:
        robot_kind, t, lo, hi = draw_robot(rng, tmp)
    specs, obstacle_kind = draw_specs(rng, t, lo, hi)
    solve = str(rng.choice(["auto", "pinv"], p=[0.6, 0.4]))
    kernel = str(rng.choice(["", "hex", "quad", "lane"], p=[0.4, 0.2, 0.25, 0.15]))
    R = int(rng.choice(BIG_FLEET_SIZES if rng.random() < 0.05 else FLEET_SIZES))
    n = t.n_dof
    c = dict(seed=seed, robot_kind=robot_kind, table=t, lo=lo, hi=hi, specs=specs, solve=solve, kernel=kernel, R=R, n=n, rng=rng, desc=None)
    try:
        desc = D.build_desc(t, specs, solve)
    except ValueError as e:
        return dict(c, why=f"build_desc: {e}")
    span = hi - lo
    q = rng.uniform(lo + 0.05 * span, hi - 0.05 * span, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    if rng.random() < 0.3:
        qd *= 5.0                                     # faster robots: the velocity cap's band
    goal = rng.uniform(-0.8, 0.8, (R, desc.goal_floats)).astype(np.float32) if desc.goal_floats else None
    kw, obs_label = draw_obstacles(rng, O, desc, q, obstacle_kind)
    side = np.random.default_rng([seed, 5])       # round-5 extras draw from a generator of their own: a seed's base case is unchanged
    if "_engine" not in kw and "spheres" in kw and kw["spheres"].shape[1] == 8 and side.random() < 0.5:
        # the capsule table read as the reference's flat-capped CYLINDERS (simulation.py:245-261): centre = the axis' midpoint, unit axis,
        # half height = half the axis length (RMP2_PRIM_CYLINDER; shared and ragged tables, every mapping)
        tab = kw["spheres"].astype(np.float64)
        axis = tab[:, 4:7] - tab[:, 0:3]
        ln = np.linalg.norm(axis, axis=1)
        axis = np.where(ln[:, None] > 0, axis / np.where(ln > 0, ln, 1.0)[:, None], np.array([0.0, 0.0, 1.0]))
        cyl = np.concatenate([0.5 * (tab[:, 0:3] + tab[:, 4:7]), tab[:, 3:4], axis, np.maximum(0.5 * ln, 0.01)[:, None]], axis=1).astype(np.float32)
        kw = dict(kw, spheres=cyl, primitive="cylinder")
        obs_label = obs_label.replace("capsules", "cylinders")
    pairs_fp32 = kw.pop("_pairs_fp32", None)
    eng_kw = kw.pop("_engine", None) or kw       # (link geometry: the engine gets table + link capsules, the oracle the pairs)
    dead = np.zeros(R, bool)
    dead_velocity = np.zeros(R, bool)
    if R >= 16 and rng.random() < 0.15:               # a few robots fed a non-finite state: NaN out + status bit, neighbours untouched
        dead[rng.choice(R, size=3, replace=False)] = True
        bad_rows = np.nonzero(dead)[0]
        q[bad_rows[0], rng.integers(0, n)] = np.nan
        qd[bad_rows[1], rng.integers(0, n)] = np.inf
        dead_velocity[bad_rows[1]] = True
        q[bad_rows[2], :] = np.nan
        obs_label += " +3 non-finite robots"
    return dict(c, desc=desc, q=q, qd=qd, goal=goal, kw=kw, eng_kw=eng_kw, dead=dead, dead_velocity=dead_velocity, obs_label=obs_label,
                pairs_fp32=pairs_fp32, side=side, obstacle_kind=obstacle_kind)


def core_from_specs(c, fk, by_function):
    """The RMP set of a case through the reference's class surface (rmp.RmpCore + the leaf classes of rmp2.py / rmp.py): sets of
    identity-map leaves and FK -> position leaves; the FK map as TaskmapByForwardKinematic, or -- by_function -- as the reference's
    TaskmapByFunction over closures of the kinematics (taskmap.py:33-42, tests/test_taskmaps.py:33-36).  None: a leaf kind this helper
    does not build."""
    from riemannian_motion_policies_amd import descriptor as D, rmp, rmp2, taskmap as T
    core = rmp.RmpCore(rmps={}, solve=c["solve"])
    goal, off = c["goal"], 0
    for i, sp in enumerate(c["specs"]):
        P = list(sp.params)      # (the very doubles the hand-built descriptor was given)
        name = f"leaf{i}"
        if sp.taskmap == D.TASKMAP_FK_POSITION:
            frame = c["table"].frame_names[sp.frame]
            first = (T.TaskmapByFunction(lambda q, f=frame: fk.forward(q, frame=f), lambda q, qd, f=frame: fk.differentiate(q, qd, frame=f))
                     if by_function else T.TaskmapByForwardKinematic(fk, frame))
            tm = T.chain_taskmaps([first, T.TaskmapFrom4x4ToPosition()])
            g = goal[:, off:off + 3]
            off += 3
            if sp.kind == D.LEAF_TARGET_ATTRACTOR:
                core.add_rmp(rmp2.TargetAttractor(g, *P[:9], taskmap=tm, name=name))
            elif sp.kind == D.LEAF_TARGET_POLICY:
                core.add_rmp(rmp.TargetPolicy(P[0], P[1], P[2], g, tm, name=name))
            else:
                return None
        elif sp.taskmap == D.TASKMAP_IDENTITY:
            if sp.kind == D.LEAF_TARGET_POLICY:
                n = c["n"]
                core.add_rmp(rmp.TargetPolicy(P[0], P[1], P[2], goal[:, off:off + n], T.IdentityTaskmap(), name=name))
                off += n
            elif sp.kind == D.LEAF_JOINT_VELOCITY_CAP:
                core.add_rmp(rmp2.JointVelocityCap(*P[:4], name=name))
            elif sp.kind == D.LEAF_JOINT_DAMPING:
                core.add_rmp(rmp2.JointDamping(*P[:3], name=name))
            elif sp.kind == D.LEAF_CSPACE_BIASING:
                core.add_rmp(rmp2.CSpaceBiasing(np.asarray(sp.vec_a, np.float32), *P[:5], name=name))
            elif sp.kind == D.LEAF_JOINT_LIMIT_AVOIDANCE:
                core.add_rmp(rmp.JointLimitAvoidance(np.asarray(sp.vec_a, np.float32), np.asarray(sp.vec_b, np.float32), P[0], P[1], name=name))
            elif sp.kind == D.LEAF_CONFIG_SPACE_BIASING:
                core.add_rmp(rmp.ConfigurationSpaceBiasing(P[0], P[1], np.asarray(sp.vec_a, np.float32), name, w=P[2]))
            else:
                return None
        else:
            return None
    return core


def round5_extras(c, eng, obstacles, got, torch, O, what):
    """What the round-4 campaign did not draw (its own list), on the case the seed fixes, from the side generator:
    the Euler task map's entry point, Engine.bind + HIP-graph capture and replay, the native obstacle exchange (real RCCL, one rank),
    the class surface with TaskmapByFunction maps.  Returns a problem string or None."""
    side, desc, q, qd, goal, R, n, t = c["side"], c["desc"], c["q"], c["qd"], c["goal"], c["R"], c["n"], c["table"]
    dead = c["dead"]
    eq = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=12345.0, posinf=2e30, neginf=-2e30), torch.nan_to_num(b, nan=12345.0, posinf=2e30, neginf=-2e30)))
    tq, tqd = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
    tg = None if goal is None else torch.from_numpy(goal).cuda()
    done = []
    if side.random() < 0.08 and not dead.any() and t.n_frames > 0:
        # rmp2_differentiate_euler: (x, xd, J, c) of [FK(frame), 4x4 -> Euler xyz] (taskmap.py:57-67) of a random frame against the oracle
        m = min(R, 64)
        fr = int(side.integers(0, t.n_frames))
        ge = [x.cpu().numpy() for x in eng.differentiate_euler(tq[:m], tqd[:m], fr)]
        we = O.differentiate_euler(desc, q[:m], qd[:m], fr)
        calm = np.abs(np.cos(we[0][:, 1])) > 0.2                       # (away from the gimbal pole, where 1 / cos(theta_y) amplifies fp32)
        scale = 1.0 + float(np.abs(qd[:m]).max()) ** 2
        dd = [float(np.abs(a - b)[calm].max()) if calm.any() else 0.0 for a, b in zip(ge, we)]
        done.append(f"euler frame {fr}")
        # (the reference's own tolerance for this Jacobian is 1e-3, tests/test_taskmaps.py:75; measured here: <= 3.3e-4)
        if dd[0] > 2e-5 or dd[1] > 5e-4 * scale or dd[2] > 1e-3 or dd[3] > 5e-3 * scale:
            return f"rmp2_differentiate_euler frame {fr}: x {dd[0]:.2e} xd {dd[1]:.2e} J {dd[2]:.2e} c {dd[3]:.2e}"
    if side.random() < 0.1:
        # Engine.bind (the pre-marshalled launch on fixed buffers), eagerly and captured into a HIP graph and replayed: bit for bit the
        # step's result -- a strict / rank-deficient handle needs rmp2_reserve before the capture (include/rmp2.h)
        out_b = torch.empty_like(tq)
        try:
            eng.reserve(R)
            launch, _ = eng.bind(tq, tqd, tg, obstacles=obstacles, out=out_b)
            launch()
            torch.cuda.synchronize()
            first = out_b.clone()
            g = torch.cuda.CUDAGraph()
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                launch_s, _ = eng.bind(tq, tqd, tg, obstacles=obstacles, out=out_b, stream=stream.cuda_stream)
                launch_s()
                stream.synchronize()
                with torch.cuda.graph(g, stream=stream):
                    launch_s()
            out_b.fill_(float("nan"))
            g.replay()
            torch.cuda.synchronize()
            done.append("bind + graph")
            if not (eq(first, torch.from_numpy(got).cuda()) and eq(out_b, first)):
                return "Engine.bind / graph replay differs from Engine.step"
        except Exception as e:   # noqa: BLE001
            from riemannian_motion_policies_amd import _native
            if not (isinstance(e, _native.Rmp2Error) and e.code == _native.ERR_UNSUPPORTED):
                return f"bind / graph capture: {type(e).__name__}: {e}"
            done.append("bind + graph REFUSED (RMP2_ERR_UNSUPPORTED): " + str(e)[:80])   # (counted in the log, not a silent skip)
    ek = c["eng_kw"]
    if side.random() < 0.25 and set(ek) == {"spheres"} and ek["spheres"].shape[1] == 4 and not dead.any():
        # the native obstacle exchange (rmp2_exchange_*: RCCL all-gather of the table + step, one call per control step) with the real
        # RCCL at one rank, depth 1 or 2, the table MOVING every step: step k must read table k -- bit for bit the plain step on it
        from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
        depth = int(side.integers(1, 3))
        K = ek["spheres"].shape[0]
        tabs = [ek["spheres"] + np.float32(0.01 * k) * np.array([1, 0, 0, 0], np.float32) for k in range(3 + depth)]
        dev_tabs = [torch.from_numpy(x).cuda() for x in tabs]
        exch = NativeObstacleExchange(K, torch.device("cuda", 0), depth=depth)
        try:
            for k in range(depth):
                exch.start(dev_tabs[k])
            out_x = torch.empty_like(tq)
            for k in range(3):
                exch.step(eng, tq, tqd, tg, out_x, next_local=dev_tabs[k + depth])
                want = eng.step(tq, tqd, tg, obstacles=eng.obstacles(spheres=dev_tabs[k]))
                torch.cuda.synchronize()
                if not eq(out_x, want):
                    return f"native exchange (depth {depth}): step {k} did not read table {k}"
            done.append(f"exchange depth {depth}")
        finally:
            exch.close()
    if side.random() < 0.5 and c["robot_kind"] in ("two_joint", "panda") and c["obstacle_kind"] == "none" and R <= 2049 and not dead.any():
        # the reference's class surface on the same set: RmpCore + leaf classes, FK maps as TaskmapByForwardKinematic and as
        # TaskmapByFunction over closures of the kinematics -- the very accelerations of the descriptor the harness built by hand
        from riemannian_motion_policies_amd import urdf as U
        from riemannian_motion_policies_amd.kinematics import UrdfForwardKinematic
        fk = UrdfForwardKinematic(*((U.PANDA_URDF, U.PANDA_ORDER) if c["robot_kind"] == "panda" else (U.TWO_JOINT_URDF, U.TWO_JOINT_ORDER)))
        old_k = os.environ.get("RMP2_KERNEL")
        if c["kernel"]:
            os.environ["RMP2_KERNEL"] = c["kernel"]
        try:
            for by_function in (False, True):
                core = core_from_specs(c, fk, by_function)
                if core is None:
                    break
                res = np.asarray(core.evaluate(q, qd))
                done.append("class surface" + (" (TaskmapByFunction)" if by_function else ""))
                if not np.array_equal(np.nan_to_num(res, nan=12345.0), np.nan_to_num(got, nan=12345.0)):
                    return f"class surface (by_function={by_function}) differs from the descriptor: max {np.nanmax(np.abs(res - got)):.3e}"
        finally:
            if old_k is None:
                os.environ.pop("RMP2_KERNEL", None)
            else:
                os.environ["RMP2_KERNEL"] = old_k
    if done:
        what["round5_extras"] = done
    return None


def run_case(seed, torch, verbose=False):
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D, _native
    from riemannian_motion_policies_amd.engine import Engine
    c = draw_case(seed)
    rng, t, specs, solve, kernel, R, n = c["rng"], c["table"], c["specs"], c["solve"], c["kernel"], c["R"], c["n"]
    what = dict(seed=seed, robot=c["robot_kind"], dof=n, frames=t.n_frames, leaves=[(s.kind, s.taskmap, s.frame) for s in specs],
                solve=solve, kernel=kernel or "default", robots=R)
    if c["desc"] is None:
        return "declined", dict(what, why=c["why"])
    desc, q, qd, goal, kw, eng_kw, dead, obs_label = (c[k] for k in ("desc", "q", "qd", "goal", "kw", "eng_kw", "dead", "obs_label"))
    dead_velocity = c["dead_velocity"]
    what["obstacles"] = obs_label
    old = os.environ.get("RMP2_KERNEL")
    if kernel:
        os.environ["RMP2_KERNEL"] = str(kernel)
    try:
        try:
            eng = Engine(desc, 0)
        finally:
            if old is None:
                os.environ.pop("RMP2_KERNEL", None)
            else:
                os.environ["RMP2_KERNEL"] = old
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in eng_kw.items() if k not in ("pair_counts", "primitive")}
        for k in ("pair_counts", "primitive"):
            if k in eng_kw:
                dev[k] = eng_kw[k]
        obstacles = eng.obstacles(**dev) if eng_kw else None
        st = torch.zeros(R, dtype=torch.int32, device="cuda")
        want_system = (rng.random() < 0.3) or verbose
        M = torch.empty((R, n, n), dtype=torch.float64, device="cuda") if want_system else None
        f = torch.empty((R, n), dtype=torch.float64, device="cuda") if want_system else None
        extra = dict(M=M, f=f) if want_system else {}
        out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), None if goal is None else torch.from_numpy(goal),
                       obstacles=obstacles, status=st, **extra)
        torch.cuda.synchronize()
        what["ran"] = eng.last_kernel()
    except _native.Rmp2Error as e:
        if e.code == _native.ERR_UNSUPPORTED:
            return "declined", dict(what, why=str(e))
        return "failed", dict(what, why=f"Rmp2Error {e.code}: {e}")
    except ValueError as e:
        return "declined", dict(what, why=f"host: {e}")
    got, stc = out.cpu().numpy(), st.cpu().numpy()
    entry_problem = None
    if rng.random() < 0.1 and not dead.any():
        # the kinematics entry points of the ABI on this robot (rmp2_forward_kinematics, rmp2_differentiate: SURVEY 8(a) a3 / a4) --
        # every frame's transform, and (x, xd, J, c) of a random frame, against the oracle, with the structural zeros of J exact
        m = min(R, 64)
        Tg = eng.forward_kinematics(torch.from_numpy(q[:m])).cpu().numpy()
        dT = float(np.abs(Tg - O.forward_kinematics(desc, q[:m])).max())
        fr = int(rng.integers(0, t.n_frames))
        gd = [x.cpu().numpy() for x in eng.differentiate(torch.from_numpy(q[:m]), torch.from_numpy(qd[:m]), fr)]
        wd = O.differentiate(desc, q[:m], qd[:m], fr)
        scale = 1.0 + float(np.abs(qd[:m]).max()) ** 2
        dd = [float(np.abs(a - b).max()) for a, b in zip(gd, wd)]
        zeros_differ = bool(((gd[2][:, [3, 7, 11]] == 0).all(axis=(0, 1)) != (wd[2][:, [3, 7, 11]] == 0).all(axis=(0, 1))).any())
        what["entry_points"] = dict(frame=fr, T=dT, x=dd[0], xd=dd[1], J=dd[2], c=dd[3])
        if dT > 5e-6 or dd[0] > 5e-6 or dd[1] > 5e-6 * scale or dd[2] > 5e-6 or dd[3] > 2e-5 * scale or zeros_differ:
            entry_problem = f"kinematics entry points: {what['entry_points']}, zero columns differ: {zeros_differ}"
    extras_problem = round5_extras(c, eng, obstacles, got, torch, O, what)
    rollout_problem, rollout_first = None, None
    if ("p_link" not in eng_kw) and rng.random() < 0.25:
        # the fused rollout (rmp2_rollout): K control steps in one launch must equal K launches of one control step each BIT FOR BIT
        # (same kernel, same arithmetic; the state, the per-step resets and the loop carry nothing else), and its first q-double-dot
        # the plain step's (another build of the same template: to the last places)
        try:
            K, sub, dt = int(rng.integers(2, 5)), int(rng.integers(1, 4)), 0.004
            tq, tqd = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
            tg = None if goal is None else torch.from_numpy(goal).cuda()
            # half of the table cases: the obstacles MOVE between control steps (Engine.obstacle_trajectory: control step k reads
            # table k) -- K rollouts of one step, each on its own table, must still be what one rollout of K steps does
            moving = ("spheres" in eng_kw) and ("link_capsules" not in eng_kw) and rng.random() < 0.5
            per_step = [obstacles] * K
            traj = obstacles
            if moving:
                base = eng_kw["spheres"]
                drift = rng.normal(size=(1, base.shape[1])).astype(np.float32) * 0.02
                drift[:, 3] = 0.0
                tables = np.stack([base + k * drift for k in range(K)]).astype(np.float32)
                lists = {k: torch.from_numpy(eng_kw[k]) for k in ("csr_offset", "csr_index") if k in eng_kw}
                if "primitive" in eng_kw:
                    lists["primitive"] = eng_kw["primitive"]
                traj = eng.obstacle_trajectory(torch.from_numpy(tables), **lists)
                per_step = [eng.obstacles(spheres=torch.from_numpy(tables[k]), **lists) for k in range(K)]
            qa, qda = tq.clone(), tqd.clone()
            la = eng.rollout(qa, qda, tg, obstacles=traj, n_control_steps=K, substeps=sub, dt=dt).clone()
            qb, qdb = tq.clone(), tqd.clone()
            first = None
            for k in range(K):
                lb = eng.rollout(qb, qdb, tg, obstacles=per_step[k], n_control_steps=1, substeps=sub, dt=dt)
                if k == 0:
                    first = lb.clone()
            torch.cuda.synchronize()
            what["rollout"] = f"K={K} sub={sub}{' moving tables' if moving else ''} ({eng.last_kernel()[:40]})"
            same = lambda a, b: bool(torch.equal(torch.nan_to_num(a, nan=12345.0, posinf=2e30, neginf=-2e30), torch.nan_to_num(b, nan=12345.0, posinf=2e30, neginf=-2e30)))
            if not (same(qa, qb) and same(qda, qdb) and same(la, lb)):
                rollout_problem = f"rollout of {K} control steps != {K} rollouts of one (max |dq| {float(torch.nan_to_num(qa - qb).abs().max()):.3e})"
            else:
                rollout_first = first.cpu().numpy().astype(np.float64)
        except _native.Rmp2Error as e:
            if e.code != _native.ERR_UNSUPPORTED:
                rollout_problem = f"rollout: Rmp2Error {e.code}: {e}"
            else:
                what["rollout"] = "declined"
    ref = O.step(desc, q, qd, goal, **kw)
    res = O.fp32_resolution(desc, q, qd, goal, **kw)
    if c.get("pairs_fp32") is not None:
        # link geometry: the closest points of two capsules in fp32 -- the oracle on the pairs of the SAME closed form evaluated in fp32
        # arithmetic against its result on the fp64 pairs: what the closest-point stage itself can resolve (nearly parallel segments)
        r32 = O.step(desc, q, qd, goal, p_link=c["pairs_fp32"][0], p_obs=c["pairs_fp32"][1])["qdd64"]
        with np.errstate(invalid="ignore"):
            res = np.fmax(res, np.abs(r32 - ref["qdd64"]).max(axis=1))
    sys_res = O.system_resolution(ref)
    verdict = O.accuracy_gate(got, ref, spread=res, system_spread=sys_res)
    envelope_kw = {}
    if not verdict["ok"].all():
        # clause E, as tests/test_gpu_accuracy_envelope.py and bench.py judge the perf fleets (oracle.ETA came down from 1e-4 to 2e-5 with
        # it): a robot outside A-D may be no further from the fp64 evaluation than twice what 17 fp32 evaluations of the reference's
        # formulae are seen to land.  Computed only for cases that need it (17 oracle passes over the case).
        envelope_kw = dict(truth=O.step(desc, q, qd, goal, precision="f64", **kw)["qdd64"], envelope=O.fp32_envelope(desc, q, qd, goal, **kw))
        verdict = O.accuracy_gate(got, ref, spread=res, system_spread=sys_res, **envelope_kw)
    summary = O.gate_summary(verdict)
    # Robots whose system is UNDETERMINED at fp32: a singular value of the oracle's M inside (1e-18, 1e-6] x sigma_max.  M is built
    # from fp32 leaves (rmp.py:133-151: relative noise ~1e-7) and resolved in fp64 with TensorFlow's cutoff 10 n eps64 sigma_max
    # (rmp.py:153-154): a direction whose singular value is fp32 noise is KEPT, and contributes (noise of f) / (noise of M) -- in the
    # reference as here, with different noise.  Such a robot is held to what a solver can promise it: the backward error against the
    # oracle's system (omega <= oracle.ETA) and a finite answer.  Counted separately; every other robot passes the full gate.
    undetermined, tiny_alone, rel = known_classes(O, desc, specs, n, q, qd, goal, kw, ref)
    backward_ok = np.isfinite(got).all(axis=1) & (verdict["omega"] <= O.ETA)
    ok = verdict["ok"] | (undetermined & backward_ok)
    # a robot fed a non-finite state may answer NaN + status bit although the reference's graph never reaches the value
    # (include/rmp2.h, RMP2_STATUS_NONFINITE): allowed for the robots this harness poisoned, counted
    nan_flagged = ~np.isfinite(got).all(axis=1) & ((stc & D.STATUS_NONFINITE) != 0)
    summary["poisoned_input_answered_nan_where_the_oracle_stays_finite"] = int((dead & nan_flagged & ~verdict["ok"]).sum())
    ok |= dead & nan_flagged
    # (round 4, after seed 504944: the kernels put a non-finite state into the force of its dof, so neither the culling nor the
    #  quarantine can hide it -- a robot the oracle resolves to NaN must come back NaN; counted to show it stays 0)
    finite_where_oracle_nan = np.zeros(R, bool)
    summary["oracle_nan_answered_finite"] = int((~np.isfinite(ref["qdd64"]).all(axis=1) & np.isfinite(got).all(axis=1)).sum())
    # KNOWN LIMITATION: a dof whose ONLY metric is a distance leaf's, through a projection n . J_j below sqrt(eps32) of |J_j| (the
    # column nearly perpendicular to every in-range pair's direction).  The reference squares the projected scalar (relative error
    # 2 eps32 / rho); the engine pulls the leaf's summed 3 x 3 metric S = sum m n n^T back as J^T S J (section 4.2: one pull-back per
    # frame instead of one per pair), accurate to eps32 |S| |J_j|^2 ABSOLUTELY -- backward stable in S, not componentwise: the entry
    # m rho^2 |J_j|^2 is then off by eps32 / rho^2 relative.  Beside any other metric on the dof that is 6e-8 of the total; alone
    # it is the dof's whole answer.  Criterion here: a positive diagonal entry of the oracle's M below 1e-6 times the largest
    # leaf-metric scale the descriptor can produce (metric_scalar / exploder_eps) and no larger entry in its row; counted, not hidden.
    # PER DOF: the excused dof's entry is taken from the oracle and the robot goes through the SAME gate on the rest (the entry
    # dominates its row and column, so the other dofs do not feel it); a robot whose other dofs are off still fails
    lim2 = excused_by_tiny_dofs(O, got, ref, tiny_alone, ~ok, res, sys_res)
    summary["tiny_projection_alone_on_a_dof_componentwise_limit"] = int(lim2.sum())
    ok |= lim2
    summary["undetermined_at_fp32_backward_error_only"] = int((undetermined & ~verdict["ok"] & backward_ok).sum())
    summary["rejected"] = int((~ok).sum())
    problems = []
    if rollout_first is not None:
        # the first q-double-dot of the rollout (often ANOTHER mapping than the plain step's: rollouts of strict / singular sets run
        # on the hex mapping at any fleet size) goes through the same gate against the oracle, every robot
        vr = O.accuracy_gate(rollout_first, ref, spread=res, system_spread=sys_res)
        if not vr["ok"].all():
            if not envelope_kw:
                envelope_kw = dict(truth=O.step(desc, q, qd, goal, precision="f64", **kw)["qdd64"], envelope=O.fp32_envelope(desc, q, qd, goal, **kw))
            vr = O.accuracy_gate(rollout_first, ref, spread=res, system_spread=sys_res, **envelope_kw)
        fin_r = np.isfinite(rollout_first).all(axis=1)
        ok_r = vr["ok"] | (undetermined & fin_r & (vr["omega"] <= O.ETA)) | (dead & ~fin_r)
        ok_r |= excused_by_tiny_dofs(O, rollout_first, ref, tiny_alone, ~ok_r, res, sys_res)   # (the known limitation above, per dof)
        what["rollout_gate"] = {k: int(vr[k].sum()) for k in ("a", "b", "c", "d")}
        if not ok_r.all():
            badr = np.nonzero(~ok_r)[0]
            problems.append(f"rollout's first control step: {len(badr)} robot(s) outside the gate, first {badr[:5].tolist()}: err {vr['err_inf'][badr[:5]].tolist()}, "
                            f"omega {vr['omega'][badr[:5]].tolist()}, cond {vr['cond'][badr[:5]].tolist()}")
    if not ok.all():
        bad = np.nonzero(~ok)[0]
        what["bad_detail"] = [dict(robot=int(b), err=float(verdict["err_inf"][b]), omega=float(verdict["omega"][b]), cond=float(verdict["cond"][b]),
                                   resolution=float(res[b]), system_resolution=float(sys_res[b]), ref_inf=float(np.abs(ref["qdd64"][b]).max()), sv_rel_min=float(rel[b].min()),
                                   undetermined=bool(undetermined[b])) for b in bad[:8]]
        problems.append(f"{len(bad)} robot(s) outside the gate, first {bad[:5].tolist()}: err {verdict['err_inf'][bad[:5]].tolist()}, "
                        f"omega {verdict['omega'][bad[:5]].tolist()}, cond {verdict['cond'][bad[:5]].tolist()}, resolution {res[bad[:5]].tolist()}")
    nonfinite_ref = ~np.isfinite(ref["qdd64"]).all(axis=1)
    nonfinite_got = ~np.isfinite(got).all(axis=1)
    flagged = (stc & D.STATUS_NONFINITE) != 0 if hasattr(D, "STATUS_NONFINITE") else (stc & 1) != 0
    if rollout_problem:
        problems.append(rollout_problem)
    if extras_problem:
        problems.append(extras_problem)
    if entry_problem:
        problems.append(entry_problem)
    if (nonfinite_ref & ~nonfinite_got & ~finite_where_oracle_nan).any():       # (the gate's both_nan branch covers the converse)
        problems.append(f"{int((nonfinite_ref & ~nonfinite_got & ~finite_where_oracle_nan).sum())} robot(s) the oracle resolves to NaN came back finite")
    if (nonfinite_got & ~flagged).any():
        problems.append(f"{int((nonfinite_got & ~flagged).sum())} non-finite result(s) not flagged in the status word")
    if (flagged & ~nonfinite_got).any():
        problems.append(f"{int((flagged & ~nonfinite_got).sum())} robot(s) flagged non-finite with a finite result")
    if want_system:
        Mg, fg = M.cpu().numpy(), f.cpu().numpy()
        okr = np.isfinite(ref["M"]).all(axis=(1, 2)) & np.isfinite(Mg).all(axis=(1, 2)) & ~nonfinite_ref
        if okr.any():
            sM = np.abs(ref["M"][okr]).max(axis=(1, 2))
            eM = np.abs(Mg[okr] - ref["M"][okr]).max(axis=(1, 2)) / np.maximum(sM, 1e-30)
            what["system_M_rel_err_max"] = float(eM.max())
        if verbose and not ok.all():
            # split the disagreement: the engine's accumulated system against the oracle's, and the engine's resolve against
            # numpy's pseudo-inverse (TensorFlow's cutoff) of the ENGINE's own system
            for b in np.nonzero(~ok)[0][:8]:
                Mb, fb = Mg[b], fg[b]
                svb = np.linalg.svd(Mb, compute_uv=False)
                own = np.linalg.pinv(Mb, rcond=10.0 * n * np.finfo(np.float64).eps) @ fb
                print(json.dumps(dict(robot=int(b), status=int(stc[b]), M_rel=float(np.abs(Mb - ref["M"][b]).max() / max(np.abs(ref["M"][b]).max(), 1e-300)),
                                      f_rel=float(np.abs(fb - ref["f"][b]).max() / max(np.abs(ref["f"][b]).max(), 1e-300)),
                                      sv_engine=(svb / svb[0]).tolist(), sv_oracle=rel[b].tolist(),
                                      got=got[b].tolist(), ref=ref["qdd64"][b].tolist(), pinv_of_engine_system=own.tolist(),
                                      M_engine_diag=np.diag(Mb).tolist(), M_oracle_diag=np.diag(ref["M"][b]).tolist(),
                                      f_engine=fb.tolist(), f_oracle=ref["f"][b].tolist(), q=q[b].tolist())), flush=True)
    what["gate"] = summary
    if problems:
        return "failed", dict(what, why="; ".join(problems))
    return "passed", what


def run_pair_case(seed, torch):
    """The one-grid step of TWO engines (rmp2_step_pair: a TwoJoint and a Panda shard of one mixed rank, BASELINE config 5): random
    RMP sets for both (distance leaves on random frames, parameters jittered per leaf; the Panda's with an inertia leaf, the
    TwoJoint's with or without), a shared or ragged sphere table, fleets just over the fused grid's threshold -- each part against
    its oracle through the gate."""
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.engine import Engine, bind_pair
    rng = np.random.default_rng(seed)
    solve = str(rng.choice(["auto", "pinv"], p=[0.7, 0.3]))
    ragged = bool(rng.random() < 0.6)
    K = int(rng.choice([1, 8, 32, 100, 256]))
    parts = []
    for name, table_fn, lo, hi in (("two_joint", Cf.two_joint_table, Cf.TWO_JOINT_Q_LOW, Cf.TWO_JOINT_Q_HIGH), ("panda", Cf.panda_table, Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH)):
        t = table_fn()
        lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
        for _ in range(20):
            specs, kind = draw_specs(rng, t, lo, hi)
            specs = [s for s in specs if s.taskmap != D.TASKMAP_FK_POINT]
            if name == "panda" and not any(s.kind == D.LEAF_JOINT_DAMPING for s in specs):
                specs.append(D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, jitter(rng, Cf.JOINT_DAMPING_PARAMS)))
            if any(s.taskmap == D.TASKMAP_FK_DISTANCE for s in specs) and sum(s.goal_len for s in specs) <= 16:
                break
        else:
            return "declined", dict(seed=seed, why="no set with distance leaves drawn")
        desc = D.build_desc(t, specs, solve)
        R = int(rng.choice([8193, 8200, 9001]))
        n = t.n_dof
        span = hi - lo
        q = rng.uniform(lo + 0.05 * span, hi - 0.05 * span, (R, n)).astype(np.float32)
        qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
        goal = rng.uniform(-0.8, 0.8, (R, desc.goal_floats)).astype(np.float32) if desc.goal_floats else None
        parts.append(dict(name=name, table=t, specs=specs, desc=desc, R=R, q=q, qd=qd, goal=goal))
    sph = np.concatenate([rng.uniform([-1.0, -1.0, -0.2], [1.0, 1.0, 1.2], (K, 3)), rng.uniform(0.03, 0.1, (K, 1))], axis=1).astype(np.float32)
    what = dict(seed=seed, solve=solve, ragged=ragged, K=K, leaves={p["name"]: [(s.kind, s.taskmap, s.frame) for s in p["specs"]] for p in parts},
                robots=[p["R"] for p in parts])
    try:
        for p in parts:
            p["eng"] = Engine(p["desc"], 0)
            kw = dict(spheres=sph)
            if ragged:
                counts = rng.integers(0, min(K, 24) + 1, size=p["R"])
                off = np.zeros(p["R"] + 1, np.int32)
                off[1:] = np.cumsum(counts)
                idx = np.concatenate([rng.integers(0, K, size=int(k)) for k in counts] + [np.zeros(0, np.int64)]).astype(np.int32)
                kw.update(csr_offset=off, csr_index=idx)
            p["kw"] = kw
            p["obs"] = p["eng"].obstacles(**{k: torch.from_numpy(v) for k, v in kw.items()})
            p["dev"] = [torch.from_numpy(p[k]).cuda() if p[k] is not None else None for k in ("q", "qd", "goal")]
            p["out"] = torch.empty((p["R"], p["table"].n_dof), dtype=torch.float32, device="cuda")
        a, b = parts
        launch = bind_pair(a["eng"], a["dev"][0], a["dev"][1], a["dev"][2], a["obs"], a["out"], b["eng"], b["dev"][0], b["dev"][1], b["dev"][2], b["obs"], b["out"])
        for p in parts:
            p["out"].fill_(float("nan"))
        launch()
        torch.cuda.synchronize()
    except Exception as e:   # noqa: BLE001
        from riemannian_motion_policies_amd import _native
        if isinstance(e, _native.Rmp2Error) and e.code == _native.ERR_UNSUPPORTED:
            return "declined", dict(what, why=str(e))
        return "failed", dict(what, why=f"{type(e).__name__}: {e}")
    what["ran"] = b["eng"].last_kernel()
    problems = []
    for p in parts:
        got = p["out"].cpu().numpy()
        ref = O.step(p["desc"], p["q"], p["qd"], p["goal"], **p["kw"])
        res = O.fp32_resolution(p["desc"], p["q"], p["qd"], p["goal"], **p["kw"])
        sys_res = O.system_resolution(ref)
        v = O.accuracy_gate(got, ref, spread=res, system_spread=sys_res)
        what[p["name"]] = O.gate_summary(v)
        und, tiny, _ = known_classes(O, p["desc"], p["specs"], p["table"].n_dof, p["q"], p["qd"], p["goal"], p["kw"], ref)
        fin = np.isfinite(got).all(axis=1)
        ok = v["ok"] | (und & fin & (v["omega"] <= O.ETA))
        lim = excused_by_tiny_dofs(O, got, ref, tiny, ~ok, res, sys_res)
        ok |= lim
        what[p["name"]]["undetermined_at_fp32_backward_error_only"] = int((und & ~v["ok"] & fin & (v["omega"] <= O.ETA)).sum())
        what[p["name"]]["tiny_projection_alone_on_a_dof_componentwise_limit"] = int(lim.sum())
        if not ok.all():
            bad = np.nonzero(~ok)[0]
            problems.append(f"{p['name']}: {len(bad)} robot(s) outside the gate, first {bad[:4].tolist()}: err {v['err_inf'][bad[:4]].tolist()}, cond {v['cond'][bad[:4]].tolist()}")
    if problems:
        return "failed", dict(what, why="; ".join(problems))
    return "passed", what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs=2, default=[0, 200])
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--log", default="")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--pairs", action="store_true", help="run the seeds as PAIR cases (two engines in one grid, rmp2_step_pair)")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available(), "the fuzz campaign needs a HIP device"
    t0 = time.time()
    counts = {"passed": 0, "declined": 0, "failed": 0}
    by_kernel, by_obstacles, by_extra, declined_why, failures = {}, {}, {}, {}, []
    robots = 0
    log = open(args.log, "w") if args.log else None
    last_note = t0
    done = 0
    for seed in range(args.seeds[0], args.seeds[1]):
        if time.time() - t0 > args.minutes * 60:
            break
        try:
            outcome, what = run_pair_case(seed, torch) if args.pairs else run_case(seed, torch, args.verbose)
        except Exception as e:   # noqa: BLE001 -- a crash of the harness or the oracle on a case is a finding too
            import traceback
            outcome, what = "failed", dict(seed=seed, why=f"{type(e).__name__}: {e}", trace=traceback.format_exc(limit=4))
        done += 1
        counts[outcome] += 1
        if outcome == "passed":
            robots += sum(what["robots"]) if isinstance(what["robots"], list) else what["robots"]
            k = what.get("ran", "?").split("<")[0].split(" (")[0][:48]
            by_kernel[k] = by_kernel.get(k, 0) + 1
            o = what["obstacles"].split(" ")[0] if "obstacles" in what else ("ragged" if what.get("ragged") else "spheres")
            by_obstacles[o] = by_obstacles.get(o, 0) + 1
            for x in what.get("round5_extras", []):   # the side checks that ran (and the ones the library refused)
                x = x.split(":")[0].rstrip("0123456789 ") if not x.startswith("bind + graph REFUSED") else "bind + graph REFUSED"
                by_extra[x] = by_extra.get(x, 0) + 1
        elif outcome == "declined":
            w = what["why"][:90]
            declined_why[w] = declined_why.get(w, 0) + 1
        else:
            failures.append(what)
        line = json.dumps(dict(outcome=outcome, **what), default=str)
        if log:
            log.write(line + "\n")
            log.flush()
        if args.verbose or outcome == "failed":
            print(line, flush=True)
        if time.time() - last_note > 45:
            print(f"[{time.time() - t0:5.0f} s] {done} cases: {counts}", flush=True)
            last_note = time.time()
    summary = dict(cases=done, seeds=[args.seeds[0], args.seeds[0] + done], seconds=round(time.time() - t0, 1), robots_checked=robots, **counts,
                   passed_by_kernel=by_kernel, passed_by_obstacle_interface=by_obstacles, side_checks=by_extra, declined_reasons=declined_why,
                   failed_seeds=[f.get("seed") for f in failures])
    print(json.dumps(summary, indent=1))
    if log:
        log.write(json.dumps(dict(summary=summary)) + "\n")
        log.close()
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
