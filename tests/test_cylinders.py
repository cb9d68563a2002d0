"""The reference's own obstacle primitive: a finite cylinder with FLAT caps (simulation.py:245-261 `Cylinder`: pybullet.GEOM_CYLINDER;
seven of them in experiments/franka_panda/06_cluttered_environment.py:39-52).  RMP2_PRIM_CYLINDER = (centre, radius, unit axis, half
height) in the table modes (the control point against the cylinder's SURFACE: side, cap or rim, signed distance along the outward
normal) and in the closest-point stage (frame origin or link capsule against the cylinder: rmp2_closest_points[_links]).

CPU: the closed forms (C oracle fp64, numpy fp64) against a brute-force scan of the surface.  GPU: every mapping against the oracle,
the stage against the numpy closed form, and the experiment-06 scene -- the script's seven cylinders with the script's RMP set."""
import os

import numpy as np
import pytest

ATOL = 1e-5


def _brute_surface(c, n_theta=720, n_axial=201, n_radial=101):
    ctr, r, u, h = c[0:3].astype(np.float64), float(c[3]), c[4:7].astype(np.float64), float(c[7])
    t = np.array([1.0, 0, 0]) if abs(u[0]) < 0.9 else np.array([0, 1.0, 0])
    e1 = np.cross(u, t)
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(u, e1)
    th = np.linspace(0, 2 * np.pi, n_theta, endpoint=False)
    ring = np.cos(th)[:, None] * e1 + np.sin(th)[:, None] * e2
    side = (ctr + np.linspace(-h, h, n_axial)[:, None, None] * u + r * ring[None]).reshape(-1, 3)
    caps = np.concatenate([(ctr + s * h * u + np.linspace(0, r, n_radial)[:, None, None] * ring[None]).reshape(-1, 3) for s in (-1, 1)])
    return np.concatenate([side, caps])


def test_point_cylinder_closed_forms_against_a_brute_force_scan():
    """numpy fp64 (configs.point_cylinder_np) and the C oracle's fp64 table mode against a scan of the cylinder's surface: the
    distance |sd| is the scanned minimum, p = Y + sd n, |n| = 1; points inside (sd < 0), beside the lateral surface, over a cap and
    beyond the rim are all drawn; a point ON the axis is finite."""
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    rng = np.random.default_rng(5)
    cyl = Cf.sample_cylinders(rng, 5)
    cyl[0, 3], cyl[0, 7] = 0.3, 0.1            # a flat disc: the cap cases
    seen = set()
    for k in range(5):
        c = cyl[k].astype(np.float64)
        P = c[0:3] + rng.uniform(-1, 1, (300, 3)) * (c[3] + c[7]) * 1.5
        Y, n, sd = Cf.point_cylinder_np(P, c[None])
        S = _brute_surface(cyl[k])
        scan = np.array([np.linalg.norm(S - p, axis=1).min() for p in P])
        assert np.abs(np.abs(sd) - scan).max() < 2e-3 * (c[3] + c[7]), k       # (grid spacing)
        assert (np.abs(sd) <= scan + 1e-6).all()                                # the closed form is never worse than a scanned point
        assert np.allclose(Y + sd[:, None] * n, P, atol=1e-6) and np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)   # (the fp32 axis is unit to 6e-8)
        w = P - c[0:3]
        a = w @ c[4:7]
        rho = np.linalg.norm(w - a[:, None] * c[4:7], axis=1)
        seen |= {("in" if s < 0 else "out", "cap" if abs(x) > c[7] else "mid", "far" if q_ > c[3] else "near") for s, x, q_ in zip(sd, a, rho)}
    assert {("in", "mid", "near"), ("out", "mid", "far"), ("out", "cap", "near"), ("out", "cap", "far")} <= seen, seen
    Y, n, sd = Cf.point_cylinder_np(cyl[1, 0:3].astype(np.float64)[None] + 0.3 * cyl[1, 7] * cyl[1, 4:7], cyl[1][None])
    assert np.isfinite(Y).all() and np.isfinite(n).all() and sd[0] < 0
    # the C oracle's table mode uses the same closed form: one obstacle leaf on the TwoJoint robot, against explicit pairs built from the
    # numpy form (outside the cylinder the two interfaces mean the same thing)
    t, desc = Cf.config5_two_joint()
    s = Cf.sample_two_joint_states(rng, 64)
    tab = np.stack([Cf.cylinder_record([1.2, 0.9, 0.45], [0.3, 0.2, 0.0], 0.08, 0.5), Cf.cylinder_record([-1.0, -1.1, 0.5], [1.2, 0.0, 0.4], 0.1, 0.6)])
    from riemannian_motion_policies_amd import descriptor as D
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    origins = O.forward_kinematics(desc, s["q"], precision="f64")[:, frames][:, :, :3, 3]
    pl, po = Cf.pairs_from_cylinders(origins, tab)
    a = O.step(desc, s["q"], s["qd"], s["goal"], spheres=tab, primitive="cylinder", precision="f64")
    b = O.step(desc, s["q"], s["qd"], s["goal"], p_link=pl, p_obs=po, precision="f64")
    assert np.abs(a["M"] - b["M"]).max() <= 2e-5 * np.abs(b["M"]).max() and np.abs(a["f"] - b["f"]).max() <= 2e-5 * np.abs(b["f"]).max()


def test_link_capsule_cylinder_closest_points_against_a_brute_force_scan():
    """configs.pairs_from_link_capsules_cylinders (bisection on the convex distance along the link's axis, fp64) against a scan of
    the link's axis x the cylinder's surface: the returned surface points are as far apart as the scanned minimum minus the link's
    radius."""
    from riemannian_motion_policies_amd import configs as Cf
    rng = np.random.default_rng(9)
    R, Cn, K = 2, 3, 4
    T = np.tile(np.eye(4), (R, Cn, 1, 1))
    for r in range(R):
        for c in range(Cn):
            Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            T[r, c, :3, :3] = Q * np.sign(np.linalg.det(Q))
            T[r, c, :3, 3] = rng.uniform(-0.3, 0.3, 3)
    lc = np.zeros((Cn, 8), np.float32)
    lc[:, 0:3], lc[:, 4:7], lc[:, 3] = rng.uniform(-0.3, 0.3, (Cn, 3)), rng.uniform(-0.3, 0.3, (Cn, 3)), rng.uniform(0.02, 0.05, Cn)
    lc[0, 4:7] = lc[0, 0:3]                                  # a degenerate link (a point)
    cyl = np.stack([Cf.cylinder_record(rng.uniform(-0.6, 0.6, 3), rng.uniform(0, np.pi, 3), rng.uniform(0.03, 0.1), rng.uniform(0.1, 0.5)) for _ in range(K)])
    pl, po = Cf.pairs_from_link_capsules_cylinders(T, lc, cyl)
    u = np.linspace(0, 1, 301)
    for r in range(R):
        for c in range(Cn):
            A = T[r, c, :3, 3] + T[r, c, :3, :3] @ lc[c, 0:3]
            B = T[r, c, :3, 3] + T[r, c, :3, :3] @ lc[c, 4:7]
            X = A[None] + u[:, None] * (B - A)[None]
            for k in range(K):
                _, _, sd = Cf.point_cylinder_np(X, cyl[k][None])
                if sd.min() <= 0:
                    continue                                  # (an axis through the cylinder: the scan has no unique minimum)
                got = np.linalg.norm(pl[r, c * K + k].astype(np.float64) - po[r, c * K + k])
                assert abs(got - (sd.min() - lc[c, 3])) < 1e-3, (r, c, k, got, sd.min())
                assert got <= sd.min() - lc[c, 3] + 1e-6       # never worse than a scanned point of the axis


def _engine(desc, kernel):
    from riemannian_motion_policies_amd.engine import Engine
    old = os.environ.get("RMP2_KERNEL")
    if kernel:
        os.environ["RMP2_KERNEL"] = kernel
    try:
        return Engine(desc, 0)
    finally:
        if old is None:
            os.environ.pop("RMP2_KERNEL", None)
        else:
            os.environ["RMP2_KERNEL"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,R", [("hex", 500), ("quad", 500), ("lane", 500), ("", 20000)])
@pytest.mark.parametrize("K", [7, 300])
def test_cylinder_table_vs_oracle(hip_lib, kernel, R, K):
    """Shared cylinder table through every mapping (K = 7: staged in LDS, the culled quad loop tests bounding spheres; K = 300: beyond the
    LDS table), shared and ragged, against the C oracle's fp64-pinned closed form."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    rng = np.random.default_rng(60 + K + R)
    s = Cf.sample_panda_states(rng, R)
    cyl = Cf.sample_cylinders(rng, K)
    if K > 100:      # dense table: short thin cylinders
        cyl[:, 3] *= 0.1
        cyl[:, 7] *= 0.1
    _, desc = Cf.config3("pinv")
    eng = _engine(desc, kernel)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    for ragged in (False, True):
        kw = {}
        if ragged:
            off, idx = Cf.sample_ragged(rng, R, K)
            kw = dict(csr_offset=off, csr_index=idx)
        dev_kw = {k: torch.from_numpy(v) for k, v in kw.items()}
        st = torch.zeros(R, dtype=torch.int32, device="cuda")
        out = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder", **dev_kw), status=st)
        torch.cuda.synchronize()
        if kernel:
            assert kernel in eng.last_kernel(), eng.last_kernel()
        n = min(R, 1024)
        sub = {k: (v[: n + 1] if k == "csr_offset" else v) for k, v in kw.items()}
        if ragged:
            sub["csr_index"] = kw["csr_index"][: kw["csr_offset"][n]]
        okw = dict(spheres=cyl, primitive="cylinder", **sub)
        ref = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], **okw)
        truth = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], precision="f64", **okw)["qdd64"]
        v = O.accuracy_gate(out[:n].cpu().numpy(), ref, truth=truth, envelope=O.fp32_envelope(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], **okw))
        assert v["ok"].all(), f"{kernel or 'default'} K={K} ragged={ragged}: {O.gate_summary(v)}"
        assert v["a"].mean() > 0.6, O.gate_summary(v)
    # the table matters, and it is not read as capsules: the same records as a capsule table give another answer
    as_caps = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(cyl)))
    torch.cuda.synchronize()
    assert (as_caps - out).abs().nan_to_num(0.0).max().item() > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("links", [False, True])
def test_cylinder_closest_point_stage(hip_lib, links):
    """rmp2_closest_points[_links] on a cylinder table: frame origins (closed form) and link capsules (bisection on the convex
    distance) against the fp64 numpy forms; both forms of the stage (a lane per pair, a lane per robot); the arrays then drive the
    explicit-pair step -- the reference's own data flow (simulation.py:462-484 -> data_management.py:22-37 -> taskmap.py:115-138)."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D, urdf as U
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(88)
    R, K = 333, 7
    table, desc = Cf.config3()
    s = Cf.sample_panda_states(rng, R)
    cyl = Cf.sample_cylinders(rng, K)
    lc = U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES) if links else None
    eng = Engine(desc, 0)
    t_dev = eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder")
    lct = None if lc is None else torch.from_numpy(lc)
    pl, po = eng.closest_points(torch.from_numpy(s["q"]), t_dev, link_capsules=lct)
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    T = O.forward_kinematics(desc, s["q"], precision="f64")[:, frames]
    if links:
        pl_ref, po_ref = Cf.pairs_from_link_capsules_cylinders(T, lc, cyl)
    else:
        pl_ref, po_ref = Cf.pairs_from_cylinders(T[:, :, :3, 3], cyl)
    pln, pon = pl.cpu().numpy(), po.cpu().numpy()
    d_got = np.linalg.norm(pln.astype(np.float64) - pon, axis=-1)
    d_ref = np.linalg.norm(pl_ref.astype(np.float64) - po_ref, axis=-1)
    assert np.abs(d_got - d_ref).max() < 3e-6, np.abs(d_got - d_ref).max()       # the distance is well conditioned ...
    # ... WHERE along a link that runs beside a cylinder's side (or a cap) the nearest point sits is not: the minimum is flat, and an
    # fp32 and an fp64 bisection settle centimetres apart on it with the same distance and the same normal.  Nearly all pairs agree:
    dp = np.maximum(np.abs(pln - pl_ref).max(axis=-1), np.abs(pon - po_ref).max(axis=-1))
    assert np.percentile(dp, 95) < 5e-5, np.percentile(dp, [50, 95, 99, 100])
    n_got = (pln.astype(np.float64) - pon) / d_got[..., None]
    n_ref = (pl_ref.astype(np.float64) - po_ref) / d_ref[..., None]
    # (the direction the leaf reads: the same -- but for the few links whose AXIS passes through a cylinder, where side and cap are
    #  equally near somewhere along it and the two precisions may settle on different faces)
    dn = np.abs(n_got - n_ref).max(axis=-1)[d_ref > 0.01]
    assert np.percentile(dn, 99) < 2e-3 and (dn > 2e-3).mean() < 5e-3, np.percentile(dn, [50, 99, 99.9, 100])
    pl1, po1 = _engine(desc, "lane").closest_points(torch.from_numpy(s["q"]), t_dev, link_capsules=lct)
    assert (pl1 - pl).abs().max().item() < 2e-6 and (po1 - po).abs().max().item() < 2e-6
    qdd = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), obstacles=eng.obstacles(p_link=pl, p_obs=po))
    torch.cuda.synchronize()
    kw = dict(p_link=pln, p_obs=pon)
    ref = O.step(desc, s["q"], s["qd"], s["goal"], **kw)
    truth = O.step(desc, s["q"], s["qd"], s["goal"], precision="f64", **kw)["qdd64"]
    v = O.accuracy_gate(qdd.cpu().numpy(), ref, truth=truth, envelope=O.fp32_envelope(desc, s["q"], s["qd"], s["goal"], **kw))
    assert v["ok"].all(), O.gate_summary(v)
    # link geometry + a cylinder table handed to the STEP: the library runs exactly this (stage into its own buffer, explicit-pair step)
    if links:
        both = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]),
                        obstacles=eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder", link_capsules=lct))
        torch.cuda.synchronize()
        assert torch.equal(both, qdd)
        from riemannian_motion_policies_amd import _native
        with pytest.raises(_native.Rmp2Error, match="rmp2_closest_points_links"):      # (a rollout cannot: the pairs are one step's)
            eng.rollout(torch.from_numpy(s["q"]).cuda(), torch.from_numpy(s["qd"]).cuda(), torch.from_numpy(s["goal"]).cuda(),
                        obstacles=eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder", link_capsules=lct), n_control_steps=2)


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["hex", "quad"])
def test_experiment06_scene_with_its_cylinders(hip_lib, kernel):
    """experiments/franka_panda/06_cluttered_environment.py: the script's seven cylinders (:39-52), its goal (:36) and its RMP set
    (:70-114: target attractor, velocity cap, damping, c-space biasing, obstacle avoidance per collision frame), on states around the
    script's start pose -- through the class surface the script uses (RmpCore + Datamanager.update_device: closest points of the links'
    capsules and the cylinders formed on the device, then the explicit-pair step) and as a fused cylinder table with frame-origin control
    points, against the oracle."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D, urdf as U
    rng = np.random.default_rng(6)
    R = 256
    table, desc = Cf.config3("pinv")
    s = Cf.sample_panda_states(rng, R)
    s["goal"][:] = Cf.EXP06_GOAL
    cyl = Cf.EXP06_CYLINDERS
    eng = _engine(desc, kernel)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    # (a) the cylinders as a fused table, control points = frame origins
    out = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder"))
    torch.cuda.synchronize()
    assert kernel in eng.last_kernel()
    okw = dict(spheres=cyl, primitive="cylinder")
    ref = O.step(desc, s["q"], s["qd"], s["goal"], **okw)
    truth = O.step(desc, s["q"], s["qd"], s["goal"], precision="f64", **okw)["qdd64"]
    v = O.accuracy_gate(out.cpu().numpy(), ref, truth=truth, envelope=O.fp32_envelope(desc, s["q"], s["qd"], s["goal"], **okw))
    assert v["ok"].all() and v["a"].mean() > 0.9, O.gate_summary(v)
    # (b) the script's data flow: closest points of the LINKS' capsules and the cylinders (the stage), then explicit pairs
    lc = U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES)
    pl, po = eng.closest_points(q, eng.obstacles(spheres=torch.from_numpy(cyl), primitive="cylinder"), link_capsules=torch.from_numpy(lc))
    out_b = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=pl, p_obs=po))
    torch.cuda.synchronize()
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    T = O.forward_kinematics(desc, s["q"], precision="f64")[:, frames]
    pl_ref, po_ref = Cf.pairs_from_link_capsules_cylinders(T, lc, cyl)          # fp64 numpy, independent of the engine
    pln, pon = pl.cpu().numpy(), po.cpu().numpy()
    d_got, d_ref = np.linalg.norm(pln.astype(np.float64) - pon, axis=-1), np.linalg.norm(pl_ref.astype(np.float64) - po_ref, axis=-1)
    assert np.abs(d_got - d_ref).max() < 3e-6                                    # the stage's distances are the closed form's ...
    kw = dict(p_link=pln, p_obs=pon)                                             # ... and the step on its pairs is the oracle's on them
    ref_b = O.step(desc, s["q"], s["qd"], s["goal"], **kw)
    truth_b = O.step(desc, s["q"], s["qd"], s["goal"], precision="f64", **kw)["qdd64"]
    env = O.fp32_envelope(desc, s["q"], s["qd"], s["goal"], **kw)
    vb = O.accuracy_gate(out_b.cpu().numpy(), ref_b, truth=truth_b, envelope=env)
    assert vb["ok"].all() and vb["a"].mean() > 0.8, O.gate_summary(vb)
    # the link geometry matters: control points on the links' surfaces give another answer than the frame origins
    assert (out_b - out).abs().max().item() > 1e-3


@pytest.mark.gpu
def test_experiment06_loop_body_with_cylinders_through_the_class_surface(hip_lib):
    """The experiment-06 loop body as the script writes it (06_cluttered_environment.py:96-131): an ObstacleAvoidance leaf per collision
    frame on [FK(frame), TaskmapJointFrame4x4ToDistance(data_manager[frame][...])], `data_manager.update_device(core, q, cylinders,
    link_capsules, primitive="cylinder")` standing where the script calls calculate_distances + data_manager.update, then
    core.evaluate(q, qd) -- against the oracle on the fp64 closest points of the links' capsules and the script's seven cylinders."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D, rmp, rmp2, taskmap as T, urdf as U
    from riemannian_motion_policies_amd.data_management import Datamanager
    from riemannian_motion_policies_amd.kinematics import UrdfForwardKinematic
    fk = UrdfForwardKinematic(U.PANDA_URDF, U.PANDA_ORDER)
    data_manager = Datamanager(fk)
    core = rmp.RmpCore(rmps={}, solve="pinv")
    ee = T.chain_taskmaps([T.TaskmapByForwardKinematic(fk, "panda_grasptarget_hand"), T.TaskmapFrom4x4ToPosition()])
    R = 128
    s = Cf.sample_panda_states(np.random.default_rng(16), R)
    s["goal"][:] = Cf.EXP06_GOAL
    core.add_rmp(rmp2.TargetAttractor(s["goal"], *Cf.TARGET_ATTRACTOR_PARAMS, taskmap=ee, name="attractor"))
    core.add_rmp(rmp2.JointVelocityCap(*Cf.JOINT_VELOCITY_CAP_PARAMS))
    core.add_rmp(rmp2.JointDamping(*Cf.JOINT_DAMPING_PARAMS))
    core.add_rmp(rmp2.CSpaceBiasing(Cf.CSPACE_BIASING_GOAL, *Cf.CSPACE_BIASING_PARAMS))
    for frame in Cf.CONTROL_POINT_FRAMES:
        tm = T.chain_taskmaps([T.TaskmapByForwardKinematic(fk, frame),
                               T.TaskmapJointFrame4x4ToDistance(data_manager[frame]['pos_on_link_in_base_frame'],
                                                                data_manager[frame]['pos_on_obstacle_in_base_frame'])])
        core.add_rmp(rmp2.ObstacleAvoidance(*Cf.OBSTACLE_AVOIDANCE_PARAMS, taskmap=tm, name=f"collision_avoidance_for_{frame}"))
    dev = torch.device("cuda", 0)
    q, qd = torch.from_numpy(s["q"]).to(dev), torch.from_numpy(s["qd"]).to(dev)
    table = fk.table
    lc = U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES)
    cyl = torch.from_numpy(Cf.EXP06_CYLINDERS).to(dev)
    data_manager.update_device(core, q, cyl, link_capsules=torch.from_numpy(lc).to(dev), primitive="cylinder")
    qdd = core.evaluate(q, qd)
    torch.cuda.synchronize()
    _, desc = Cf.config3("pinv")
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    Tw = O.forward_kinematics(desc, s["q"], precision="f64")[:, frames]
    pl, po = Cf.pairs_from_link_capsules_cylinders(Tw, lc, Cf.EXP06_CYLINDERS)
    kw = dict(p_link=pl, p_obs=po)
    ref = O.step(desc, s["q"], s["qd"], s["goal"], **kw)
    truth = O.step(desc, s["q"], s["qd"], s["goal"], precision="f64", **kw)["qdd64"]
    v = O.accuracy_gate(qdd.cpu().numpy(), ref, truth=truth, envelope=O.fp32_envelope(desc, s["q"], s["qd"], s["goal"], **kw), envelope_factor=4.0)
    assert v["ok"].all() and v["a"].mean() > 0.8, O.gate_summary(v)
    # the holders the leaves read are the stage's arrays (views, nothing copied), the distances the closed form's
    first = data_manager[Cf.CONTROL_POINT_FRAMES[0]]['pos_on_link_in_base_frame'].value
    assert first.is_cuda and tuple(first.shape) == (R, 7, 3)
    d0 = (first - data_manager[Cf.CONTROL_POINT_FRAMES[0]]['pos_on_obstacle_in_base_frame'].value).norm(dim=-1).cpu().numpy()
    d_ref = np.linalg.norm(pl[:, :7].astype(np.float64) - po[:, :7], axis=-1)
    assert np.abs(d0 - d_ref).max() < 3e-6
