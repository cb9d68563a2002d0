"""GPU parity for SURVEY 8(a) rows a11 + a21: the chain [FK(frame), TaskmapRelative4x4, 4x4->position] with the
CollisionAvoidance leaf (TwoJoint experiment 05, experiments/two_joint_robot/05_obstacle_avoidance.py:44-61).

Checkers: the committed nested-autograd golden vectors (tests/golden/exp05.npz) and the C oracle.
Tolerance: |qdd_hip - qdd_ref| <= 1e-5 * max(1, |qdd_ref|_inf) (BASELINE north_star).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-5
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_mod(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _check(got, ref, what):
    err = np.abs(np.asarray(got, np.float64) - ref).max(axis=-1)
    tol = ATOL * np.maximum(1.0, np.abs(ref).max(axis=-1))
    assert (err <= tol).all(), f"{what}: worst {err.max():.3e}"


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


@pytest.mark.parametrize("kernel", ["hex", "lane", "quad"])
@pytest.mark.parametrize("solve", ["auto", "pinv"])
@pytest.mark.parametrize("key", ["tj", "pd"])
def test_exp05_golden(torch_mod, golden_dir, key, solve, kernel):
    """All three mappings carry the attached-point leaves (the quad mapping since round 3: the per-pair Jacobians collapsed
    into one pull-back per frame), both resolves (hex + pinv = its strict careful path; quad + pinv = the closed-form 2 x 2
    pseudo-inverse for the TwoJoint robot, and the quad mapping up to the combined system followed by rmp2_pinv_kernel for the
    Panda -- round 4: round 3 skipped that combination)."""
    torch = torch_mod
    from riemannian_motion_policies_amd import configs as Cf
    g = np.load(os.path.join(golden_dir, "exp05.npz"))
    _, desc = Cf.exp05_two_joint(solve) if key == "tj" else Cf.exp05_panda(solve)
    eng = _engine(desc, kernel)
    R, n = g[f"{key}_q"].shape
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda")
    f = torch.empty((R, n), dtype=torch.float64, device="cuda")
    o = eng.obstacles(p_link=torch.from_numpy(g[f"{key}_rel"]), p_obs=torch.from_numpy(g[f"{key}_nvec"]),
                      dist=torch.from_numpy(g[f"{key}_dist"]))
    qdd = eng.step(torch.from_numpy(g[f"{key}_q"]), torch.from_numpy(g[f"{key}_qd"]), torch.from_numpy(g[f"{key}_goal"]),
                   obstacles=o, M=M, f=f)
    torch.cuda.synchronize()
    assert {"hex": "hex", "lane": "one lane", "quad": "quad"}[kernel] in eng.last_kernel(), eng.last_kernel()
    assert np.abs(M.cpu().numpy() - g[f"{key}_M"]).max() < 5e-6
    assert np.abs(f.cpu().numpy() - g[f"{key}_f"]).max() < 2e-6
    _check(qdd.cpu().numpy(), g[f"{key}_qdd"], f"exp05 {key}/{solve}/{kernel}")


@pytest.mark.parametrize("R,B", [(1, 1), (65, 3), (1000, 5), (30000, 2)])
def test_exp05_batches_vs_oracle(torch_mod, R, B):
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(R)
    _, desc = Cf.exp05_panda()
    s = Cf.sample_panda_states(rng, R)
    rel, nv, dist = Cf.sample_point_pairs(rng, R, len(D.distance_leaf_indices(desc)), B)
    eng = Engine(desc, 0)
    o = eng.obstacles(p_link=torch.from_numpy(rel), p_obs=torch.from_numpy(nv), dist=torch.from_numpy(dist))
    qdd = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), obstacles=o)
    torch.cuda.synchronize()
    # default dispatch: the quad mapping at every fleet size (round 2: hex up to 20 480 robots, the lane-per-robot kernel beyond)
    assert "quad" in eng.last_kernel(), eng.last_kernel()
    sub = slice(0, min(R, 1500))
    ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], p_link=rel[sub], p_obs=nv[sub], dist=dist[sub])
    _check(qdd.cpu().numpy()[sub], ref["qdd64"], f"exp05 panda R={R} B={B}")


def test_exp05_script_through_class_surface(torch_mod):
    """The experiment-05 control loop body written with the reference's names (compat shims): Datamanager.update
    with calculate_distances-style tuples (simulation.py:481-483) -> RmpCore.evaluate(q, qd).numpy()."""
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "compat"))
    try:
        import data_management
        import kinematics
        import rmp
        import taskmap
    finally:
        sys.path.pop(0)
    from riemannian_motion_policies_amd import configs as Cf, urdf
    fkine = kinematics.UrdfForwardKinematic(urdf_filepath=urdf.TWO_JOINT_URDF, order=urdf.TWO_JOINT_ORDER)
    data_manager = data_management.Datamanager(fkine)
    core = rmp.RmpCore()
    ee = taskmap.chain_taskmaps([taskmap.TaskmapByForwardKinematic(fkine, frame='link_23'),
                                 taskmap.TaskmapFrom4x4ToPosition()])
    goal = [1.4, -1.4, 0.1]
    core.add_rmp(rmp.TargetPolicy(alpha=0.1, beta=0.1, c=0.1, goal=goal, name='target', taskmap=ee))
    for frame in fkine.frame_names:
        tm = taskmap.chain_taskmaps([taskmap.TaskmapByForwardKinematic(fkine, frame),
                                     taskmap.TaskmapRelative4x4(relative_pos=data_manager[frame]['relative_position']),
                                     taskmap.TaskmapFrom4x4ToPosition()])
        core.add_rmp(rmp.CollisionAvoidance(d=data_manager[frame]['distance'], vec=data_manager[frame]['normal_vec'],
                                            eta_rep=0.1 * np.e, nu_rep=0.3, eta_damp=1, nu_damp=0.3, r=1.1, c=1e5,
                                            taskmap=tm, name=f'collision_avoidance_for_{frame}'))
    rng = np.random.default_rng(8)
    _, desc = Cf.exp05_two_joint()
    for _ in range(3):
        q = rng.uniform(-2.0, 2.0, 2).astype(np.float32)
        qd = rng.uniform(-0.1, 0.1, 2).astype(np.float32)
        T = O.forward_kinematics(desc, q[None, :], "f64")[0]                 # [F,4,4]
        tuples, rel_ref = [], {}
        for i, frame in enumerate(fkine.frame_names):
            p_link = (T[i, :3, 3] + rng.uniform(-0.1, 0.1, 3)).astype(np.float32)
            p_obs = np.array([1.6, -0.8, 0.1], np.float32)                   # the cylinder of 05_obstacle_avoidance.py:31
            nvec = (p_link - p_obs) / np.linalg.norm(p_link - p_obs)
            dist = np.float32(np.linalg.norm(p_link - p_obs) - 0.1)
            tuples.append((frame, p_link, p_obs, nvec.astype(np.float32), dist, f'{frame} to obstacle'))
            rel_ref[frame] = T[i, :3, :3].T @ (p_link - T[i, :3, 3])
        data_manager.update(q, tuples)
        for frame in fkine.frame_names:                                       # data_management.py:44-53
            assert np.abs(data_manager[frame]['relative_position'].numpy()[0] - rel_ref[frame]).max() < 1e-6
        qdd = core.evaluate(q, qd).numpy()
        assert qdd.shape == (2,)
        rel = np.stack([data_manager[f]['relative_position'].numpy() for f in fkine.frame_names], axis=0).reshape(1, 3, 3)
        nv = np.stack([t[3] for t in tuples]).reshape(1, 3, 3)
        dd = np.array([[t[4] for t in tuples]], np.float32)
        ref = O.step(desc, q[None, :], qd[None, :], np.asarray(goal, np.float32)[None, :], p_link=rel, p_obs=nv, dist=dd)
        _check(qdd[None, :], ref["qdd64"], "exp05 script")


def test_exp05_abi_errors(torch_mod):
    torch = torch_mod
    from riemannian_motion_policies_amd import _native, configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.engine import Engine
    _, desc = Cf.exp05_two_joint()
    eng = Engine(desc, 0)
    lib = _native.lib()
    R = 4
    q = torch.zeros((R, 2), device="cuda")
    goal = torch.zeros((R, 3), device="cuda")
    pl = torch.zeros((R, 3, 3), device="cuda")
    res = D.Outputs()
    res.qdd = torch.zeros_like(q).data_ptr()
    o = eng.obstacles(p_link=pl, p_obs=pl)                                    # no dist
    rc = lib.rmp2_step(eng._h, q.data_ptr(), q.data_ptr(), goal.data_ptr(), 3, C.byref(o), C.byref(res), R, None)
    assert rc == -1 and b"dist" in lib.rmp2_last_error(eng._h)
    sph = eng.obstacles(spheres=torch.zeros((1, 4), device="cuda"))           # table modes carry no point data
    rc = lib.rmp2_step(eng._h, q.data_ptr(), q.data_ptr(), goal.data_ptr(), 3, C.byref(sph), C.byref(res), R, None)
    assert rc == -1
    o = eng.obstacles(p_link=pl, p_obs=pl, dist=torch.ones((R, 3), device="cuda"))
    cfg = D.RolloutCfg(2, 1, 0.01)
    rc = lib.rmp2_rollout(eng._h, q.data_ptr(), q.data_ptr(), goal.data_ptr(), 3, C.byref(o), C.byref(cfg), C.byref(res), R,
                          None)
    assert rc == -2 and b"rollout" in lib.rmp2_last_error(eng._h)


# ---- attached-point leaves fed from a primitive table + link capsules (round 4): pairs formed inside the step, per step ----

def _point_fields(desc, lc, prims, q):
    """What the reference's loop feeds the attached-point leaves each control step (05_obstacle_avoidance.py:51-72:
    Simulation.calculate_distances -> Datamanager.update), in fp64 numpy, independent of the engine: closest points of every
    (link capsule, primitive) pair (configs.pairs_from_link_capsules, pinned by a brute-force scan in tests/test_oracle_pins.py),
    then distance = |p_link - p_obs|, normal_vec = (p_link - p_obs) / distance, relative_position = R^T (p_link - p_frame)
    (data_management.py:33-53)."""
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    T = O.forward_kinematics(desc, q, precision="f64")[:, frames]
    pl, po = Cf.pairs_from_link_capsules(T, lc, prims)
    K = prims.shape[0]
    pl64, po64 = pl.astype(np.float64), po.astype(np.float64)
    diff = pl64 - po64
    dist = np.linalg.norm(diff, axis=-1)
    nvec = diff / dist[..., None]
    Tr = np.repeat(T, K, axis=1)                                   # frame of every pair
    rel = np.einsum("rpji,rpj->rpi", Tr[:, :, :3, :3], pl64 - Tr[:, :, :3, 3])
    return rel.astype(np.float32), nvec.astype(np.float32), dist.astype(np.float32)


def _exp05_case(robot, prim, R, seed=5):
    from riemannian_motion_policies_amd import configs as Cf, urdf as U
    rng = np.random.default_rng(seed)
    if robot == "tj":
        table, desc = Cf.exp05_two_joint()
        s = Cf.sample_two_joint_states(rng, R)
        lc = U.link_capsules(U.TWO_JOINT_URDF, table, table.frame_names)
        # obstacles beside the planar arm (reach 2, z ~ 0.1): within the leaf's radius r = 1.1 of some link, clear of contact
        ctr = np.array([[1.0, 0.8, 0.6], [-0.9, 1.1, 0.5], [0.3, -1.4, 0.45], [-1.3, -0.5, 0.6]])
    else:
        table, desc = Cf.exp05_panda()
        s = Cf.sample_panda_states(rng, R)
        lc = U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES)
        ctr = np.array([[0.75, 0.45, 0.95], [-0.55, 0.5, 0.9], [0.1, -0.85, 0.8], [0.85, -0.4, 0.25]])
    if prim == "spheres":
        prims = np.concatenate([ctr, np.full((len(ctr), 1), 0.12)], axis=1).astype(np.float32)
    else:
        axis = rng.normal(size=ctr.shape)
        axis[:, 2] *= 0.3                                              # (mostly horizontal: the clearance above the arm is kept)
        axis *= 0.15 / np.linalg.norm(axis, axis=1, keepdims=True)
        prims = np.concatenate([ctr - axis, np.full((len(ctr), 1), 0.1), ctr + axis, np.zeros((len(ctr), 1))], axis=1).astype(np.float32)
    return table, desc, s, lc.astype(np.float32), prims


@pytest.mark.parametrize("kernel", ["quad", "hex"])
@pytest.mark.parametrize("prim", ["spheres", "capsules"])
@pytest.mark.parametrize("robot,R", [("tj", 333), ("pd", 97), ("pd", 20000)])
def test_attached_point_leaves_fed_from_a_table(torch_mod, robot, R, prim, kernel):
    """TaskmapRelative4x4 + CollisionAvoidance (taskmap.py:79-99, rmp.py:264-315) with the Datamanager fields formed INSIDE the
    step from a primitive table and the leaves' link capsules: the same q-double-dot as the explicit arrays holding those
    fields (computed in fp64 numpy, independent of the engine), and as the oracle on them."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd.engine import Engine
    if kernel == "hex" and R > 5000:
        pytest.skip("the 16-lanes-per-robot mapping is the latency build: small fleets")
    table, desc, s, lc, prims = _exp05_case(robot, prim, R)
    eng = Engine(desc, 0) if kernel == "quad" else _engine(desc, kernel)   # (quad is what the dispatcher picks by itself)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    fused = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(prims), link_capsules=torch.from_numpy(lc)))
    torch.cuda.synchronize()
    assert kernel in eng.last_kernel(), eng.last_kernel()
    n = min(R, 1024)
    rel, nvec, dist = _point_fields(desc, lc, prims, s["q"][:n])
    assert (dist > 0.02).all() and (dist < 1.1).mean() > 0.2, "the case must keep pairs inside the leaf's radius, clear of contact"
    ref = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], p_link=rel, p_obs=nvec, dist=dist)
    verdict = O.accuracy_gate(fused[:n].cpu().numpy(), ref)
    assert (verdict["a"] | verdict["b"]).all(), f"table-fed vs oracle: {O.gate_summary(verdict)}"
    assert verdict["a"].mean() > 0.9, O.gate_summary(verdict)
    arrays = eng.step(q[:n], qd[:n], goal[:n], obstacles=eng.obstacles(p_link=torch.from_numpy(rel), p_obs=torch.from_numpy(nvec),
                                                                        dist=torch.from_numpy(dist)))
    torch.cuda.synchronize()
    v2 = O.accuracy_gate(fused[:n].cpu().numpy(), dict(ref, qdd64=arrays.cpu().numpy().astype(np.float64)))
    assert (v2["a"] | v2["b"]).all(), f"table-fed vs explicit arrays: {O.gate_summary(v2)}"
    # the obstacles matter: without them the step differs
    far = prims.copy()
    far[:, :3] += 50.0
    if prim == "capsules":
        far[:, 4:7] += 50.0
    none = eng.step(q[:n], qd[:n], goal[:n], obstacles=eng.obstacles(spheres=torch.from_numpy(far), link_capsules=torch.from_numpy(lc)))
    torch.cuda.synchronize()
    assert (none - fused[:n]).abs().max().item() > 1e-2


def _oracle_rollout_points(desc, lc, prims, q, qd, goal, K, sub, dt, precision="f32"):
    """The reference's exp-05 loop (05_obstacle_avoidance.py:51-72) with the CPU oracle as the controller: every control step
    the closest points are taken anew (fp64 closed form), the Datamanager fields derived, the oracle stepped; `sub` plant
    ticks in fp32 fused-multiply-add arithmetic as in tests/test_gpu_dropin.py."""
    import oracle as O
    q, qd = q.astype(np.float32).copy(), qd.astype(np.float32).copy()
    dt32 = np.float32(dt)
    peak = np.zeros(len(q))

    def fma(a, b, c):
        return (np.float64(a) * np.float64(b) + np.float64(c)).astype(np.float32)
    for _ in range(K):
        rel, nvec, dist = _point_fields(desc, lc, prims, q)
        with np.errstate(all="ignore"):
            qdd = O.step(desc, q, qd, goal, precision=precision, p_link=rel, p_obs=nvec, dist=dist)["qdd64"].astype(np.float32)
            peak = np.maximum(peak, np.nan_to_num(np.abs(qdd).max(axis=1), nan=np.inf))
            for _ in range(sub):
                qd = fma(dt32, qdd, qd)
                q = fma(dt32, qd, q)
    return q, qd, peak


@pytest.mark.parametrize("kernel", ["quad", "hex"])
@pytest.mark.parametrize("prim", ["spheres", "capsules"])
@pytest.mark.parametrize("robot", ["tj", "pd"])
def test_attached_point_leaves_roll_out(torch_mod, robot, prim, kernel):
    """rmp2_rollout with attached-point leaves (round 3 refused it: their pair data is per control step): the pairs are formed
    from the table and the link capsules every control step inside the launch, against the oracle's closed loop with the
    closest points taken anew each step -- K = 3 / 10 / 40 control steps, tolerances and the 'tame trajectory' rule of
    tests/test_gpu_dropin.py::test_fused_rollout_against_the_oracle."""
    torch = torch_mod
    from riemannian_motion_policies_amd.engine import Engine
    R = 64
    table, desc, s, lc, prims = _exp05_case(robot, prim, R, seed=9)
    eng = Engine(desc, 0) if kernel == "quad" else _engine(desc, kernel)
    goal = torch.from_numpy(s["goal"]).cuda()
    obs = eng.obstacles(spheres=torch.from_numpy(prims), link_capsules=torch.from_numpy(lc))
    sub, dt = 10, 0.01
    for K, tol_q, tol_v in ((3, 1e-5, 1e-5), (10, 1e-5, 1e-4), (40, 1e-3, 1e-3)):
        q_ref, qd_ref, peak = _oracle_rollout_points(desc, lc, prims, s["q"], s["qd"], s["goal"], K, sub, dt)
        qf, qdf = torch.from_numpy(s["q"]).cuda(), torch.from_numpy(s["qd"]).cuda()
        st = torch.zeros(R, dtype=torch.int32, device="cuda")
        eng.rollout(qf, qdf, goal, obstacles=obs, n_control_steps=K, substeps=sub, dt=dt, status=st)
        torch.cuda.synchronize()
        assert kernel in eng.last_kernel(), eng.last_kernel()
        tame = np.isfinite(q_ref).all(axis=1) & np.isfinite(qd_ref).all(axis=1) & (peak <= 20.0)
        if K > 3:   # (sets without an inertia leaf amplify rounding sooner: the oracle's own fp32 / fp64 agreement decides)
            q64, qd64, _ = _oracle_rollout_points(desc, lc, prims, s["q"], s["qd"], s["goal"], K, sub, dt, precision="f64")
            with np.errstate(all="ignore"):
                tame &= (np.abs(q64 - q_ref).max(axis=1) <= 0.1 * tol_q * np.maximum(1.0, np.abs(q_ref).max(axis=1))) & \
                        (np.abs(qd64 - qd_ref).max(axis=1) <= 0.1 * tol_v * np.maximum(1.0, np.abs(qd_ref).max(axis=1)))
        assert tame.mean() > 0.5, f"K={K}: only {tame.mean():.2f} of the oracle trajectories are tame"
        eq = np.abs(qf.cpu().numpy() - q_ref).max(axis=1) / np.maximum(1.0, np.abs(q_ref).max(axis=1))
        ev = np.abs(qdf.cpu().numpy() - qd_ref).max(axis=1) / np.maximum(1.0, np.abs(qd_ref).max(axis=1))
        assert (eq[tame] <= tol_q).all() and (ev[tame] <= tol_v).all(), \
            f"{robot} {prim} K={K}: q err {eq[tame].max():.2e}, qd err {ev[tame].max():.2e} (tame {tame.sum()})"
        assert np.abs(q_ref[tame] - s["q"][tame]).max() > 1e-3   # the fleet moved
    # explicit arrays still cannot roll out (their data is one control step's): refused with a message that names the way
    rel, nvec, dist = _point_fields(desc, lc, prims, s["q"])
    with pytest.raises(Exception, match="rollout"):
        eng.rollout(torch.from_numpy(s["q"]).cuda(), torch.from_numpy(s["qd"]).cuda(), goal,
                    obstacles=eng.obstacles(p_link=torch.from_numpy(rel), p_obs=torch.from_numpy(nvec), dist=torch.from_numpy(dist)),
                    n_control_steps=2, substeps=sub, dt=dt)


@pytest.mark.parametrize("kernel", ["quad", "hex"])
def test_attached_point_leaf_with_a_sphere_centred_on_the_link_axis(torch_mod, kernel):
    """Round-4 advisor finding: the nearest points of link axis and primitive coincide (a sphere centred ON the axis) -- there is no
    common normal, and 1 / |X - Y| made normal and lever arm NaN.  The fields take the fixed direction +z instead: finite, and the
    same as the explicit arrays built with that convention (configs.pairs_from_link_capsules)."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D
    from riemannian_motion_policies_amd.engine import Engine
    table, desc, s, lc, prims = _exp05_case("tj", "spheres", 64)
    eng = Engine(desc, 0) if kernel == "quad" else _engine(desc, kernel)
    # robot 0 at q = 0 (its kinematics are exact in fp32: products with 0 and 1): the origin of link_23 -- a zero-length link
    # capsule -- is (2, 0, 0.075 + 0.05) = (2, 0, 0.125) in fp32, and sphere 0 is centred exactly there
    s["q"][0] = 0.0
    prims = prims.copy()
    prims[0, :3] = [2.0, 0.0, np.float32(0.075) + np.float32(0.05)]
    assert (lc[2, 0:3] == 0).all() and (lc[2, 4:7] == 0).all() and prims[0, 2] == np.float32(0.125)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    st = torch.zeros(64, dtype=torch.int32, device="cuda")
    fused = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(prims), link_capsules=torch.from_numpy(lc)), status=st)
    torch.cuda.synchronize()
    got = fused.cpu().numpy()
    assert np.isfinite(got).all() and not (st.cpu().numpy() & D.STATUS_NONFINITE).any()
    rel, nvec, dist = _point_fields(desc, lc, prims, s["q"])
    assert np.isfinite(rel).all() and np.isfinite(nvec).all()
    ref = O.step(desc, s["q"], s["qd"], s["goal"], p_link=rel, p_obs=nvec, dist=dist)
    # (the fp64 kinematics put the origin at z = 0.1250000037: 3.7e-9 above the sphere's centre -- the same +z the convention takes)
    assert abs(dist[0, 2 * len(prims)] - (0.075 + 0.12)) < 1e-6 and np.allclose(nvec[0, 2 * len(prims)], [0.0, 0.0, -1.0])
    verdict = O.accuracy_gate(got, ref)
    assert (verdict["a"] | verdict["b"]).all(), O.gate_summary(verdict)
