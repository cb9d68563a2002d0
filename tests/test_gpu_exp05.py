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
    into one pull-back per frame), both resolves (hex + pinv = its strict careful path; the quad mapping's resolve is AUTO:
    a strict handle never reaches it)."""
    if kernel == "quad" and solve == "pinv":
        pytest.skip("solve = pinv is served by the lane (3..9 dofs) / hex mappings")
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
