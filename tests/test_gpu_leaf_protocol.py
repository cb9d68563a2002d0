"""The reference's leaf protocol  rmp.evaluate(x, xd) -> (xdd_des, A)  (rmp2.py:25-29, rmp.py:202-206; SURVEY 8(b) "leaf
protocol") through rmp2_leaf_evaluate, every leaf class, against the autograd oracle's op-for-op leaf functions.
Tolerance: 1e-5 * max(1, |.|) per row, as everywhere."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _close(got, ref, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    tol = 1e-5 * max(1.0, np.abs(ref).max())
    assert np.abs(got - ref).max() <= tol, f"{what}: {np.abs(got - ref).max():.3e} > {tol:.1e}"


def test_every_leaf_class_evaluates_like_the_oracle(hip_lib):
    import torch
    import torch_autodiff_oracle as TA
    from riemannian_motion_policies_amd import configs as Cf, rmp, rmp2, taskmap
    assert torch.cuda.is_available()
    rng = np.random.default_rng(11)
    F32 = torch.float32
    t = lambda a: torch.tensor(np.asarray(a), dtype=F32)
    B, n = 37, 9
    q = rng.uniform(Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH, (B, n)).astype(np.float32)
    q[:6, 3] = np.float32(Cf.PANDA_Q_LOW[3] + 0.02)             # inside a joint-limit band
    # |qd| in [0, 0.15] or [0.25, 0.45]: below and beyond JointVelocityCap's cut-off (0.35), clear of the Q4 pole at 0.2 and of
    # the clipped ratio beyond 0.5 (where 1 / (1 - ratio^2) ~ 7.5e4 hangs on the last bit of (region - 1e-6) / region)
    mag = np.where(rng.random((B, n)) < 0.5, rng.uniform(0.0, 0.15, (B, n)), rng.uniform(0.25, 0.45, (B, n)))
    qd = (mag * rng.choice([-1.0, 1.0], (B, n))).astype(np.float32)
    x3 = rng.uniform(-0.5, 0.8, (B, 3)).astype(np.float32)
    v3 = rng.uniform(-0.3, 0.3, (B, 3)).astype(np.float32)
    goal3 = [0.4, -0.2, 0.5]
    ident = taskmap.IdentityTaskmap()

    leaf = rmp2.TargetAttractor(goal3, *Cf.TARGET_ATTRACTOR_PARAMS, taskmap=ident)
    xdd, A = leaf.evaluate(x3, v3)
    r_xdd, r_A = TA.target_attractor(Cf.TARGET_ATTRACTOR_PARAMS, goal3, t(x3), t(v3))
    _close(xdd, r_xdd.numpy(), "TargetAttractor xdd"); _close(A.numpy(), r_A.numpy(), "TargetAttractor A")

    leaf = rmp2.JointVelocityCap(*Cf.JOINT_VELOCITY_CAP_PARAMS)
    xdd, A = leaf.evaluate(q, qd)
    r_xdd, r_A = TA.joint_velocity_cap(Cf.JOINT_VELOCITY_CAP_PARAMS, t(q), t(qd))
    _close(xdd, r_xdd.numpy(), "JointVelocityCap xdd"); _close(A, r_A.numpy(), "JointVelocityCap A")

    leaf = rmp2.JointDamping(*Cf.JOINT_DAMPING_PARAMS)
    xdd, A = leaf.evaluate(q, qd)
    r_xdd, r_A = TA.joint_damping(Cf.JOINT_DAMPING_PARAMS, t(q), t(qd))
    _close(xdd, r_xdd.numpy(), "JointDamping xdd"); _close(A, r_A.numpy(), "JointDamping A")

    leaf = rmp2.CSpaceBiasing(Cf.CSPACE_BIASING_GOAL, *Cf.CSPACE_BIASING_PARAMS)
    xdd, A = leaf.evaluate(q, qd)
    r_xdd, r_A = TA.cspace_biasing(Cf.CSPACE_BIASING_PARAMS, Cf.CSPACE_BIASING_GOAL, t(q), t(qd))
    _close(xdd, r_xdd.numpy(), "CSpaceBiasing xdd"); _close(A, r_A.numpy(), "CSpaceBiasing A")

    d = rng.uniform(0.0, 0.7, (B, 1)).astype(np.float32)        # some beyond metric_modulation_radius = 0.5 (metric exactly 0)
    dd = rng.uniform(-0.3, 0.3, (B, 1)).astype(np.float32)
    leaf = rmp2.ObstacleAvoidance(*Cf.OBSTACLE_AVOIDANCE_PARAMS, taskmap=ident, name="oa")
    xdd, A = leaf.evaluate(d, dd)
    r_xdd, r_A = TA.obstacle_avoidance(Cf.OBSTACLE_AVOIDANCE_PARAMS, t(d), t(dd))
    _close(xdd, r_xdd.numpy(), "ObstacleAvoidance xdd"); _close(A, r_A.numpy(), "ObstacleAvoidance A")
    assert (A.numpy()[d[:, 0] > 0.5] == 0).all()

    leaf = rmp.JointLimitAvoidance(Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH, *Cf.JOINT_LIMIT_PARAMS)
    xdd, A = leaf.evaluate(q, qd)
    for b in range(B):   # the oracle's version is written for one robot per call, as the reference runs
        r_xdd, r_A = TA.joint_limit_avoidance(Cf.JOINT_LIMIT_PARAMS, Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH, t(q[b:b + 1]), t(qd[b:b + 1]))
        _close(xdd[b:b + 1], r_xdd.numpy(), "JointLimitAvoidance xdd"); _close(A[b:b + 1], r_A.numpy(), "JointLimitAvoidance A")
    assert np.abs(A[0] - A[0].T).max() > 1e-3                   # column-scaled: not symmetric (quirk Q2)

    leaf = rmp.ConfigurationSpaceBiasing(*Cf.PANDA04_CONFIG_SPACE_BIASING_PARAMS[:2], q0=Cf.PANDA04_Q0, name="csb",
                                         w=Cf.PANDA04_CONFIG_SPACE_BIASING_PARAMS[2])
    xdd, A = leaf.evaluate(q, qd)
    r_xdd, r_A = TA.config_space_biasing(Cf.PANDA04_CONFIG_SPACE_BIASING_PARAMS, Cf.PANDA04_Q0, t(q), t(qd))
    _close(xdd, r_xdd.numpy(), "ConfigurationSpaceBiasing xdd")
    _close(A, np.broadcast_to(r_A.numpy(), A.shape), "ConfigurationSpaceBiasing A")

    for k, xs, vs, g in ((3, x3, v3, goal3), (n, q, qd, list(Cf.PANDA_Q_READY))):   # TargetPolicy on 3-d and on the identity map
        leaf = rmp.TargetPolicy(*Cf.TARGET_POLICY_PARAMS, goal=g, taskmap=ident)
        xdd, A = leaf.evaluate(xs, vs)
        for b in range(0, B, 5):   # global norms in the reference: one row per call
            r_xdd, r_A = TA.target_policy(Cf.TARGET_POLICY_PARAMS, g, t(xs[b:b + 1]), t(vs[b:b + 1]))
            _close(xdd[b:b + 1], r_xdd.numpy(), f"TargetPolicy k={k} xdd"); _close(A[b:b + 1], r_A.numpy(), f"TargetPolicy k={k} A")

    dist = rng.uniform(0.05, 1.3, B).astype(np.float32)         # some beyond r = 1.1
    nv = rng.normal(size=(B, 3))
    nv = (nv / np.linalg.norm(nv, axis=1, keepdims=True)).astype(np.float32)
    leaf = rmp.CollisionAvoidance(dist, nv, *Cf.COLLISION_AVOIDANCE_PARAMS, taskmap=ident)
    xdd, A = leaf.evaluate(x3, v3)
    r_xdd, r_A = TA.collision_avoidance(Cf.COLLISION_AVOIDANCE_PARAMS, dist, nv, t(x3), t(v3))
    _close(xdd, r_xdd.numpy(), "CollisionAvoidance xdd"); _close(A, r_A.numpy(), "CollisionAvoidance A")


def test_leaf_protocol_abi_errors(hip_lib):
    import torch
    from riemannian_motion_policies_amd import _native, descriptor as D
    lib = _native.lib()
    rec = D.Leaf()
    rec.kind = D.LEAF_TARGET_ATTRACTOR
    buf = torch.zeros(64, device="cuda")
    p = buf.data_ptr()
    assert lib.rmp2_leaf_evaluate(0, C.byref(rec), 9, p, p, p, None, None, p, p, 1, None) == -1    # k must be 3
    assert b"dimension" in lib.rmp2_last_error(None)
    assert lib.rmp2_leaf_evaluate(0, C.byref(rec), 3, p, p, None, None, None, p, p, 1, None) == -1  # goal missing
    rec.kind = D.LEAF_COLLISION_AVOIDANCE
    assert lib.rmp2_leaf_evaluate(0, C.byref(rec), 3, p, p, None, None, None, p, p, 1, None) == -1  # dist / nvec missing
    rec.kind = 77
    assert lib.rmp2_leaf_evaluate(0, C.byref(rec), 3, p, p, None, None, None, p, p, 1, None) == -1
    rec.kind = D.LEAF_JOINT_DAMPING
    assert lib.rmp2_leaf_evaluate(0, C.byref(rec), 5, p, p, None, None, None, p, p, 0, None) == 0   # empty batch
    assert lib.rmp2_leaf_evaluate(99, C.byref(rec), 5, p, p, None, None, None, p, p, 1, None) == -3  # no such device
