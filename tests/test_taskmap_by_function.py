"""`TaskmapByFunction(forward_fn, differentiate_fn)` with the reference's signature (taskmap.py:33-42) and `chain_taskmaps`
with the reference's chain rule (taskmap.py:142-168).

The GPU tests are the body of the reference's own tests/test_taskmaps.py:31-76 -- a hand-made FK map per joint frame
(closures over `fkine.forward` / `fkine.differentiate`), chained with TaskmapFrom4x4ToPosition and TaskmapFrom4x4ToEuler,
random joint vectors inside the limits, position Jacobian to 1e-6 and Euler Jacobian to 1e-3 -- with the CPU oracle in
PyBullet's place (PyBullet is not installed here: SURVEY 8(c))."""
import numpy as np
import pytest


def _fk():
    from riemannian_motion_policies_amd import urdf
    from riemannian_motion_policies_amd.kinematics import UrdfForwardKinematic
    return UrdfForwardKinematic(urdf.PANDA_URDF, urdf.PANDA_ORDER)


def test_closures_over_the_kinematics_are_recognised_without_a_gpu():
    """A pair of closures that is exactly fkine.forward / fkine.differentiate of a frame compiles like
    TaskmapByForwardKinematic (decided by calling them against recording stand-ins: no GPU, no source inspection); closures
    that touch q or the result are opaque and refused by the compiler with the supported spellings in the message."""
    from riemannian_motion_policies_amd import descriptor as D, taskmap as T
    fk = _fk()
    name = "panda_hand_joint"
    tm = T.TaskmapByFunction(forward_fn=lambda q: fk.forward(q, frame=name),
                             differentiate_fn=lambda q, qd: fk.differentiate(q, qd, frame=name))
    assert tm.forward_fn is not None and tm.differentiate_fn is not None          # the reference's attribute names
    kind, st0, _ = T.classify(T.chain_taskmaps([tm, T.TaskmapFrom4x4ToPosition()]))
    assert kind == D.TASKMAP_FK_POSITION and isinstance(st0, T.TaskmapByForwardKinematic) and st0.frame == name and st0.fkine is fk
    # the reference's test calls chain_taskmaps(map_1, map_2) (tests/test_taskmaps.py:39-40): taken too
    kind2, _, _ = T.classify(T.chain_taskmaps(tm, T.TaskmapFrom4x4ToPosition()))
    assert kind2 == D.TASKMAP_FK_POSITION
    # the methods of the kinematics object are what they were (the stand-ins are gone), and the stage object is stable
    assert "forward" not in fk.__dict__ and "differentiate" not in fk.__dict__
    assert tm.stages()[0] is tm.stages()[0]
    # a closure over a loop variable is read when the set is compiled (late binding, as Python has it)
    name = "panda_joint3"
    assert tm.stages()[0].frame == "panda_joint3"
    # positional frame, and a tf.constant-like frame object
    class Const:
        def numpy(self):
            return b"panda_joint5"
    tm2 = T.TaskmapByFunction(lambda q: fk.forward(q, Const()), lambda q, qd: fk.differentiate(q, qd, Const()))
    assert tm2.stages()[0].frame == "panda_joint5"
    # opaque closures: scaled result, shifted q, two different frames, no kinematics at all
    for fwd, dif in ((lambda q: 2.0 * fk.forward(q, frame="panda_joint3"), lambda q, qd: fk.differentiate(q, qd, frame="panda_joint3")),
                     (lambda q: fk.forward(q + 1.0, frame="panda_joint3"), lambda q, qd: fk.differentiate(q + 1.0, qd, frame="panda_joint3")),
                     (lambda q: fk.forward(q, frame="panda_joint3"), lambda q, qd: fk.differentiate(q, qd, frame="panda_joint4")),
                     (lambda q: q, lambda q, qd: (q, qd, None, None))):
        opaque = T.TaskmapByFunction(fwd, dif)
        assert opaque.stages() == [opaque]
        with pytest.raises(NotImplementedError, match="TaskmapByFunction\\(forward_fn=lambda q: fkine.forward"):
            T.classify(T.chain_taskmaps([opaque, T.TaskmapFrom4x4ToPosition()]))


def test_chain_rule_of_two_function_maps_on_the_host():
    """J = J2 J1, xd = J2 xd1, c = c2 + J2 c1 (taskmap.py:150-160) for maps nobody has a kernel for: a quadratic map of q
    chained with the Euler selector's closed form, against fp64 finite differences of the composed function."""
    from riemannian_motion_policies_amd import taskmap as T
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(3)
    A = rng.normal(size=(16, 5))

    def pose(q):   # q [B,5] -> vec(T) [B,16]: a rotation by the rotation vector q[:3], translated by q[3:]
        out = []
        for row in np.atleast_2d(q):
            M = np.eye(4)
            M[:3, :3] = Rotation.from_rotvec(row[:3]).as_matrix()
            M[:2, 3] = row[3:]
            out.append(M.reshape(-1))
        return np.asarray(out)

    def pose_diff(q, qd):
        q, qd = np.atleast_2d(q).astype(np.float64), np.atleast_2d(qd).astype(np.float64)
        h = 1e-5
        n = q.shape[1]
        J = np.stack([(pose(q + h * np.eye(n)[i]) - pose(q - h * np.eye(n)[i])) / (2 * h) for i in range(n)], axis=-1)
        x = pose(q)
        xd = np.einsum("bkm,bm->bk", J, qd)
        c = (pose(q + h * qd) - 2 * x + pose(q - h * qd)) / h ** 2
        return x, xd, J, c

    first = T.TaskmapByFunction(pose, pose_diff)
    for second, fn in ((T.TaskmapFrom4x4ToEuler(), lambda x: T.TaskmapFrom4x4ToEuler().forward(x)),
                       (T.TaskmapFrom4x4ToPosition(), lambda x: T.TaskmapFrom4x4ToPosition().forward(x))):
        ch = T.chain_taskmaps([first, second])
        q = rng.uniform(-0.6, 0.6, (1, 5))
        qd = rng.uniform(-0.3, 0.3, (1, 5))
        x, xd, J, c = ch.differentiate(q, qd)

        def f64(qq):
            Tt = pose(qq).reshape(-1, 4, 4)
            if isinstance(second, T.TaskmapFrom4x4ToPosition):
                return Tt[:, :3, 3]
            return np.stack((np.arctan2(Tt[:, 2, 1], Tt[:, 2, 2]), -np.arcsin(Tt[:, 2, 0]), np.arctan2(Tt[:, 1, 0], Tt[:, 0, 0])), -1)
        h = 1e-4
        assert np.abs(ch.forward(q) - f64(q)).max() < 1e-6
        assert np.abs(x - f64(q)).max() < 1e-6
        assert np.abs(xd - (f64(q + h * qd) - f64(q - h * qd)) / (2 * h)).max() < 1e-5
        Jfd = np.stack([(f64(q + h * np.eye(5)[i]) - f64(q - h * np.eye(5)[i])) / (2 * h) for i in range(5)], axis=-1)
        assert np.abs(J - Jfd).max() < 1e-5
        assert np.abs(c - (f64(q + h * qd) - 2 * f64(q) + f64(q - h * qd)) / h ** 2).max() < 2e-4


@pytest.mark.gpu
def test_reference_taskmap_test_body_against_the_oracle(hip_lib):
    """tests/test_taskmaps.py:31-76 of the reference: per joint frame a hand-made FK map, chained to position and to Euler
    angles; 50 joint vectors inside the limits, qd = 0; J_trans to 1e-6, J_rot to 1e-3, R(euler) to 1e-4 -- against the C
    oracle's fp64 kinematics.  Both routes: the recognised one (GPU kernels) and an opaque pair of closures (the reference's
    chain rule on the host over what the closures return)."""
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, taskmap as T
    from scipy.spatial.transform import Rotation
    fk = _fk()
    _, desc = Cf.config2()
    rng = np.random.default_rng(20)
    q = rng.uniform(Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH, (50, 9)).astype(np.float32)
    qd = np.zeros_like(q)
    for i, joint_name in enumerate(fk.frame_names):
        made = T.TaskmapByFunction(forward_fn=lambda q: fk.forward(q, frame=joint_name),
                                   differentiate_fn=lambda q, qd: fk.differentiate(q, qd, frame=joint_name))
        opaque = T.TaskmapByFunction(forward_fn=lambda q: np.array(fk.forward(q, frame=joint_name)),
                                     differentiate_fn=lambda q, qd: tuple(np.array(a) for a in fk.differentiate(q, qd, frame=joint_name)))
        assert isinstance(made.stages()[0], T.TaskmapByForwardKinematic) and opaque.stages() == [opaque]
        x64, _, J64, _ = O.differentiate(desc, q, qd, i, precision="f64")
        xe64, _, Je64, _ = O.differentiate_euler(desc, q, qd, i, precision="f64")
        gimbal = np.abs(np.cos(xe64[:, 1])) < 2e-2          # (the reference draws the same states and meets no gimbal pose either)
        for tm in (made, opaque):
            to_pos = T.chain_taskmaps(tm, T.TaskmapFrom4x4ToPosition())
            to_eul = T.chain_taskmaps(tm, T.TaskmapFrom4x4ToEuler())
            _, _, J_trans, _ = to_pos.differentiate(q, qd)
            eul, _, J_rot, _ = to_eul.differentiate(q, qd)
            assert np.abs(np.asarray(J_trans) - J64[:, [3, 7, 11], :]).max() < 1e-6, joint_name
            assert np.abs(np.asarray(J_rot) - Je64)[~gimbal].max() < 1e-3, joint_name
            R_got = Rotation.from_euler("xyz", np.asarray(eul, np.float64)).as_matrix()
            assert np.abs(R_got - x64.reshape(-1, 4, 4)[:, :3, :3]).max() < 1e-4, joint_name
            assert np.abs(np.asarray(to_pos.forward(q)).reshape(-1, 3) - x64[:, [3, 7, 11]]).max() < 1e-6


@pytest.mark.gpu
def test_rmp_core_compiles_a_leaf_on_a_hand_made_fk_map(hip_lib):
    """A TargetAttractor whose task map is chain_taskmaps([TaskmapByFunction(closures over fkine), 4x4 -> position]) is the
    config-2 set: same q-double-dot as with TaskmapByForwardKinematic, and both within 1e-5 of the oracle."""
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, rmp, rmp2, taskmap as T
    fk = _fk()
    frame = "panda_grasptarget_hand"
    s = Cf.sample_panda_states(np.random.default_rng(4), 64)
    outs = []
    for first in (T.TaskmapByForwardKinematic(fk, frame),
                  T.TaskmapByFunction(lambda q: fk.forward(q, frame=frame), lambda q, qd: fk.differentiate(q, qd, frame=frame))):
        core = rmp.RmpCore(rmps={}, solve="pinv")
        core.add_rmp(rmp2.TargetAttractor(s["goal"], *Cf.TARGET_ATTRACTOR_PARAMS,
                                          taskmap=T.chain_taskmaps([first, T.TaskmapFrom4x4ToPosition()]), name="attractor"))
        core.add_rmp(rmp.JointLimitAvoidance(Cf.PANDA_Q_LOW, Cf.PANDA_Q_HIGH, *Cf.JOINT_LIMIT_PARAMS))
        core.add_rmp(rmp2.JointDamping(*Cf.JOINT_DAMPING_PARAMS))
        outs.append(np.asarray(core.evaluate(s["q"], s["qd"])))
    assert np.array_equal(outs[0], outs[1])
    ref = O.step(Cf.config2("pinv")[1], s["q"], s["qd"], s["goal"])["qdd64"]
    assert np.abs(outs[1] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    opaque = T.TaskmapByFunction(lambda q: np.array(fk.forward(q, frame=frame)), lambda q, qd: fk.differentiate(q, qd, frame=frame))
    core = rmp.RmpCore(rmps={})
    core.add_rmp(rmp2.TargetAttractor(s["goal"], *Cf.TARGET_ATTRACTOR_PARAMS,
                                      taskmap=T.chain_taskmaps([opaque, T.TaskmapFrom4x4ToPosition()]), name="attractor"))
    with pytest.raises(NotImplementedError, match="has no kernel"):
        core.evaluate(s["q"], s["qd"])
