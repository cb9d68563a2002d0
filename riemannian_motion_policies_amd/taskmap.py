"""Task-map descriptors with the reference's names (taskmap.py:13-168).

In the reference every task map is a differentiable TensorFlow function and RmpCore
re-differentiates it with nested GradientTapes for every RMP, every step.  Here a task map
is a *descriptor*: `chain_taskmaps([...])` records the stages and the RMP-set compiler
(`rmp.RmpCore`) pattern-matches the supported chains onto the kernels' built-in task maps

    IdentityTaskmap                                   -> RMP2_TASKMAP_IDENTITY
    [FK(frame), TaskmapFrom4x4ToPosition]             -> RMP2_TASKMAP_FK_POSITION
    [FK(frame), TaskmapJointFrame4x4ToDistance(...)]  -> RMP2_TASKMAP_FK_DISTANCE

whose (x, xd, J, c) the HIP kernels compute analytically in one pass over the kinematic tree.
`forward` / `differentiate` stay callable for the FK-only, FK->position and FK->Euler chains and run on
the GPU (rmp2_forward_kinematics / rmp2_differentiate / rmp2_differentiate_euler); there is no CPU path for
those.  `TaskmapByFunction(forward_fn, differentiate_fn)` has the reference's signature (taskmap.py:33-42): a map
made of two closures chains with the others through the reference's chain rule (host arithmetic on whatever the
closures return), and one whose closures are exactly `fkine.forward` / `fkine.differentiate` of a frame compiles
like TaskmapByForwardKinematic.
"""
from __future__ import annotations

import numpy as np

from . import descriptor as D


class Taskmap:
    def forward(self, q):
        raise NotImplementedError

    def differentiate(self, q, qd):
        raise NotImplementedError

    # descriptor protocol ------------------------------------------------------------
    def stages(self):
        return [self]


class IdentityTaskmap(Taskmap):
    """x = q, xd = qd, J = I, c = 0   (taskmap.py:13-20)."""

    def forward(self, q):
        return np.asarray(q, dtype=np.float32)

    def differentiate(self, q, qd):
        q = np.atleast_2d(np.asarray(q, dtype=np.float32))
        qd = np.atleast_2d(np.asarray(qd, dtype=np.float32))
        n = q.shape[-1]
        J = np.broadcast_to(np.eye(n, dtype=np.float32), (q.shape[0], n, n)).copy()
        return q, qd, J, np.zeros_like(q)


class TaskmapByForwardKinematic(Taskmap):
    """q -> vec(T_frame) (taskmap.py:22-31); delegates to UrdfForwardKinematic (GPU)."""

    def __init__(self, fkine, frame):
        self.fkine = fkine
        self.frame = frame if isinstance(frame, str) else _to_str(frame)

    def forward(self, q):
        return self.fkine.forward(q, self.frame)

    def differentiate(self, q, qd):
        return self.fkine.differentiate(q, qd, self.frame)


class TaskmapFrom4x4ToPosition(Taskmap):
    """vec(T) -> T[:3, 3]; constant selector Jacobian, c = 0 (taskmap.py:45-54)."""

    ROWS = (3, 7, 11)

    def forward(self, input):
        T = np.asarray(input, dtype=np.float32).reshape(-1, 4, 4)
        return T[:, :3, 3]

    def differentiate(self, q, qd):
        return _selector_differentiate(self.ROWS, np.reshape(_host(q), (-1, 16)), np.reshape(_host(qd), (-1, 16)))


class TaskmapFrom4x4ToEuler(Taskmap):
    """vec(T) -> (theta_x, theta_y, theta_z) with R = Rz Ry Rx (taskmap.py:57-67, euler_from_rotation_matrix
    kinematics.py:74-96, gimbal guard :88 included).  No leaf of the reference consumes it (its tests do,
    tests/test_taskmaps.py:42-44); chained behind an FK map, differentiate() runs on the GPU
    (rmp2_differentiate_euler)."""

    def forward(self, input):
        T = np.asarray(input, dtype=np.float32).reshape(-1, 4, 4)
        r00, r10, r20, r21, r22 = T[:, 0, 0], T[:, 1, 0], T[:, 2, 0], T[:, 2, 1], T[:, 2, 2]
        ty = -np.arcsin(r20)
        cy = np.cos(ty)
        safe = np.where(np.abs(cy) < 1e-6, np.ones_like(cy), cy)
        tz = np.arctan2(r10 / safe, r00 / safe)
        tx = np.arctan2(r21 / safe, r22 / safe)
        return np.stack((tx, ty, tz), axis=-1).astype(np.float32)

    def differentiate(self, q, qd):
        """(x, xd, J, c) of the 16 -> 3 map itself (what rmp_differentiate gives the reference, taskmap.py:62-67), closed form
        on the host: J2 [B,3,16], c2 = xd^T (Hessian) xd.  Only what a chain behind a HAND-MADE FK map needs (the chain behind
        TaskmapByForwardKinematic runs on the GPU: rmp2_differentiate_euler).  atan2(y / s, x / s) does not depend on s, so the
        gimbal guard's `safe` carries no derivative."""
        x = np.reshape(_host(q), (-1, 16)).astype(np.float64)
        xd = np.reshape(_host(qd), (-1, 16)).astype(np.float64)
        B = x.shape[0]
        J = np.zeros((B, 3, 16))
        c = np.zeros((B, 3))
        r00, r10, r20, r21, r22 = x[:, 0], x[:, 4], x[:, 8], x[:, 9], x[:, 10]

        def atan2_terms(yi, xi):
            y, xx, yd, xxd = x[:, yi], x[:, xi], xd[:, yi], xd[:, xi]
            rho = xx * xx + y * y
            gy, gx = xx / rho, -y / rho
            curv = (2 * xx * y / rho ** 2) * (xxd * xxd - yd * yd) + 2 * xxd * yd * (y * y - xx * xx) / rho ** 2
            return gy, gx, curv

        gy, gx, cx = atan2_terms(9, 10)      # theta_x = atan2(r21, r22)
        J[:, 0, 9], J[:, 0, 10], c[:, 0] = gy, gx, cx
        one = 1.0 - r20 * r20
        J[:, 1, 8] = -1.0 / np.sqrt(one)     # theta_y = -asin(r20)
        c[:, 1] = -r20 / one ** 1.5 * xd[:, 8] ** 2
        gy, gx, cz = atan2_terms(4, 0)       # theta_z = atan2(r10, r00)
        J[:, 2, 4], J[:, 2, 0], c[:, 2] = gy, gx, cz
        out = self.forward(x.astype(np.float32))
        return out, np.einsum("bkm,bm->bk", J, xd).astype(np.float32), J.astype(np.float32), c.astype(np.float32)


class TaskmapFrom4x4ToQuaternions(Taskmap):
    def forward(self, input):
        raise NotImplementedError  # NotImplemented in the reference as well (taskmap.py:70-72)


class TaskmapRelative4x4(Taskmap):
    """vec(T_ref) -> vec(T_ref @ [I | relative_pos_b]) for every pair b (taskmap.py:79-99): a point rigidly
    attached to the reference frame.  Used in the chain [FK(frame), TaskmapRelative4x4, 4x4->position] by the
    CollisionAvoidance leaves of experiments/two_joint_robot/05_obstacle_avoidance.py:51-61.

    `relative_pos` is an array holder ([B,3] for one robot, [R,B,3] for a fleet; joint-frame coordinates) read
    at every RmpCore.evaluate, like the reference's tf.Variable (data_management.py:16)."""

    def __init__(self, relative_pos):
        self.relative_pos = relative_pos

    def forward(self, input):
        from .data_management import as_array
        rel = np.asarray(as_array(self.relative_pos), dtype=np.float32).reshape(-1, 3)
        T = np.asarray(input, dtype=np.float32).reshape(-1, 4, 4)
        T = np.broadcast_to(T, (rel.shape[0], 4, 4))
        T_rel = np.broadcast_to(np.eye(4, dtype=np.float32), (rel.shape[0], 4, 4)).copy()
        T_rel[:, :3, 3] = rel
        return (T @ T_rel).reshape(-1, 16)


class TaskmapJointFrame4x4ToDistance(Taskmap):
    """vec(T_frame) -> |p_link - p_obs| per closest-point pair (taskmap.py:115-138).

    `pos_on_link_in_base_frame` / `pos_on_obstacle_in_base_frame` are array holders
    (data_management.ArrayVar, numpy arrays or torch tensors) of shape [B,3] (one robot) or
    [R,B,3] (fleet); they are read at every RmpCore.evaluate, like the reference's
    tf.Variables.  Quirk Q5 is kept: the derivative treats the control point as translating
    with the frame ORIGIN.
    """

    def __init__(self, pos_on_link_in_base_frame, pos_on_obstacle_in_base_frame):
        self.pos_on_link_in_base_frame = pos_on_link_in_base_frame
        self.pos_on_obstacle_in_base_frame = pos_on_obstacle_in_base_frame


class TaskmapSphereDistance(Taskmap):
    """FK(frame) -> distance of the frame origin to every sphere of a shared table.

    New (no reference counterpart): the array-backed "interface A" of SURVEY 8(b): the pairs
    are formed inside the kernel from `spheres[K,4]` handed to RmpCore.evaluate, equivalent to
    TaskmapJointFrame4x4ToDistance with p_link = frame origin and p_obs = nearest surface point.
    """


class TaskmapByFunction(Taskmap):
    """A task map given by two closures, with the reference's constructor (taskmap.py:33-42;
    tests/test_taskmaps.py:33-36 builds one over `fkine.forward` / `fkine.differentiate`):

        TaskmapByFunction(forward_fn=lambda q: fkine.forward(q, frame=name),
                          differentiate_fn=lambda q, qd: fkine.differentiate(q, qd, frame=name))

    `.forward` / `.differentiate` call the closures, and the object chains with the other maps through the reference's chain
    rule (`chain_taskmaps`: J = J2 J1, xd = J2 xd1, c = c2 + J2 c1, taskmap.py:150-160).  `chain_taskmaps` itself returns
    one of these, as in the reference; the ones it builds also carry the list of stages they were made of, which is what the
    RMP-set compiler reads.  A hand-made one has no stage list: `stages()` then finds out what the closures do by CALLING
    them once with a recording stand-in for the kinematics' methods -- a pair of closures that hands `q` (and `qd`) to ONE
    `UrdfForwardKinematic`'s forward / differentiate with the same frame and returns the result untouched is the map
    TaskmapByForwardKinematic(fkine, frame) and compiles to the kernels' built-in FK map; anything else has no kernel and
    RmpCore says which spellings have."""

    def __init__(self, forward_fn, differentiate_fn):
        self.forward_fn = forward_fn
        self.differentiate_fn = differentiate_fn
        self._stages = None

    def forward(self, q):
        return self.forward_fn(q)

    def differentiate(self, q, qd):
        return self.differentiate_fn(q, qd)

    def stages(self):
        if self._stages is not None:
            return self._stages
        fk = _recognise_fk_closures(self.forward_fn, self.differentiate_fn)
        if fk is None:
            return [self]          # opaque: classify() refuses it with the supported spellings
        last = getattr(self, "_recognised", None)
        if last is not None and last.fkine is fk.fkine and last.frame == fk.frame:
            return [last]          # (the same stage object as long as the closures mean the same map: RmpCore's quick signature)
        self._recognised = fk
        return [fk]


def _closure_values(fn):
    import inspect
    try:
        cv = inspect.getclosurevars(fn)
    except TypeError:
        return []
    return list(cv.nonlocals.values()) + list(cv.globals.values())


def _recognise_fk_closures(forward_fn, differentiate_fn):
    """TaskmapByForwardKinematic(fkine, frame) if the two closures are exactly `fkine.forward(q, frame)` and
    `fkine.differentiate(q, qd, frame)` of one UrdfForwardKinematic -- decided by running them once against recording
    stand-ins (instance attributes shadowing the methods for the duration of the call): each must call its method exactly once,
    with the probe arrays it was given, and return the stand-in's result object itself.  The frame is read at this moment (a
    closure over a loop variable sees the variable's current value -- the reference would read it at every evaluate)."""
    from .kinematics import UrdfForwardKinematic
    cands = []
    for v in _closure_values(forward_fn) + _closure_values(differentiate_fn):
        if isinstance(v, UrdfForwardKinematic) and all(v is not c for c in cands):
            cands.append(v)
    for fk in cands:
        n = fk.n_joints
        q_probe, qd_probe = np.zeros((1, n), np.float32), np.zeros((1, n), np.float32)
        calls = []
        token_f, token_d = np.zeros((1, 4, 4), np.float32), (object(), object(), object(), object())

        def rec_forward(q, frame=None, _c=calls, **kw):
            _c.append(("forward", q, frame if frame is not None else kw.get("frame")))
            return token_f

        def rec_diff(q, qd, frame=None, _c=calls, **kw):
            _c.append(("differentiate", q, qd, frame if frame is not None else kw.get("frame")))
            return token_d

        saved = {k: fk.__dict__.get(k) for k in ("forward", "differentiate", "__call__")}
        fk.forward, fk.differentiate = rec_forward, rec_diff
        try:
            try:
                out_f = forward_fn(q_probe)
                out_d = differentiate_fn(q_probe, qd_probe)
            except Exception:
                continue
        finally:
            for k, v in saved.items():
                if v is None:
                    fk.__dict__.pop(k, None)
                else:
                    fk.__dict__[k] = v
        if (len(calls) == 2 and calls[0][0] == "forward" and calls[1][0] == "differentiate" and out_f is token_f
                and out_d is token_d and calls[0][1] is q_probe and calls[1][1] is q_probe and calls[1][2] is qd_probe
                and calls[0][2] is not None and _to_str(calls[0][2]) == _to_str(calls[1][3])
                and _to_str(calls[0][2]) in fk.frame_names):
            return TaskmapByForwardKinematic(fk, calls[0][2])
    return None


def _selector_differentiate(rows, x, xd):
    """A constant row selector as a task map of its own (what rmp_differentiate gives the reference for
    TaskmapFrom4x4ToPosition, taskmap.py:45-54): x[rows], xd[rows], J = the 0/1 selector, c = 0."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float32))
    xd = np.atleast_2d(np.asarray(xd, dtype=np.float32))
    J = np.zeros((x.shape[0], len(rows), x.shape[1]), np.float32)
    J[:, np.arange(len(rows)), list(rows)] = 1.0
    return x[:, list(rows)], xd[:, list(rows)], J, np.zeros((x.shape[0], len(rows)), np.float32)


def _host(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def _chain_taskmaps(taskmap_1, taskmap_2):
    """taskmap.py:142-162.  Known chains keep their kernels (FK -> Euler: rmp2_differentiate_euler; FK -> position: rows of
    rmp2_differentiate); every other pair goes through the reference's chain rule on the two maps' own differentiate()."""
    st = taskmap_1.stages() + taskmap_2.stages()

    def combined_forward(q):
        return taskmap_2.forward(taskmap_1.forward(q))

    def combined_differentiate(q, qd):
        if len(st) == 2 and isinstance(st[0], TaskmapByForwardKinematic) and isinstance(st[1], TaskmapFrom4x4ToEuler):
            return st[0].fkine.differentiate_euler(q, qd, st[0].frame)
        if len(st) == 2 and isinstance(st[0], TaskmapByForwardKinematic) and isinstance(st[1], TaskmapFrom4x4ToPosition):
            x, xd, J, c = st[0].fkine.differentiate(q, qd, st[0].frame)
            rows = list(TaskmapFrom4x4ToPosition.ROWS)
            return x[:, rows], xd[:, rows], J[:, rows, :], c[:, rows]
        out_1, dout1_dt, J_1, c_1 = (_host(a) for a in taskmap_1.differentiate(q, qd))
        out_2, _, J_2, c_2 = (_host(a) for a in taskmap_2.differentiate(out_1, dout1_dt))
        dout_dt = np.einsum("bkm,bm->bk", J_2, np.broadcast_to(dout1_dt, (J_2.shape[0], dout1_dt.shape[-1])))
        J = J_2 @ J_1
        c = c_2 + np.einsum("bkm,bm->bk", J_2, np.broadcast_to(c_1, (J_2.shape[0], c_1.shape[-1])))
        return out_2, dout_dt, J, c

    chained = TaskmapByFunction(combined_forward, combined_differentiate)
    chained._stages = st
    return chained


def chain_taskmaps(taskmap_list, *more):
    """taskmap.py:164-168 (one list argument); the reference's own test still calls it with the maps as positional arguments
    (tests/test_taskmaps.py:39-40), so that form is taken too."""
    if more or not isinstance(taskmap_list, (list, tuple)):
        taskmap_list = [taskmap_list, *more]
    chained = taskmap_list[0]
    for tm in taskmap_list[1:]:
        chained = _chain_taskmaps(chained, tm)
    return chained


def _to_str(frame):
    if isinstance(frame, bytes):
        return frame.decode("ascii")
    if hasattr(frame, "numpy"):  # tf.constant(frame, dtype=tf.string)
        v = frame.numpy()
        return v.decode("ascii") if isinstance(v, bytes) else str(v)
    return str(frame)


def classify(taskmap):
    """-> (RMP2_TASKMAP_*, TaskmapByForwardKinematic | None, last stage)."""
    st = taskmap.stages()
    if len(st) == 1 and isinstance(st[0], IdentityTaskmap):
        return D.TASKMAP_IDENTITY, None, st[0]
    if len(st) == 2 and isinstance(st[0], TaskmapByForwardKinematic):
        if isinstance(st[1], TaskmapFrom4x4ToPosition):
            return D.TASKMAP_FK_POSITION, st[0], st[1]
        if isinstance(st[1], (TaskmapJointFrame4x4ToDistance, TaskmapSphereDistance)):
            return D.TASKMAP_FK_DISTANCE, st[0], st[1]
    if (len(st) == 3 and isinstance(st[0], TaskmapByForwardKinematic) and isinstance(st[1], TaskmapRelative4x4)
            and isinstance(st[2], TaskmapFrom4x4ToPosition)):
        return D.TASKMAP_FK_POINT, st[0], st[1]
    names = " -> ".join(type(s).__name__ for s in st)
    raise NotImplementedError(
        f"task-map chain [{names}] has no kernel; supported: IdentityTaskmap, "
        "[FK, 4x4->position], [FK, 4x4->distance], [FK, relative 4x4, 4x4->position] (SURVEY 8(b)), where FK is "
        "TaskmapByForwardKinematic(fkine, frame) or TaskmapByFunction(forward_fn=lambda q: fkine.forward(q, frame), "
        "differentiate_fn=lambda q, qd: fkine.differentiate(q, qd, frame)) -- closures that do anything else to q or to the "
        "result cannot be compiled (their .forward / .differentiate still work on the host)")
