"""Task-map descriptors with the reference's names (taskmap.py:13-168).

In the reference every task map is a differentiable TensorFlow function and RmpCore
re-differentiates it with nested GradientTapes for every RMP, every step.  Here a task map
is a *descriptor*: `chain_taskmaps([...])` records the stages and the RMP-set compiler
(`rmp.RmpCore`) pattern-matches the supported chains onto the kernels' built-in task maps

    IdentityTaskmap                                   -> RMP2_TASKMAP_IDENTITY
    [FK(frame), TaskmapFrom4x4ToPosition]             -> RMP2_TASKMAP_FK_POSITION
    [FK(frame), TaskmapJointFrame4x4ToDistance(...)]  -> RMP2_TASKMAP_FK_DISTANCE

whose (x, xd, J, c) the HIP kernels compute analytically in one pass over the kinematic tree.
`forward` / `differentiate` stay callable for the FK-only and FK->position chains and run on
the GPU (rmp2_forward_kinematics / rmp2_differentiate); there is no CPU path.
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
    """Result of chain_taskmaps (taskmap.py:33-42): keeps the stage list."""

    def __init__(self, stage_list):
        self._stages = list(stage_list)

    def stages(self):
        return self._stages

    def forward(self, q):
        out = q
        for s in self._stages:
            out = s.forward(out)
        return out

    def differentiate(self, q, qd):
        st = self.stages()
        if len(st) == 2 and isinstance(st[0], TaskmapByForwardKinematic) and isinstance(st[1], TaskmapFrom4x4ToEuler):
            return st[0].fkine.differentiate_euler(q, qd, st[0].frame)
        kind, fk = classify(self)[:2]
        if kind == D.TASKMAP_FK_POSITION:
            x, xd, J, c = fk.differentiate(q, qd)
            rows = list(TaskmapFrom4x4ToPosition.ROWS)
            return x[:, rows], xd[:, rows], J[:, rows, :], c[:, rows]
        raise NotImplementedError("differentiate() is offered for FK, FK->position and FK->Euler chains; "
                                  "distance chains are differentiated inside RmpCore.evaluate")


def _chain_taskmaps(taskmap_1, taskmap_2):
    return TaskmapByFunction(taskmap_1.stages() + taskmap_2.stages())


def chain_taskmaps(taskmap_list):
    """taskmap.py:164-168 (one list argument)."""
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
        "[FK, 4x4->position], [FK, 4x4->distance], [FK, relative 4x4, 4x4->position] (SURVEY 8(b))")
