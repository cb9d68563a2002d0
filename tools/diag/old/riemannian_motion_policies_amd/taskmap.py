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
    """taskmap.py:57-67 -- used only by the reference's tests; not on the control path."""

    def forward(self, input):
        raise NotImplementedError("Euler task map is outside the accelerated path (SURVEY 8(f)-4)")


class TaskmapFrom4x4ToQuaternions(Taskmap):
    def forward(self, input):
        raise NotImplementedError  # NotImplemented in the reference as well (taskmap.py:70-72)


class TaskmapRelative4x4(Taskmap):
    """taskmap.py:79-99 -- TwoJoint experiment 05 only; SURVEY 8(f)-4 ("next" row)."""

    def __init__(self, relative_pos):
        self.relative_pos = relative_pos

    def forward(self, input):
        raise NotImplementedError("TaskmapRelative4x4 is outside the accelerated path (SURVEY 8(f)-4)")


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
        kind, fk = classify(self)[:2]
        if kind == D.TASKMAP_FK_POSITION:
            x, xd, J, c = fk.differentiate(q, qd)
            rows = list(TaskmapFrom4x4ToPosition.ROWS)
            return x[:, rows], xd[:, rows], J[:, rows, :], c[:, rows]
        raise NotImplementedError("differentiate() is offered for FK and FK->position chains; "
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
    names = " -> ".join(type(s).__name__ for s in st)
    raise NotImplementedError(
        f"task-map chain [{names}] has no kernel; supported: IdentityTaskmap, "
        "[FK, 4x4->position], [FK, 4x4->distance] (SURVEY 8(b))")
