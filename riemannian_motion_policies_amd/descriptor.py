"""ctypes mirror of include/rmp2.h and the RMP-set -> `rmp2_desc` compiler.

The descriptor is the flat, immutable "program" an engine is created from: the robot's
kinematic table (urdf.py) plus one `rmp2_leaf` record per leaf policy.  It replaces the
reference's Python-object graph (RmpCore.rmps dict of policy objects holding task-map
objects holding a UrdfForwardKinematic, rmp.py:111-131).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Sequence

import numpy as np

from .urdf import KinematicTable

ABI_VERSION = 4
MAX_FRAMES = 32
MAX_DOF = 16
MAX_LEAVES = 48
MAX_PARAMS = 12

# leaf kinds / task maps / solve modes / obstacle modes -- keep in sync with include/rmp2.h
LEAF_TARGET_ATTRACTOR = 1
LEAF_JOINT_VELOCITY_CAP = 2
LEAF_JOINT_DAMPING = 3
LEAF_OBSTACLE_AVOIDANCE = 4
LEAF_CSPACE_BIASING = 5
LEAF_TARGET_POLICY = 6
LEAF_JOINT_LIMIT_AVOIDANCE = 7
LEAF_CONFIG_SPACE_BIASING = 8
LEAF_COLLISION_AVOIDANCE = 9

TASKMAP_IDENTITY = 0
TASKMAP_FK_POSITION = 1
TASKMAP_FK_DISTANCE = 2
TASKMAP_FK_POINT = 3

SOLVE_AUTO = 0
SOLVE_PINV = 1
SOLVE_MODES = {"auto": SOLVE_AUTO, "pinv": SOLVE_PINV}

OBS_NONE = 0
OBS_EXPLICIT_PAIRS = 1
OBS_SHARED_SPHERES = 2
OBS_RAGGED_SPHERES = 3

PRIM_SPHERE = 0
PRIM_CAPSULE = 1
PRIM_CYLINDER = 2   # (cx, cy, cz, radius, ux, uy, uz, half_height): finite cylinder, flat caps (simulation.py:245-261)

STATUS_NONFINITE = 1
STATUS_RANK_DROP = 2
STATUS_PINV_PATH = 4
STATUS_JACOBI = 8   # PINV mode, certifying step: full rank not certified, resolved by the Jacobi pseudo-inverse (diagnostic)


class Robot(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32),
        ("n_dof", C.c_int32),
        ("parent", C.c_int32 * MAX_FRAMES),
        ("joint_type", C.c_int32 * MAX_FRAMES),
        ("q_index", C.c_int32 * MAX_FRAMES),
        ("axis", (C.c_float * 3) * MAX_FRAMES),
        ("T_const", (C.c_float * 12) * MAX_FRAMES),
    ]


class Leaf(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("taskmap", C.c_int32),
        ("frame", C.c_int32),
        ("goal_offset", C.c_int32),
        ("params", C.c_float * MAX_PARAMS),
        ("vec_a", C.c_float * MAX_DOF),
        ("vec_b", C.c_float * MAX_DOF),
    ]


class Desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("solve_mode", C.c_int32),
        ("n_leaves", C.c_int32),
        ("goal_floats", C.c_int32),
        ("robot", Robot),
        ("leaves", Leaf * MAX_LEAVES),
    ]


class Obstacles(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("n_spheres", C.c_int32),
        ("n_pairs", C.c_int32),
        ("primitive", C.c_int32),
        ("pair_begin", C.c_int32 * (MAX_LEAVES + 1)),
        ("spheres", C.c_void_p),
        ("p_link", C.c_void_p),
        ("p_obs", C.c_void_p),
        ("csr_offset", C.c_void_p),
        ("csr_index", C.c_void_p),
        ("dist", C.c_void_p),
        ("link_capsules", C.c_void_p),
    ]


class Outputs(C.Structure):
    _fields_ = [
        ("qdd", C.c_void_p),
        ("status", C.c_void_p),
        ("M", C.c_void_p),
        ("f", C.c_void_p),
    ]


class RolloutCfg(C.Structure):
    _fields_ = [("n_control_steps", C.c_int32), ("substeps", C.c_int32), ("dt", C.c_float), ("table_steps", C.c_int32)]


class LeafSpec:
    """Plain-data description of one leaf (what the policy classes serialise to)."""

    def __init__(self, kind: int, taskmap: int, frame: int = -1, params: Sequence[float] = (),
                 vec_a: Sequence[float] | None = None, vec_b: Sequence[float] | None = None,
                 goal_len: int = 0, name: str = ""):
        self.kind, self.taskmap, self.frame = int(kind), int(taskmap), int(frame)
        self.params = [float(p) for p in params]
        self.vec_a = None if vec_a is None else [float(v) for v in np.asarray(vec_a, dtype=np.float64).ravel()]
        self.vec_b = None if vec_b is None else [float(v) for v in np.asarray(vec_b, dtype=np.float64).ravel()]
        self.goal_len = int(goal_len)
        self.name = name

    def signature(self):
        return (self.kind, self.taskmap, self.frame, tuple(self.params), tuple(self.vec_a or ()),
                tuple(self.vec_b or ()), self.goal_len)


def fill_robot(rb: Robot, table: KinematicTable) -> None:
    F, n = table.n_frames, table.n_dof
    if F > MAX_FRAMES:
        raise ValueError(f"{F} frames > RMP2_MAX_FRAMES={MAX_FRAMES}")
    if n > MAX_DOF:
        raise ValueError(f"{n} dof > RMP2_MAX_DOF={MAX_DOF}")
    rb.n_frames, rb.n_dof = F, n
    for i in range(F):
        rb.parent[i] = int(table.parent[i])
        rb.joint_type[i] = int(table.joint_type[i])
        rb.q_index[i] = int(table.q_index[i])
        for k in range(3):
            rb.axis[i][k] = float(table.axis[i, k])
        flat = table.T_const[i, :3, :].reshape(12)
        for k in range(12):
            rb.T_const[i][k] = float(flat[k])


def build_desc(table: KinematicTable, leaves: Iterable[LeafSpec], solve: str | int = "auto") -> Desc:
    """Compile (robot table, leaf list) into the C descriptor.  Leaf order is kept: it is the
    reference's dict insertion order (rmp.py:127-128,142) and fixes the fp64 summation order."""
    leaves = list(leaves)
    if len(leaves) > MAX_LEAVES:
        raise ValueError(f"{len(leaves)} leaves > RMP2_MAX_LEAVES={MAX_LEAVES}")
    d = Desc()
    d.abi_version = ABI_VERSION
    d.solve_mode = SOLVE_MODES[solve] if isinstance(solve, str) else int(solve)
    d.n_leaves = len(leaves)
    fill_robot(d.robot, table)
    n = table.n_dof
    goal_off = 0
    for i, lf in enumerate(leaves):
        if len(lf.params) > MAX_PARAMS:
            raise ValueError(f"leaf {lf.name!r}: too many params")
        if lf.taskmap != TASKMAP_IDENTITY and not (0 <= lf.frame < table.n_frames):
            raise ValueError(f"leaf {lf.name!r}: frame index {lf.frame} out of range")
        rec = d.leaves[i]
        rec.kind, rec.taskmap, rec.frame = lf.kind, lf.taskmap, lf.frame if lf.taskmap != TASKMAP_IDENTITY else -1
        for k, p in enumerate(lf.params):
            rec.params[k] = p
        for name, vec in (("vec_a", lf.vec_a), ("vec_b", lf.vec_b)):
            if vec is not None:
                if len(vec) != n:
                    raise ValueError(f"leaf {lf.name!r}: {name} has {len(vec)} entries, robot has {n} dof")
                arr = getattr(rec, name)
                for k, v in enumerate(vec):
                    arr[k] = v
        if lf.goal_len:
            rec.goal_offset = goal_off
            goal_off += lf.goal_len
        else:
            rec.goal_offset = -1
    d.goal_floats = goal_off
    return d


def distance_leaf_indices(desc: Desc) -> List[int]:
    """Leaves that consume per-pair obstacle data (pair_begin ranges): distance and attached-point maps."""
    return [i for i in range(desc.n_leaves) if desc.leaves[i].taskmap in (TASKMAP_FK_DISTANCE, TASKMAP_FK_POINT)]
