"""The BASELINE.json workload configurations as data: RMP sets (parameters copied from the
reference's experiment scripts) and the synthetic input distributions of SURVEY section 8(d).

Used by bench.py, the parity tests and the golden-fixture generator so that all three see the
same sets and the same seeded inputs.  Nothing here computes on the hot path.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import descriptor as D
from .urdf import KinematicTable, panda_table, two_joint_table

# simulation.py:137-139 (FrankaPanda) restricted to idx_controllable = [0..6, 9, 10]
PANDA_Q_READY = np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4, 0.02, 0.02], dtype=np.float64)
PANDA_Q_LOW = np.array([-2.9671, -1.8326, -2.9671, -3.1416, -2.9671, -0.0873, -2.9671, 0.0, 0.0])
PANDA_Q_HIGH = np.array([2.9671, 1.8326, 2.9671, 0.0, 2.9671, 3.8223, 2.9671, 0.04, 0.04])
# simulation.py:84-86 (TwoJointRobot)
TWO_JOINT_Q_LOW = np.array([-np.pi, -np.pi])
TWO_JOINT_Q_HIGH = np.array([np.pi, np.pi])

# experiments/franka_panda/06_cluttered_environment.py:70-75
TARGET_ATTRACTOR_PARAMS = [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02]
# :78-80
JOINT_VELOCITY_CAP_PARAMS = [0.5, 0.15, 5.0, 0.05]
# :83-85
JOINT_DAMPING_PARAMS = [1.0, 0.005, 0.3]
# :88-92   (metric_scalar, position_gain, damping_gain, robust_position_term_thresh, inertia)
CSPACE_BIASING_PARAMS = [0.005, 1.0, 2.0, 0.5, 0.0001]
CSPACE_BIASING_GOAL = [0.0, -0.9, 0.0, -2.8, 0.0, 2.0, 0.7853981633974483, 0.02, 0.02]
# :109-114
OBSTACLE_AVOIDANCE_PARAMS = [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]
# experiments/two_joint_robot/03_jointlimit_avoiding.py:36
JOINT_LIMIT_PARAMS = [0.3, 1.0]
# experiments/two_joint_robot/01_target_rmp_only.py:44
TARGET_POLICY_PARAMS = [0.1, 0.5, 0.1]

# experiments/two_joint_robot/05_obstacle_avoidance.py:48 and :57-58
EXP05_TARGET_POLICY_PARAMS = [0.1, 0.1, 0.1]
COLLISION_AVOIDANCE_PARAMS = [0.1 * np.e, 0.3, 1.0, 0.3, 1.1, 1e5]

# 8 of the 10 collision frames, in frame order
CONTROL_POINT_FRAMES = ["panda_joint2", "panda_joint3", "panda_joint4", "panda_joint5", "panda_joint7",
                        "panda_hand_joint", "panda_finger_joint1", "panda_finger_joint2"]
TWO_JOINT_CONTROL_POINT_FRAMES = ["joint_1", "joint_2", "link_23"]
N_SPHERES = 32


def config1(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """TwoJoint, TargetPolicy on FK(link_23)->pos   (BASELINE config 1)."""
    t = two_joint_table()
    specs = [D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, t.frame_index("link_23"),
                        TARGET_POLICY_PARAMS, goal_len=3, name="target")]
    return t, D.build_desc(t, specs, solve)


def config2(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """Panda: TargetAttractor + JointLimitAvoidance + JointDamping   (BASELINE config 2)."""
    t = panda_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"),
                   TARGET_ATTRACTOR_PARAMS, goal_len=3, name="attractor"),
        D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, JOINT_LIMIT_PARAMS,
                   vec_a=PANDA_Q_LOW, vec_b=PANDA_Q_HIGH, name="joint_limit_avoidance"),
        D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, JOINT_DAMPING_PARAMS, name="joint_damping"),
    ]
    return t, D.build_desc(t, specs, solve)


def config3(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """Panda cluttered environment: the experiment-06 set with ObstacleAvoidance on 8
    control-point frames (BASELINE configs 3/4).  The same descriptor serves the shared-sphere
    and the explicit-pair obstacle interfaces (the mode is a per-step input)."""
    t = panda_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"),
                   TARGET_ATTRACTOR_PARAMS, goal_len=3, name="attractor"),
        D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, JOINT_VELOCITY_CAP_PARAMS,
                   name="joint_velocity_cap"),
        D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, JOINT_DAMPING_PARAMS, name="joint_damping"),
        D.LeafSpec(D.LEAF_CSPACE_BIASING, D.TASKMAP_IDENTITY, -1, CSPACE_BIASING_PARAMS,
                   vec_a=CSPACE_BIASING_GOAL, name="cspace_target"),
    ]
    for fr in CONTROL_POINT_FRAMES:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, t.frame_index(fr),
                                OBSTACLE_AVOIDANCE_PARAMS, name=f"collision_avoidance_for_{fr}"))
    return t, D.build_desc(t, specs, solve)


def config5_two_joint(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """TwoJoint half of the mixed fleet: config-1 set + obstacle leaves on its 3 frames."""
    t = two_joint_table()
    specs = [D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, t.frame_index("link_23"),
                        TARGET_POLICY_PARAMS, goal_len=3, name="target")]
    for fr in TWO_JOINT_CONTROL_POINT_FRAMES:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, t.frame_index(fr),
                                OBSTACLE_AVOIDANCE_PARAMS, name=f"collision_avoidance_for_{fr}"))
    return t, D.build_desc(t, specs, solve)


def exp05_two_joint(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """experiments/two_joint_robot/05_obstacle_avoidance.py:44-61: TargetPolicy on FK(link_23)->pos plus one
    CollisionAvoidance per frame on the chain [FK(frame), TaskmapRelative4x4, 4x4->pos]."""
    t = two_joint_table()
    specs = [D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, t.frame_index("link_23"),
                        EXP05_TARGET_POLICY_PARAMS, goal_len=3, name="target")]
    for fr in t.frame_names:
        specs.append(D.LeafSpec(D.LEAF_COLLISION_AVOIDANCE, D.TASKMAP_FK_POINT, t.frame_index(fr),
                                COLLISION_AVOIDANCE_PARAMS, name=f"collision_avoidance_for_{fr}"))
    return t, D.build_desc(t, specs, solve)


def exp05_panda(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """The exp-05 leaf on the 9-dof arm (not a reference script; exercises the attached-point map at n = 9):
    TargetAttractor + JointDamping + CollisionAvoidance on the 8 control-point frames."""
    t = panda_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"),
                   TARGET_ATTRACTOR_PARAMS, goal_len=3, name="attractor"),
        D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, JOINT_DAMPING_PARAMS, name="joint_damping"),
    ]
    for fr in CONTROL_POINT_FRAMES:
        specs.append(D.LeafSpec(D.LEAF_COLLISION_AVOIDANCE, D.TASKMAP_FK_POINT, t.frame_index(fr),
                                COLLISION_AVOIDANCE_PARAMS, name=f"collision_avoidance_for_{fr}"))
    return t, D.build_desc(t, specs, solve)


# experiments/two_joint_robot/04_driving_into_jointlimits.py:48-51
EXP04_TARGET_POLICY_PARAMS = [0.1, 1.0, 0.1]
EXP04_JOINT_LIMIT_PARAMS = [0.2, 1.0]
# experiments/franka_panda/04_nullspace_control.py:46-52
PANDA04_TARGET_POLICY_PARAMS = [0.1, 1.0, 0.1]
PANDA04_CONFIG_SPACE_BIASING_PARAMS = [0.01, 0.1, 0.05]
PANDA04_Q0 = [np.pi / 2, -0.05, 0, -2.01, 0, 2.22, 0.79, 0.02, 0.02]


def exp04_two_joint(solve="auto", with_damping=False) -> Tuple[KinematicTable, D.Desc]:
    """experiments/two_joint_robot/04_driving_into_jointlimits.py:46-52: TargetPolicy on the IDENTITY map (goal is a
    joint vector) + JointLimitAvoidance.  with_damping appends a JointDamping leaf (not in the script): the set then
    carries an inertia term, so that the elimination resolve of every mapping is exercised, not only the pseudo-inverse."""
    t = two_joint_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_IDENTITY, -1, EXP04_TARGET_POLICY_PARAMS, goal_len=2,
                   name="rotate_joint1"),
        D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, EXP04_JOINT_LIMIT_PARAMS,
                   vec_a=TWO_JOINT_Q_LOW, vec_b=TWO_JOINT_Q_HIGH, name="joint_limit_avoidance"),
    ]
    if with_damping:
        specs.append(D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, JOINT_DAMPING_PARAMS, name="joint_damping"))
    return t, D.build_desc(t, specs, solve)


def exp04_panda_identity_target(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """The exp-04 leaf pair on the 9-dof arm (not a reference script; the identity-map TargetPolicy at n = 9):
    TargetPolicy(identity, goal = joint vector) + JointLimitAvoidance + JointDamping."""
    t = panda_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_IDENTITY, -1, EXP04_TARGET_POLICY_PARAMS, goal_len=9,
                   name="joint_target"),
        D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, EXP04_JOINT_LIMIT_PARAMS,
                   vec_a=PANDA_Q_LOW, vec_b=PANDA_Q_HIGH, name="joint_limit_avoidance"),
        D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, JOINT_DAMPING_PARAMS, name="joint_damping"),
    ]
    return t, D.build_desc(t, specs, solve)


def panda04_nullspace(solve="auto") -> Tuple[KinematicTable, D.Desc]:
    """experiments/franka_panda/04_nullspace_control.py:41-52: TargetPolicy on FK(panda_grasptarget_hand)->pos +
    ConfigurationSpaceBiasing (the older leaves of rmp.py on the Panda)."""
    t = panda_table()
    specs = [
        D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"),
                   PANDA04_TARGET_POLICY_PARAMS, goal_len=3, name="target"),
        D.LeafSpec(D.LEAF_CONFIG_SPACE_BIASING, D.TASKMAP_IDENTITY, -1, PANDA04_CONFIG_SPACE_BIASING_PARAMS,
                   vec_a=PANDA04_Q0, name="jointspace_biasing"),
    ]
    return t, D.build_desc(t, specs, solve)


def sample_point_pairs(rng: np.random.Generator, R: int, n_leaves: int, B: int):
    """Datamanager fields of the attached-point leaves (data_management.py:14-16) as arrays:
    relative_position [R, n_leaves*B, 3] (joint frame), normal_vec [R, n_leaves*B, 3] (unit), distance
    [R, n_leaves*B] (some beyond the metric radius r = 1.1, where the spline weight is cut to 0)."""
    P = n_leaves * B
    rel = rng.uniform(-0.15, 0.15, size=(R, P, 3))
    nv = rng.normal(size=(R, P, 3))
    nv /= np.linalg.norm(nv, axis=-1, keepdims=True)
    d = rng.uniform(0.05, 1.3, size=(R, P))
    return rel.astype(np.float32), nv.astype(np.float32), d.astype(np.float32)


# ---- synthetic inputs (SURVEY 8(d) "Value distributions / seeds") -------------------------

def sample_panda_states(rng: np.random.Generator, R: int) -> Dict[str, np.ndarray]:
    q = np.empty((R, 9))
    q[:, :7] = PANDA_Q_READY[:7] + rng.uniform(-0.5, 0.5, size=(R, 7))
    q[:, 7:] = rng.uniform(0.0, 0.04, size=(R, 2))
    q = np.clip(q, PANDA_Q_LOW, PANDA_Q_HIGH)
    qd = rng.uniform(-0.1, 0.1, size=(R, 9))
    goal = rng.uniform([0.3, -0.7, 0.3], [0.7, 0.7, 0.7], size=(R, 3))
    return {"q": q.astype(np.float32), "qd": qd.astype(np.float32), "goal": goal.astype(np.float32)}


def sample_two_joint_states(rng: np.random.Generator, R: int) -> Dict[str, np.ndarray]:
    q = rng.uniform(-np.pi, np.pi, size=(R, 2))
    qd = rng.uniform(-0.1, 0.1, size=(R, 2))
    goal = np.stack([rng.uniform(0.1, 1.4, R), rng.uniform(-1.4, 1.4, R), np.full(R, 0.1)], axis=1)
    return {"q": q.astype(np.float32), "qd": qd.astype(np.float32), "goal": goal.astype(np.float32)}


def sample_spheres(rng: np.random.Generator, K: int = N_SPHERES) -> np.ndarray:
    """simulation.py:496-498: centre cylindrical r~U(.4,.9), phi~U(0,2pi), z~U(0,1); radius U(.05,.1)."""
    r, phi, z = rng.uniform(0.4, 0.9, K), rng.uniform(0, 2 * np.pi, K), rng.uniform(0, 1, K)
    rad = rng.uniform(0.05, 0.1, K)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z, rad], axis=1).astype(np.float32)


def sample_capsules(rng: np.random.Generator, K: int = N_SPHERES) -> np.ndarray:
    """The reference's cylinder clutter (simulation.py:495-500: centre cylindrical r~U(.4,.9), phi~U(0,2pi),
    z~U(0,1); rpy~U(0,pi)^3; radius U(.05,.1); height .5) as capsules [K, 8] = (a, radius, b, 0): the cylinder
    axis swept by its radius."""
    r, phi, z = rng.uniform(0.4, 0.9, K), rng.uniform(0, 2 * np.pi, K), rng.uniform(0, 1, K)
    centre = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)
    rpy = rng.uniform(0, np.pi, size=(K, 3))
    cr, sr, cp, sp = np.cos(rpy[:, 0]), np.sin(rpy[:, 0]), np.cos(rpy[:, 1]), np.sin(rpy[:, 1])
    cy, sy = np.cos(rpy[:, 2]), np.sin(rpy[:, 2])
    # third column of Rz(yaw) Ry(pitch) Rx(roll) (PyBullet's getQuaternionFromEuler): the cylinder's local z axis
    axis = np.stack([cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr], axis=1)
    rad = rng.uniform(0.05, 0.1, K)
    half = 0.25
    out = np.zeros((K, 8))
    out[:, 0:3] = centre - half * axis
    out[:, 3] = rad
    out[:, 4:7] = centre + half * axis
    return out.astype(np.float32)


def pairs_from_capsules(origins: np.ndarray, capsules: np.ndarray):
    """Explicit closest-point pairs equivalent to a capsule table, in fp64 numpy (independent of the engine and
    of the C oracle): origins [R, C, 3], capsules [K, 8] -> p_link, p_obs [R, C*K, 3]."""
    R, Cn, _ = origins.shape
    K = capsules.shape[0]
    o = origins.astype(np.float64)[:, :, None, :]
    a = capsules[None, None, :, 0:3].astype(np.float64)
    b = capsules[None, None, :, 4:7].astype(np.float64)
    u = b - a
    uu = (u * u).sum(-1, keepdims=True)
    t = np.where(uu > 0, ((o - a) * u).sum(-1, keepdims=True) / np.where(uu > 0, uu, 1.0), 0.0)
    c = a + np.clip(t, 0.0, 1.0) * u
    diff = o - c
    dist = np.sqrt((diff * diff).sum(-1, keepdims=True))
    p_obs = c + capsules[None, None, :, 3:4].astype(np.float64) * diff / dist
    p_link = np.broadcast_to(o, (R, Cn, K, 3))
    return (p_link.reshape(R, Cn * K, 3).astype(np.float32).copy(),
            p_obs.reshape(R, Cn * K, 3).astype(np.float32).copy())


def pairs_from_spheres(origins: np.ndarray, spheres: np.ndarray):
    """Explicit closest-point pairs equivalent to the sphere table: origins [R, C, 3] (control
    point = frame origin), spheres [K,4] -> p_link, p_obs [R, C*K, 3] (fp32)."""
    R, Cn, _ = origins.shape
    K = spheres.shape[0]
    o = origins.astype(np.float32)[:, :, None, :]
    c = spheres[None, None, :, :3].astype(np.float32)
    diff = o - c
    dist = np.sqrt((diff * diff).sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)
    p_obs = (c + spheres[None, None, :, 3:4] * (diff / dist)).astype(np.float32)
    p_link = np.broadcast_to(o, (R, Cn, K, 3)).astype(np.float32)
    return p_link.reshape(R, Cn * K, 3).copy(), p_obs.reshape(R, Cn * K, 3).copy()


def sample_ragged(rng: np.random.Generator, R: int, K: int = N_SPHERES):
    """Config 5: per-robot obstacle count k_r ~ U{0..K} as a CSR index list into the shared table."""
    counts = rng.integers(0, K + 1, size=R)
    offset = np.zeros(R + 1, np.int32)
    offset[1:] = np.cumsum(counts)
    index = np.concatenate([rng.permutation(K)[:c] for c in counts]).astype(np.int32) if offset[-1] else \
        np.zeros(0, np.int32)
    return offset, index


def pairs_from_link_capsules(T: np.ndarray, link_capsules: np.ndarray, table: np.ndarray, dtype=np.float64):
    """Closest points of LINK capsules and obstacle primitives in fp64 numpy (independent of the engine): T [R, C, 4, 4]
    world transforms of the distance leaves' frames, link_capsules [C, 8] in frame coordinates, table [K, 4] spheres or
    [K, 8] capsules -> p_link, p_obs [R, C*K, 3].  Segment-segment closest points by the clamped normal equations."""
    # (dtype = np.float32: the same closed form in fp32 arithmetic -- what ANY fp32 evaluation of it can resolve; tools/fuzz_parity.py
    #  takes the difference of the oracle's results on the two pair sets as the resolution of a link-geometry case: nearly parallel
    #  segments make the normal equations ill-conditioned)
    T = T.astype(dtype)
    R, Cn = T.shape[:2]
    K = table.shape[0]
    lc = link_capsules.astype(dtype)
    A = T[:, :, :3, 3] + np.einsum("rcij,cj->rci", T[:, :, :3, :3], lc[:, 0:3])
    B = T[:, :, :3, 3] + np.einsum("rcij,cj->rci", T[:, :, :3, :3], lc[:, 4:7])
    rl = lc[:, 3]
    tb = table.astype(dtype)
    Cc = tb[:, 0:3]
    Dd = tb[:, 4:7] if tb.shape[1] == 8 else tb[:, 0:3]
    ro = tb[:, 3]
    p1, q1 = A[:, :, None, :], B[:, :, None, :]
    p2, q2 = Cc[None, None, :, :], Dd[None, None, :, :]
    d1, d2, r = q1 - p1, np.broadcast_to(q2 - p2, (R, Cn, K, 3)), p1 - p2
    a = (d1 * d1).sum(-1) + 0 * r[..., 0]
    e = (d2 * d2).sum(-1) + 0 * r[..., 0]
    f = (d2 * r).sum(-1)
    c = (d1 * r).sum(-1)
    b = (d1 * d2).sum(-1)
    with np.errstate(all="ignore"):
        denom = a * e - b * b
        s = np.where(denom > 0, np.clip((b * f - c * e) / np.where(denom > 0, denom, 1.0), 0, 1), 0.0)
        t = np.where(e > 0, (b * s + f) / np.where(e > 0, e, 1.0), 0.0)
        s = np.where(t < 0, np.clip(-c / np.where(a > 0, a, 1.0), 0, 1), np.where(t > 1, np.clip((b - c) / np.where(a > 0, a, 1.0), 0, 1), s))
        t = np.clip(t, 0, 1)
        s = np.where(a > 0, s, 0.0)
        s = np.where((e > 0) | (a <= 0), s, np.clip(-c / np.where(a > 0, a, 1.0), 0, 1))
    X = p1 + s[..., None] * d1
    Y = p2 + t[..., None] * d2
    n = X - Y
    nn = np.linalg.norm(n, axis=-1, keepdims=True)
    # intersecting axes: no common normal -- the fixed direction +z (as the device form, rmp2_device.h link_pair_fields)
    n = np.where(nn == 0, np.array([0.0, 0.0, 1.0], dtype=n.dtype), n / np.where(nn == 0, 1.0, nn))
    p_link = X - rl[None, :, None, None] * n
    p_obs = Y + ro[None, None, :, None] * n
    return p_link.reshape(R, Cn * K, 3).astype(np.float32), p_obs.reshape(R, Cn * K, 3).astype(np.float32)


# ---- the reference's own obstacle primitive: finite cylinders with flat caps (simulation.py:245-261) ---------------------------

def cylinder_record(position, rpy, radius, height) -> np.ndarray:
    """(cx, cy, cz, radius, ux, uy, uz, half_height) of a `Cylinder(base_position, base_orientation, radius, height)` of the reference
    (simulation.py:245-261; the orientation is an Euler triple handed to PyBullet's getQuaternionFromEuler: R = Rz(yaw) Ry(pitch)
    Rx(roll), the cylinder's axis is its local z)."""
    roll, pitch, yaw = (float(a) for a in rpy)
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    axis = np.array([cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr])
    return np.concatenate([np.asarray(position, np.float64), [radius], axis / np.linalg.norm(axis), [0.5 * height]]).astype(np.float32)


# experiments/franka_panda/06_cluttered_environment.py:39-52: the seven cylinders of the cluttered scene
EXP06_CYLINDERS = np.stack([
    cylinder_record([0.35, -0.2, 0.55], [0.1, 0, 0], 0.025, 0.2),
    cylinder_record([0.1, -0.4, 0.125], [0.1, 0, 0], 0.025, 0.3),
    cylinder_record([0.33, -0.3, 0.7], [-1.7, 0.7, 0], 0.025, 0.3),
    cylinder_record([0.55, 0.5 - 0.25, 0.5], [0.1, 0, 0], 0.025, 0.3),
    cylinder_record([0.8, 0.5 - 0.25, 0.3], [0.1, 0, 0], 0.025, 0.3),
    cylinder_record([0.5, 0.5 - 0.1, 0.31], [3.14 / 2, 0, 0], 0.025, 0.3),
    cylinder_record([0.35 + 0.1, 0.5 - 0.4, 0.11], [3.14 / 2, 0, 0], 0.025, 0.3),
])
EXP06_GOAL = np.array([0.2, -0.2, 0.5], dtype=np.float32)   # 06_cluttered_environment.py:36


def sample_cylinders(rng: np.random.Generator, K: int = N_SPHERES) -> np.ndarray:
    """The reference's cylinder clutter as what it is (simulation.py:495-500, the draws of sample_capsules): [K, 8] cylinder records
    of height 0.5."""
    r, phi, z = rng.uniform(0.4, 0.9, K), rng.uniform(0, 2 * np.pi, K), rng.uniform(0, 1, K)
    rpy = rng.uniform(0, np.pi, size=(K, 3))
    rad = rng.uniform(0.05, 0.1, K)
    return np.stack([cylinder_record([r[k] * np.cos(phi[k]), r[k] * np.sin(phi[k]), z[k]], rpy[k], rad[k], 0.5) for k in range(K)])


def point_cylinder_np(p: np.ndarray, cyl: np.ndarray):
    """fp64 numpy, independent of the engine and of the C oracle: p [..., 3] against cylinder records cyl [..., 8] (broadcast) ->
    (Y nearest surface point, n outward unit normal there, sd signed distance: p = Y + sd n).  Outside: nearest point of the solid;
    inside: the nearer of side and cap.  (tests/test_oracle_pins.py pins it by a brute-force scan of the surface.)"""
    p = np.asarray(p, np.float64)
    cyl = np.asarray(cyl, np.float64)
    c, r, u, h = cyl[..., 0:3], cyl[..., 3], cyl[..., 4:7], cyl[..., 7]
    w = p - c
    a = (w * u).sum(-1)
    rv = w - a[..., None] * u
    rho = np.linalg.norm(rv, axis=-1)
    # radial direction; on the axis: the fixed perpendicular the engine takes
    ax = np.abs(u)
    kx = (ax[..., 0] <= ax[..., 1]) & (ax[..., 0] <= ax[..., 2])
    ky = ~kx & (ax[..., 1] <= ax[..., 2])
    t = np.stack([kx, ky, ~kx & ~ky], axis=-1).astype(np.float64)
    perp = np.cross(u, t)
    perp /= np.linalg.norm(perp, axis=-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.where((rho > 0)[..., None], rv / np.where(rho > 0, rho, 1.0)[..., None], perp)
    sa = np.where(a < 0, -1.0, 1.0)
    da, dr = np.abs(a) - h, rho - r
    inside = (da <= 0) & (dr <= 0)
    side_in = inside & (dr >= da)
    cap_in = inside & ~side_in
    ac = np.where(side_in, a, np.where(cap_in, sa * h, np.clip(a, -h, h)))
    rc = np.where(side_in, r, np.where(cap_in, rho, np.minimum(rho, r)))
    Y = c + ac[..., None] * u + rc[..., None] * e
    ga, gr = a - ac, rho - rc
    gap = np.sqrt(ga * ga + gr * gr)
    with np.errstate(invalid="ignore", divide="ignore"):
        n_out = (ga[..., None] * u + gr[..., None] * e) / np.where(gap > 0, gap, 1.0)[..., None]
    n = np.where(side_in[..., None], e, np.where(cap_in[..., None], sa[..., None] * u, n_out))
    sd = np.where(side_in, dr, np.where(cap_in, da, gap))
    return Y, n, sd


def pairs_from_cylinders(origins: np.ndarray, cylinders: np.ndarray):
    """Explicit closest-point pairs of frame-origin control points against a cylinder table, fp64 numpy: origins [R, C, 3],
    cylinders [K, 8] -> p_link, p_obs [R, C*K, 3] (p_link = the origin, p_obs = the nearest point of the cylinder's surface)."""
    R, Cn, _ = origins.shape
    K = cylinders.shape[0]
    o = np.broadcast_to(origins.astype(np.float64)[:, :, None, :], (R, Cn, K, 3))
    Y, _, _ = point_cylinder_np(o, cylinders[None, None, :, :])
    return o.reshape(R, Cn * K, 3).astype(np.float32).copy(), Y.reshape(R, Cn * K, 3).astype(np.float32).copy()


def pairs_from_link_capsules_cylinders(T: np.ndarray, link_capsules: np.ndarray, cylinders: np.ndarray, iters: int = 60):
    """Closest points of LINK capsules and cylinders in fp64 numpy: the signed distance of a point to a convex body is convex along
    the link's axis, its derivative there n(s) . D monotone -- bisection on its sign (60 halvings).  T [R, C, 4, 4], link_capsules
    [C, 8], cylinders [K, 8] -> p_link = X - r_link n, p_obs = Y, each [R, C*K, 3]."""
    T = T.astype(np.float64)
    R, Cn = T.shape[:2]
    K = cylinders.shape[0]
    lc = link_capsules.astype(np.float64)
    A = (T[:, :, :3, 3] + np.einsum("rcij,cj->rci", T[:, :, :3, :3], lc[:, 0:3]))[:, :, None, :]
    B = (T[:, :, :3, 3] + np.einsum("rcij,cj->rci", T[:, :, :3, :3], lc[:, 4:7]))[:, :, None, :]
    Dv = np.broadcast_to(B - A, (R, Cn, K, 3))
    cyl = cylinders[None, None, :, :]

    def slope(s):
        X = A + s[..., None] * Dv
        Y, n, sd = point_cylinder_np(X, cyl)
        return (n * Dv).sum(-1), X, Y, n

    g0 = slope(np.zeros((R, Cn, K)))[0]
    g1 = slope(np.ones((R, Cn, K)))[0]
    lo, hi = np.zeros((R, Cn, K)), np.ones((R, Cn, K))
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        g = slope(mid)[0]
        lo = np.where(g < 0, mid, lo)
        hi = np.where(g < 0, hi, mid)
    s = np.where(~(g0 < 0), 0.0, np.where(~(g1 > 0), 1.0, 0.5 * (lo + hi)))
    _, X, Y, n = slope(s)
    p_link = X - lc[None, :, None, 3:4] * n
    return p_link.reshape(R, Cn * K, 3).astype(np.float32), Y.reshape(R, Cn * K, 3).astype(np.float32)
