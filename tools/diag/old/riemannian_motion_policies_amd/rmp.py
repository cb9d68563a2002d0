"""`RmpCore` and the older leaf policies, with the reference's names (rmp.py:111-382).

RmpCore keeps the reference's registry protocol (add_rmp / remove_rmp_by_name / __str__,
rmp.py:114-131) but `evaluate` no longer loops over policies in Python: the whole set is
compiled to a flat descriptor once (re-compiled only when the set changes) and each call is
one fused HIP control step for all robots in `q`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import descriptor as D
from .data_management import as_array
from .rmp2 import RiemannianMotionPolicy
from .taskmap import (IdentityTaskmap, TaskmapJointFrame4x4ToDistance, TaskmapSphereDistance, classify)
from .urdf import KinematicTable


class QddResult(np.ndarray):
    """ndarray with the `.numpy()` the reference's callers use (06_cluttered_environment.py:124)."""

    def numpy(self):
        return np.asarray(self)


def _null_table(n_dof: int) -> KinematicTable:
    """Robot with no frames: identity-task-map-only sets (e.g. experiments/two_joint_robot/03)."""
    z = np.zeros
    return KinematicTable(frame_names=[], parent=z(0, np.int32), joint_type=z(0, np.int32), q_index=z(0, np.int32),
                          axis=z((0, 3), np.float32), T_const=z((0, 4, 4), np.float32), has_collision=z(0, bool),
                          order=[f"q{i}" for i in range(n_dof)], link_names=[], limits_lower=z(0, np.float32),
                          limits_upper=z(0, np.float32))


class RmpCore:
    """Manages multiple RMPs and resolves them into one joint acceleration (rmp.py:111-155).

    Differences to the reference, all additive: `evaluate` also accepts a fleet `q[R,n]`;
    `device` / `solve` select the GPU and the resolve mode ("auto": LU with pseudo-inverse
    fall-through, "pinv": always the reference's pseudo-inverse).  The reference's shared
    mutable default `rmps={}` (rmp.py:114) is NOT reproduced: each core owns its dict.
    """

    def __init__(self, rmps=None, device: int = 0, solve: str = "auto"):
        self.rmps = {} if rmps is None else rmps
        self.device = device
        self.solve = solve
        self.spheres = None  # shared sphere table [K,4] for TaskmapSphereDistance leaves
        self._engine = None
        self._signature = None

    def __str__(self):
        out = ''
        if len(self.rmps) > 0:
            out += '\n' + 'used RMPs:' + '\n'
            for i, rmp in enumerate(self.rmps.values()):
                out += '\t'.join([str(i), rmp.name, str(type(rmp))]) + '\n'
        else:
            out += 'no RMPs in use.' + '\n'
        return out

    def add_rmp(self, rmp):
        self.rmps[rmp.name] = rmp

    def remove_rmp_by_name(self, name):
        self.rmps.pop(name)

    # ------------------------------------------------------------------------------
    def _compile(self, n_dof):
        fks = []
        for rmp in self.rmps.values():
            _, fk, _ = classify(rmp.taskmap)
            if fk is not None and all(fk.fkine is not f for f in fks):
                fks.append(fk.fkine)
        if len(fks) > 1:
            raise NotImplementedError("all FK task maps of one RmpCore must share one UrdfForwardKinematic")
        table = fks[0].table if fks else _null_table(n_dof)
        if table.n_dof != n_dof:
            raise ValueError(f"q has {n_dof} entries, the robot has {table.n_dof} dof")
        specs = [rmp.leaf_spec(lambda fk: table.frame_index(fk.frame)) for rmp in self.rmps.values()]
        sig = (id(table), self.solve, tuple(s.signature() for s in specs))
        if sig != self._signature:
            from .engine import Engine
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(D.build_desc(table, specs, self.solve), self.device)
            self._signature = sig
        return self._engine

    def evaluate(self, q, qd, spheres=None):
        """q, qd: [n] (one robot, as in the reference) or [R, n].  Returns qdd of the same shape."""
        single = np.ndim(q) == 1
        q2 = np.atleast_2d(np.asarray(as_array(q), dtype=np.float32))
        qd2 = np.atleast_2d(np.asarray(as_array(qd), dtype=np.float32))
        R, n = q2.shape
        eng = self._compile(n)
        # goals are mutable attributes of the leaves (01_target_rmp_only.py:61-63): re-read every call
        goals = []
        per_robot = False
        for rmp in self.rmps.values():
            g = rmp._goal()
            if g is not None:
                g = np.asarray(as_array(g), dtype=np.float32)
                per_robot |= g.ndim == 2
                goals.append(g)
        goal = None
        if goals:
            goal = np.concatenate([g if g.ndim == 2 else np.broadcast_to(g, (R, g.shape[0])) for g in goals], axis=1) \
                if per_robot else np.concatenate(goals)
        # distance data: gather the holders of all distance leaves
        obstacles = None
        dist = [rmp.taskmap.stages()[-1] for rmp in self.rmps.values() if classify(rmp.taskmap)[0] == D.TASKMAP_FK_DISTANCE]
        if dist:
            if all(isinstance(t, TaskmapJointFrame4x4ToDistance) for t in dist):
                pl = [np.asarray(as_array(t.pos_on_link_in_base_frame), np.float32) for t in dist]
                po = [np.asarray(as_array(t.pos_on_obstacle_in_base_frame), np.float32) for t in dist]
                pl = [a if a.ndim == 3 else np.broadcast_to(a, (R,) + a.shape) for a in pl]
                po = [a if a.ndim == 3 else np.broadcast_to(a, (R,) + a.shape) for a in po]
                obstacles = eng.obstacles(p_link=np.concatenate(pl, axis=1), p_obs=np.concatenate(po, axis=1),
                                          pair_counts=[a.shape[1] for a in pl])
            elif all(isinstance(t, TaskmapSphereDistance) for t in dist):
                sp = spheres if spheres is not None else self.spheres
                if sp is None:
                    raise ValueError("TaskmapSphereDistance leaves need evaluate(..., spheres=[K,4])")
                obstacles = eng.obstacles(spheres=as_array(sp) if not isinstance(sp, torch.Tensor) else sp)
            else:
                raise NotImplementedError("mixing explicit-pair and sphere distance task maps in one core")
        out = eng.step(q2, qd2, goal=goal, obstacles=obstacles)
        res = out.cpu().numpy()
        return (res[0] if single else res).view(QddResult)


# ---- older leaf policies (rmp.py:226-382) ----------------------------------------------

class TargetPolicy(RiemannianMotionPolicy):
    """rmp.py:226-260 (quirk Q8: c*log in h, 1/c*log in soft_norm)."""
    KIND = D.LEAF_TARGET_POLICY

    def __init__(self, alpha, beta, c, goal, taskmap, name='Target_RMP'):
        super().__init__(name, taskmap)
        self.goal = goal
        self.c = c
        self.alpha = alpha
        self.beta = beta
        self.sigma_H = 1
        self.sigma_w = 3

    def _params(self):
        return [self.alpha, self.beta, self.c]

    def _goal(self):
        return self.goal

    def _allowed_taskmaps(self):
        return (D.TASKMAP_FK_POSITION, D.TASKMAP_IDENTITY)


class CollisionAvoidance(RiemannianMotionPolicy):
    """rmp.py:264-315 -- TwoJoint experiment 05 only; SURVEY 8(f)-4 ("next" row), no kernel yet."""

    def __init__(self, d, vec, eta_rep, nu_rep, eta_damp, nu_damp, r, c, taskmap, name='collision_avoidance'):
        super().__init__(name, taskmap)
        self.d, self.vec = d, vec
        self.eta_rep, self.nu_rep, self.eta_damp, self.nu_damp, self.r, self.c = eta_rep, nu_rep, eta_damp, nu_damp, r, c

    def leaf_spec(self, frame_index_of):
        raise NotImplementedError("CollisionAvoidance (rmp.py:264-315) is outside the accelerated path (SURVEY 8(f)-4)")


class ConfigurationSpaceBiasing(RiemannianMotionPolicy):
    """rmp.py:318-347: PD controller towards q0 with metric w*I."""
    KIND = D.LEAF_CONFIG_SPACE_BIASING

    def __init__(self, gamma_p, gamma_d, q0, name, w=0.05):
        super().__init__(name, taskmap=IdentityTaskmap())
        self.gamma_p = gamma_p
        self.gamma_d = gamma_d
        self.q_0 = q0
        self.w = w

    def _params(self):
        return [self.gamma_p, self.gamma_d, self.w]

    def _vectors(self):
        return np.asarray(self.q_0, dtype=np.float32), None


class JointLimitAvoidance(RiemannianMotionPolicy):
    """rmp.py:349-382 (non-symmetric metric H*diag(w): quirk Q2, reproduced)."""
    KIND = D.LEAF_JOINT_LIMIT_AVOIDANCE

    def __init__(self, lower_limits, upper_limits, gamma_p, gamma_d, name='joint_limit_avoidance'):
        super().__init__(name, taskmap=IdentityTaskmap())
        self.lower_limits = np.asarray(lower_limits, dtype=np.float32)
        self.upper_limits = np.asarray(upper_limits, dtype=np.float32)
        self.gamma_p = gamma_p
        self.gamma_d = gamma_d

    def _params(self):
        return [self.gamma_p, self.gamma_d]

    def _vectors(self):
        return self.lower_limits, self.upper_limits
