"""RMP2-style leaf policies with the reference's names and constructor signatures
(rmp2.py:6-226).  Each class is a *descriptor*: it stores its parameters and serialises to
one `rmp2_leaf` record; the (xdd_des, A) formulas themselves run inside the HIP kernels
(csrc/rmp2_device.h, csrc/rmp2_quad.h), quirks included.  The reference's leaf protocol `rmp.evaluate(x, xd)` is
served by a small kernel of its own (rmp2_leaf_evaluate); RmpCore.evaluate never goes through it.
"""
from __future__ import annotations

import numpy as np

from . import descriptor as D
from .taskmap import IdentityTaskmap, classify


class LeafResult(np.ndarray):
    """ndarray with the `.numpy()` callers of the reference's tensors use."""

    def numpy(self):
        return np.asarray(self)


def as_plain(v):
    """Array view of a goal / data holder (list, ndarray, or a Datamanager-style holder with .numpy())."""
    if hasattr(v, "detach"):
        return v.detach().cpu().numpy()
    return v.numpy() if hasattr(v, "numpy") else v


class RiemannianMotionPolicy:
    """Abstract base (rmp2.py:6-29)."""

    KIND = None

    def __init__(self, name, taskmap):
        self.name = name
        self.taskmap = taskmap

    def evaluate(self, x, xd, device: int = 0):
        """The reference's leaf protocol (rmp2.py:25-29, rmp.py:202-206): x, xd [B, k] (or [k]) -> (xdd_des [B, k],
        A [B, k, k]) as ndarrays with `.numpy()`, computed on the GPU by rmp2_leaf_evaluate.  RmpCore.evaluate does not
        use this (the leaves run inside the fused control step); it exists for callers that inspect one leaf."""
        import ctypes as C
        import torch
        from . import _native
        if not torch.cuda.is_available():
            raise _native.Rmp2Error("no HIP device visible; leaf evaluation has no CPU fallback")
        dev = torch.device("cuda", device)
        xt = torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(-1, np.asarray(x).shape[-1]).to(dev).contiguous()
        vt = torch.as_tensor(np.asarray(xd, dtype=np.float32)).reshape(xt.shape).to(dev).contiguous()
        B, k = xt.shape
        spec = D.LeafSpec(self.KIND, D.TASKMAP_IDENTITY, -1, self._params(), *self._vectors(), name=self.name)
        rec = D.Leaf()
        rec.kind = spec.kind
        for i, p in enumerate(spec.params):
            rec.params[i] = p
        for name, vec in (("vec_a", spec.vec_a), ("vec_b", spec.vec_b)):
            if vec is not None:
                if len(vec) != k:
                    raise ValueError(f"{type(self).__name__}: {name} has {len(vec)} entries, x has {k} columns")
                for i, v in enumerate(vec):
                    getattr(rec, name)[i] = v
        g = self._goal()
        gt = None if g is None else torch.as_tensor(np.asarray(as_plain(g), dtype=np.float32).reshape(-1)).to(dev)
        if gt is not None and gt.numel() != k:
            raise ValueError(f"{type(self).__name__}: goal has {gt.numel()} entries, x has {k} columns")
        dist = nvec = None
        if self.KIND == D.LEAF_COLLISION_AVOIDANCE:
            dist = torch.as_tensor(np.asarray(as_plain(self.d), dtype=np.float32).reshape(-1)).to(dev).contiguous()
            nvec = torch.as_tensor(np.asarray(as_plain(self.vec), dtype=np.float32).reshape(-1, 3)).to(dev).contiguous()
            if dist.numel() != B or nvec.shape[0] != B:
                raise ValueError("CollisionAvoidance: d / vec must have one entry per row of x")
        xdd = torch.empty((B, k), dtype=torch.float32, device=dev)
        A = torch.empty((B, k, k), dtype=torch.float32, device=dev)
        ptr = lambda t: None if t is None else t.data_ptr()
        rc = _native.lib().rmp2_leaf_evaluate(device, C.byref(rec), k, xt.data_ptr(), vt.data_ptr(), ptr(gt), ptr(dist),
                                              ptr(nvec), xdd.data_ptr(), A.data_ptr(), B,
                                              torch.cuda.current_stream(dev).cuda_stream)
        _native.check(rc)
        torch.cuda.synchronize(dev)
        return xdd.cpu().numpy().view(LeafResult), A.cpu().numpy().view(LeafResult)

    # descriptor protocol -----------------------------------------------------------
    def _params(self):
        raise NotImplementedError

    def _vectors(self):
        return None, None

    def _goal(self):
        return None

    def _allowed_taskmaps(self):
        return (D.TASKMAP_IDENTITY,)

    def leaf_spec(self, frame_index_of) -> D.LeafSpec:
        kind, fk, _ = classify(self.taskmap)
        if kind not in self._allowed_taskmaps():
            raise NotImplementedError(f"{type(self).__name__} on task map kind {kind} has no kernel")
        frame = frame_index_of(fk) if fk is not None else -1
        va, vb = self._vectors()
        g = self._goal()
        return D.LeafSpec(self.KIND, kind, frame, self._params(), va, vb,
                          goal_len=0 if g is None else int(g.shape[-1] if hasattr(g, 'shape') else np.asarray(g).shape[-1]), name=self.name)


class TargetAttractor(RiemannianMotionPolicy):
    """rmp2.py:31-83."""
    KIND = D.LEAF_TARGET_ATTRACTOR

    def __init__(self, goal, accel_p_gain, accel_d_gain, accel_norm_eps, metric_alpha_length_scale,
                 min_metric_alpha, max_metric_scalar, min_metric_scalar, proximity_metric_boost_scalar,
                 proximity_metric_boost_length_scale, taskmap, name='attractor'):
        super().__init__(name, taskmap)
        self.goal = goal
        self.accel_p_gain = accel_p_gain
        self.accel_d_gain = accel_d_gain
        self.accel_norm_eps = accel_norm_eps
        self.metric_alpha_length_scale = metric_alpha_length_scale
        self.min_metric_alpha = min_metric_alpha
        self.max_metric_scalar = max_metric_scalar
        self.min_metric_scalar = min_metric_scalar
        self.proximity_metric_boost_scalar = proximity_metric_boost_scalar
        self.proximity_metric_boost_length_scale = proximity_metric_boost_length_scale

    def _params(self):
        return [self.accel_p_gain, self.accel_d_gain, self.accel_norm_eps, self.metric_alpha_length_scale,
                self.min_metric_alpha, self.max_metric_scalar, self.min_metric_scalar,
                self.proximity_metric_boost_scalar, self.proximity_metric_boost_length_scale]

    def _goal(self):
        return self.goal

    def _allowed_taskmaps(self):
        return (D.TASKMAP_FK_POSITION,)


class JointVelocityCap(RiemannianMotionPolicy):
    """rmp2.py:86-112 (dense, indefinite metric with a pole: quirk Q4, reproduced)."""
    KIND = D.LEAF_JOINT_VELOCITY_CAP

    def __init__(self, max_velocity, velocity_damping_region, damping_gain, metric_weight,
                 name='joint_velocity_cap'):
        super().__init__(name, taskmap=IdentityTaskmap())
        self.max_velocity = max_velocity
        self.velocity_damping_region = velocity_damping_region
        self.damping_gain = damping_gain
        self.metric_weight = metric_weight
        self.eps = 1e-6
        self.damped_velocity_cutoff = self.max_velocity - self.velocity_damping_region

    def _params(self):
        return [self.max_velocity, self.velocity_damping_region, self.damping_gain, self.metric_weight]


class JointDamping(RiemannianMotionPolicy):
    """rmp2.py:115-137."""
    KIND = D.LEAF_JOINT_DAMPING

    def __init__(self, accel_d_gain, metric_scalar, inertia, name='joint_damping'):
        super().__init__(name=name, taskmap=IdentityTaskmap())
        self.accel_d_gain = accel_d_gain
        self.metric_scalar = metric_scalar
        self.inertia = inertia

    def _params(self):
        return [self.accel_d_gain, self.metric_scalar, self.inertia]


class ObstacleAvoidance(RiemannianMotionPolicy):
    """rmp2.py:140-196; one instance per control-point frame, B pairs each."""
    KIND = D.LEAF_OBSTACLE_AVOIDANCE

    def __init__(self, margin, damping_gain, damping_std_dev, damping_robustness_eps,
                 damping_velocity_gate_length_scale, repulsion_gain, repulsion_std_dev, metric_modulation_radius,
                 metric_scalar, metric_exploder_std_dev, metric_exploder_eps, taskmap, name):
        super().__init__(name=name, taskmap=taskmap)
        self.margin = margin
        self.damping_gain = damping_gain
        self.damping_std_dev = damping_std_dev
        self.damping_robustness_eps = damping_robustness_eps
        self.damping_velocity_gate_length_scale = damping_velocity_gate_length_scale
        self.repulsion_gain = repulsion_gain
        self.repulsion_std_dev = repulsion_std_dev
        self.metric_modulation_radius = metric_modulation_radius
        self.metric_scalar = metric_scalar
        self.metric_exploder_std_dev = metric_exploder_std_dev
        self.metric_exploder_eps = metric_exploder_eps

    def _params(self):
        return [self.margin, self.damping_gain, self.damping_std_dev, self.damping_robustness_eps,
                self.damping_velocity_gate_length_scale, self.repulsion_gain, self.repulsion_std_dev,
                self.metric_modulation_radius, self.metric_scalar, self.metric_exploder_std_dev,
                self.metric_exploder_eps]

    def _allowed_taskmaps(self):
        return (D.TASKMAP_FK_DISTANCE,)


class CSpaceBiasing(RiemannianMotionPolicy):
    """rmp2.py:198-226 (configuration-space target reaching)."""
    KIND = D.LEAF_CSPACE_BIASING

    def __init__(self, goal, metric_scalar, position_gain, damping_gain, robust_position_term_thresh, inertia,
                 taskmap=None, name='cspace_target'):
        super().__init__(name=name, taskmap=taskmap if taskmap is not None else IdentityTaskmap())
        self.goal = goal
        self.metric_scalar = metric_scalar
        self.position_gain = position_gain
        self.damping_gain = damping_gain
        self.robust_position_term_thresh = robust_position_term_thresh
        self.inertia = inertia

    def _params(self):
        return [self.metric_scalar, self.position_gain, self.damping_gain, self.robust_position_term_thresh,
                self.inertia]

    def _vectors(self):
        return np.asarray(self.goal, dtype=np.float32), None
