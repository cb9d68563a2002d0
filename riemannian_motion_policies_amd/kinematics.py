"""`UrdfForwardKinematic` with the reference's constructor and methods (kinematics.py:155-270), and the module-level
rotation helpers the reference's scripts and tests import from `kinematics` (kinematics.py:22-152).

Setup (URDF -> tables) runs on the host once (urdf.py); `forward` / `differentiate` run on
the GPU through rmp2_forward_kinematics / rmp2_differentiate.  The rotation helpers are host-side fp32 utilities with the
reference's names, argument shapes and batch convention (leading batch axis, results answer `.numpy()`); the control step
does not call them -- the device walk forms its joint rotations itself (Rodrigues, csrc/rmp2_device.h).
"""
from __future__ import annotations

import numpy as np
import torch

from . import descriptor as D
from .taskmap import _to_str
from .urdf import KinematicTable, compile_urdf


class HostTensor(np.ndarray):
    """An ndarray that also answers `.numpy()`, as the reference's tf.Tensor results do."""

    def numpy(self):
        return np.asarray(self)


def _host(a) -> HostTensor:
    return np.ascontiguousarray(a, dtype=np.float32).view(HostTensor)


def _f32(a) -> np.ndarray:
    return np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a, dtype=np.float32)


def reduce_matrix_prod(all_T) -> HostTensor:
    """all_T [N,4,4] -> T_0 @ T_1 @ ... @ T_{N-1} (kinematics.py:11-20: how the reference's FK chains a frame's transforms; the
    device walk multiplies along the kinematic tree itself)."""
    A = _f32(all_T)
    if A.ndim != 3 or A.shape[-2:] != (4, 4):
        raise ValueError(f"expected [N,4,4], got {A.shape}")
    m = np.eye(4, dtype=np.float32)
    for k in range(A.shape[0]):
        m = (m @ A[k]).astype(np.float32)
    return _host(m)


def _axis_rotation(angle, axis: int) -> HostTensor:
    a = _f32(angle)
    if a.ndim != 2 or a.shape[1] != 1:
        raise ValueError(f"angle must have shape [batch, 1], got {a.shape}")  # the reference's input_signature
    c, s = np.cos(a[:, 0]), np.sin(a[:, 0])
    i, j = [(1, 2), (2, 0), (0, 1)][axis]  # the plane the axis turns: R[i,i] = c, R[i,j] = -s, R[j,i] = s, R[j,j] = c
    R = np.zeros((a.shape[0], 3, 3), dtype=np.float32)
    R[:, axis, axis] = 1.0
    R[:, i, i], R[:, i, j], R[:, j, i], R[:, j, j] = c, -s, s, c
    return _host(R)


def R_x(angle) -> HostTensor:
    """angle [B,1] -> [B,3,3] rotation about x (kinematics.py:22-32)."""
    return _axis_rotation(angle, 0)


def R_y(angle) -> HostTensor:
    """angle [B,1] -> [B,3,3] rotation about y (kinematics.py:34-44)."""
    return _axis_rotation(angle, 1)


def R_z(angle) -> HostTensor:
    """angle [B,1] -> [B,3,3] rotation about z (kinematics.py:46-56)."""
    return _axis_rotation(angle, 2)


def homogenous_transformation(R, t) -> HostTensor:
    """R [B,3,3], t [B,3] -> T [B,4,4] = [[R, t], [0, 1]] (kinematics.py:58-71)."""
    R, t = _f32(R), _f32(t)
    if R.ndim != 3 or R.shape[-2:] != (3, 3) or t.shape != (R.shape[0], 3):
        raise ValueError(f"expected R [B,3,3] and t [B,3], got {R.shape} and {t.shape}")
    T = np.zeros((R.shape[0], 4, 4), dtype=np.float32)
    T[:, :3, :3], T[:, :3, 3], T[:, 3, 3] = R, t, 1.0
    return _host(T)


def euler_from_rotation_matrix(rotation_matrix) -> HostTensor:
    """R [B,3,3] -> (theta_x, theta_y, theta_z) [B,3] with R = Rz Ry Rx (kinematics.py:74-96): theta_y = -asin(r20), the
    other two by atan2 of entries divided by cos(theta_y), that divisor replaced by 1 where |cos| < 1e-6 (gimbal lock).
    The device map of the same name is rmp2_differentiate_euler (taskmap.TaskmapFrom4x4ToEuler)."""
    R = _f32(rotation_matrix)
    ty = -np.arcsin(R[:, 2, 0])
    cy = np.cos(ty)
    safe = np.where(np.abs(cy) < np.float32(1e-6), np.float32(1.0), cy)
    tz = np.arctan2(R[:, 1, 0] / safe, R[:, 0, 0] / safe)
    tx = np.arctan2(R[:, 2, 1] / safe, R[:, 2, 2] / safe)
    return _host(np.stack((tx, ty, tz), axis=-1))


def rotation_matrix_from_rotation_vector(vec, angle) -> HostTensor:
    """unit axes vec [B,3], angle [B] -> [B,3,3] = cos I + sin [u]x + (1 - cos) u u^T (kinematics.py:99-121)."""
    u, a = _f32(vec), _f32(angle)
    if u.ndim != 2 or u.shape[1] != 3 or a.shape != (u.shape[0],):
        raise ValueError(f"expected vec [B,3] and angle [B], got {u.shape} and {a.shape}")
    c, s = np.cos(a)[:, None, None], np.sin(a)[:, None, None]
    K = np.zeros((u.shape[0], 3, 3), dtype=np.float32)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 2] = -u[:, 2], u[:, 1], -u[:, 0]
    K = K - K.transpose(0, 2, 1)
    return _host(c * np.eye(3, dtype=np.float32) + s * K + (1.0 - c) * (u[:, :, None] * u[:, None, :]))


def rotation_matrix_from_rpy(rpy) -> HostTensor:
    """rpy [B,3] -> R_x(roll) @ R_y(pitch) @ R_z(yaw) (kinematics.py:123-127; quirk Q7: the URDF standard multiplies the other
    way round -- the reference's order is what urdf.compile_urdf bakes into the tables)."""
    a = _f32(rpy)
    return _host(R_x(a[:, 0:1]) @ R_y(a[:, 1:2]) @ R_z(a[:, 2:3]))


def rotation_matrix_from_quaternions(quaternions) -> HostTensor:
    """(q0 = w, q1, q2, q3) -> 3 x 3 (kinematics.py:129-152; scalar first, not batched, as the reference)."""
    q0, q1, q2, q3 = (np.float32(v) for v in _f32(quaternions))
    return _host([[2 * (q0 * q0 + q1 * q1) - 1, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
                  [2 * (q1 * q2 + q0 * q3), 2 * (q0 * q0 + q2 * q2) - 1, 2 * (q2 * q3 - q0 * q1)],
                  [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 2 * (q0 * q0 + q3 * q3) - 1]])


def get_H_forEulerXYZ(eulers) -> np.ndarray:
    """The reference's matrix H of the xyz Euler angles (helper/trigonometry_helper.py:19-39, imported into the reference's
    `kinematics` namespace at kinematics.py:8); fp64 like the reference's numpy.  As written there it satisfies
    omega = H d(eulers)/dt (world-frame angular velocity from the angle rates; the reference's docstring states the inverse
    relation, its matrix is this one -- tests/test_rotation_helpers.py checks it against a finite difference)."""
    _, beta, gamma = (float(v) for v in np.asarray(eulers, dtype=np.float64).reshape(3))
    sb, cb, sg, cg = np.sin(beta), np.cos(beta), np.sin(gamma), np.cos(gamma)
    return np.array([[cb * cg, -sg, 0.0], [cb * sg, cg, 0.0], [-sb, 0.0, 1.0]])


class UrdfForwardKinematic:
    def __init__(self, urdf_filepath, order, device: int = 0):
        self.filepath = urdf_filepath
        self.order = list(order)
        self.n_joints = len(self.order)
        self.table: KinematicTable = compile_urdf(urdf_filepath, self.order)
        self.frame_names = list(self.table.frame_names)
        self.device = device
        self._engine = None

    def _fk_engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine(D.build_desc(self.table, []), self.device)
        return self._engine

    def forward(self, q, frame):
        """q [1,n] (or [R,n]) -> T [1,4,4] (or [R,4,4]) of `frame` (kinematics.py:212-247)."""
        q2 = np.atleast_2d(np.asarray(q, dtype=np.float32)) if not isinstance(q, torch.Tensor) else q
        T = self._fk_engine().forward_kinematics(q2)[:, self.table.frame_index(_to_str(frame))]
        return T if isinstance(q, torch.Tensor) else T.cpu().numpy()

    __call__ = forward

    def forward_all(self, q):
        """q [R,n] -> T [R, n_frames, 4, 4] of every frame (order of .frame_names) from one launch."""
        q2 = np.atleast_2d(np.asarray(q, dtype=np.float32)) if not isinstance(q, torch.Tensor) else q
        T = self._fk_engine().forward_kinematics(q2)
        return T if isinstance(q, torch.Tensor) else T.cpu().numpy()

    def differentiate(self, q, qd, frame):
        """-> x[R,16], xd[R,16], J[R,16,n], c[R,16] of vec(T_frame) (kinematics.py:250-270)."""
        as_np = not isinstance(q, torch.Tensor)
        q2 = np.atleast_2d(np.asarray(q, dtype=np.float32)) if as_np else q
        qd2 = np.atleast_2d(np.asarray(qd, dtype=np.float32)) if as_np else qd
        out = self._fk_engine().differentiate(q2, qd2, self.table.frame_index(_to_str(frame)))
        return tuple(o.cpu().numpy() for o in out) if as_np else out

    def differentiate_euler(self, q, qd, frame):
        """-> x[R,3], xd[R,3], J[R,3,n], c[R,3] of the chain [FK(frame), 4x4 -> Euler xyz] (taskmap.py:57-67)."""
        as_np = not isinstance(q, torch.Tensor)
        q2 = np.atleast_2d(np.asarray(q, dtype=np.float32)) if as_np else q
        qd2 = np.atleast_2d(np.asarray(qd, dtype=np.float32)) if as_np else qd
        out = self._fk_engine().differentiate_euler(q2, qd2, self.table.frame_index(_to_str(frame)))
        return tuple(o.cpu().numpy() for o in out) if as_np else out
