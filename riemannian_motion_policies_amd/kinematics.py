"""`UrdfForwardKinematic` with the reference's constructor and methods (kinematics.py:155-270).

Setup (URDF -> tables) runs on the host once (urdf.py); `forward` / `differentiate` run on
the GPU through rmp2_forward_kinematics / rmp2_differentiate.
"""
from __future__ import annotations

import numpy as np
import torch

from . import descriptor as D
from .taskmap import _to_str
from .urdf import KinematicTable, compile_urdf


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
