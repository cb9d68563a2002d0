"""Array-backed stand-in for the reference's Datamanager (data_management.py:4-53).

The reference keeps per-frame tf.Variables that PyBullet closest-point tuples are stacked
into; the distance task maps hold references to them.  Here the holders are `ArrayVar`s
(assign()/value), the same per-frame / per-key dictionary is offered, and RmpCore.evaluate
gathers the pair arrays of all distance leaves into the [R,P,3] device arrays the kernel
reads.  PyBullet glue (Datamanager.preprocess -> one eager FK per tuple) is out of scope
(SURVEY section 2 row 9); `update` accepts the same tuple list but only stores the points.
"""
from __future__ import annotations

import numpy as np

KEYS3 = ("pos_on_link_in_base_frame", "pos_on_obstacle_in_base_frame", "normal_vec", "relative_position")


class ArrayVar:
    """Mutable array holder with tf.Variable-like assign()."""

    def __init__(self, value):
        self.value = np.asarray(value, dtype=np.float32)

    def assign(self, value):
        self.value = np.asarray(value, dtype=np.float32)
        return self

    def numpy(self):
        return self.value

    def __array__(self, dtype=None, copy=None):
        return self.value if dtype is None else self.value.astype(dtype)


def as_array(holder):
    """numpy view of an ArrayVar / ndarray / torch tensor / tf tensor."""
    if isinstance(holder, ArrayVar):
        return holder.value
    if hasattr(holder, "detach"):
        return holder.detach().cpu().numpy()
    if hasattr(holder, "numpy"):
        return np.asarray(holder.numpy(), dtype=np.float32)
    return np.asarray(holder, dtype=np.float32)


class Datamanager:
    def __init__(self, fkine):
        self.fkine = fkine
        self.state = {
            frame: {**{k: ArrayVar(np.zeros((0, 3), np.float32)) for k in KEYS3},
                    "distance": ArrayVar(np.zeros((0,), np.float32))}
            for frame in fkine.frame_names
        }

    def __getitem__(self, key):
        return self.state[key]

    def update(self, q, distance_data):
        """distance_data: tuples (frame_name, p_link[3], p_obs[3], normal[3], distance, descr)
        as produced by the reference's Simulation.calculate_distances (simulation.py:462-484)."""
        for frame in self.fkine.frame_names:
            rows = [d for d in distance_data if d[0] == frame]
            if not rows:
                continue
            st = self.state[frame]
            st["pos_on_link_in_base_frame"].assign(np.stack([d[1] for d in rows]))
            st["pos_on_obstacle_in_base_frame"].assign(np.stack([d[2] for d in rows]))
            st["normal_vec"].assign(np.stack([d[3] for d in rows]))
            st["distance"].assign(np.asarray([d[4] for d in rows], dtype=np.float32))
