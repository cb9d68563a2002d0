"""Array-backed stand-in for the reference's Datamanager (data_management.py:4-53).

The reference keeps per-frame tf.Variables that PyBullet closest-point tuples are stacked
into; the distance task maps hold references to them.  Here the holders are `ArrayVar`s
(assign()/value), the same per-frame / per-key dictionary is offered, and RmpCore.evaluate
gathers the pair arrays of all distance leaves into the [R,P,3] device arrays the kernel
reads.  `update` accepts the tuple list of the reference's Simulation.calculate_distances and fills
all five fields, 'relative_position' included (preprocess: one device FK per frame).
"""
from __future__ import annotations

import numpy as np

KEYS3 = ("pos_on_link_in_base_frame", "pos_on_obstacle_in_base_frame", "normal_vec", "relative_position")


def _is_tensor(v):
    return hasattr(v, "detach") and hasattr(v, "device")


class ArrayVar:
    """Mutable array holder with tf.Variable-like assign().  The value is a host array or, when the device stages fill it
    (RmpCore.update_distances / Datamanager.update_device), a torch tensor that stays where it is: numpy() copies it to the
    host, as_tensor() hands it to the engine without a host hop.  assign_lazy(fn) defers the value to its first read (the
    device stage derives three of the five Datamanager fields from the other two; most policy sets read none of them)."""

    def __init__(self, value):
        self.assign(value)

    def assign(self, value):
        self._thunk = None
        self._owner = None
        self._value = value.detach() if _is_tensor(value) else np.asarray(value, dtype=np.float32)
        return self

    def assign_lazy(self, fn, owner=None):
        """The value is fn() at its first read.  `owner`: who will produce it (RmpCore's closest-point stage marks the holders
        it feeds, and steps fused -- without ever forming the arrays -- while every holder still carries its mark)."""
        self._thunk, self._value, self._owner = fn, None, owner
        return self

    @property
    def owner(self):
        return self._owner

    @property
    def value(self):
        if self._thunk is not None:
            fn, owner = self._thunk, self._owner
            self.assign(fn())
            self._owner = owner
        return self._value

    @value.setter
    def value(self, v):
        self.assign(v)

    def numpy(self):
        v = self.value
        return v.cpu().numpy() if _is_tensor(v) else v

    def __array__(self, dtype=None, copy=None):
        v = self.numpy()
        return v if dtype is None else v.astype(dtype)


def as_array(holder):
    """numpy view of an ArrayVar / ndarray / torch tensor / tf tensor."""
    if isinstance(holder, ArrayVar):
        return holder.numpy()
    if hasattr(holder, "detach"):
        return holder.detach().cpu().numpy()
    if hasattr(holder, "numpy"):
        return np.asarray(holder.numpy(), dtype=np.float32)
    return np.asarray(holder, dtype=np.float32)


def as_tensor(holder, device):
    """fp32 torch tensor on `device` of an ArrayVar / ndarray / torch tensor: a tensor already there is not copied."""
    import torch
    v = holder.value if isinstance(holder, ArrayVar) else holder
    if _is_tensor(v):
        if v.device == device and v.dtype == torch.float32:
            return v.detach() if v.requires_grad else v
        return v.detach().to(device=device, dtype=torch.float32)
    return torch.from_numpy(np.ascontiguousarray(as_array(v), dtype=np.float32)).to(device)


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
        by_frame = {}
        for d in distance_data:
            by_frame.setdefault(d[0], []).append(d)
        T_all = None
        for frame in self.fkine.frame_names:
            rows = by_frame.get(frame)
            if not rows:
                continue
            st = self.state[frame]
            p_link = np.asarray([d[1] for d in rows], dtype=np.float32)
            st["pos_on_link_in_base_frame"].assign(p_link)
            st["pos_on_obstacle_in_base_frame"].assign(np.asarray([d[2] for d in rows], dtype=np.float32))
            st["normal_vec"].assign(np.asarray([d[3] for d in rows], dtype=np.float32))
            st["distance"].assign(np.asarray([d[4] for d in rows], dtype=np.float32))
            # data_management.py:33-53 preprocess/_get_relative_pos: the point on the link expressed in the joint
            # frame, rel = R_base_joint^T (p_link - p_joint); ONE device FK for all frames instead of one per tuple
            if T_all is None:
                T_all = np.asarray(self.fkine.forward_all(np.asarray(q, dtype=np.float32)[None, :]))[0]
            T = T_all[self.fkine.table.frame_index(frame)]
            st["relative_position"].assign((p_link - T[:3, 3][None, :]) @ T[:3, :3])

    def update_device(self, core, q, primitives, link_capsules=None, primitive=None):
        """The same five fields, filled on the device for a whole fleet without PyBullet and without a host hop: `core` is the
        RmpCore whose distance leaves read this manager's holders; its closest-point stage (rmp2_closest_points_links) writes
        pos_on_link / pos_on_obstacle for every (robot, leaf, primitive) pair, and distance, normal_vec and relative_position
        follow from them (simulation.py:462-484 reports the same tuple; data_management.py:33-53 the relative position).
        q: [R, n] tensor on the core's device; primitives: [K,4] spheres, [K,8] capsules or -- primitive="cylinder" -- [K,8] finite
        cylinders (centre, radius, unit axis, half height: the reference's own obstacles, simulation.py:245-261); link_capsules:
        urdf.link_capsules(...) rows in the order of the core's distance leaves, or None for the frame origins as control points."""
        import torch
        pairs = core.update_distances(q, primitives, link_capsules=link_capsules, primitive=primitive)   # (lazy: nothing has run yet)
        src = pairs.source
        eng, single = src.eng, src.single
        frames = pairs.frames
        L, K = len(frames), src.K
        table = self.fkine.table
        memo = {}

        # distance, normal_vec and relative_position follow from the stage's output: derived on their first read, for all
        # frames at once (the exp-06 set reads none of them; at fleet size each is a pass over a 200 MB array)
        def derived():
            if not memo:
                pl_all, po_all = src.arrays()                                      # [R, L * K, 3]
                T = eng.forward_kinematics(src.q)                                  # [R, F, 4, 4] on the device
                diff = pl_all - po_all
                dist = torch.linalg.norm(diff, dim=-1)
                memo["dist"] = dist
                memo["nvec"] = diff / dist.clamp_min(1e-12).unsqueeze(-1)
                idx = torch.as_tensor([table.frame_index(f) for f in frames], device=eng.device)
                Tf = T.index_select(1, idx)                                        # [R, L, 4, 4]
                memo["rel"] = torch.einsum("rlbk,rlkj->rlbj", pl_all.view(-1, L, K, 3) - Tf[:, :, None, :3, 3], Tf[:, :, :3, :3])
            return memo
        for i, frame in enumerate(frames):
            st = self.state[frame]
            # (the same holders, already marked, when the core's leaves were built on this manager's state)
            st["pos_on_link_in_base_frame"].assign_lazy(lambda i=i: src.view(i, 0), owner=src)
            st["pos_on_obstacle_in_base_frame"].assign_lazy(lambda i=i: src.view(i, 1), owner=src)
            sl = slice(i * K, (i + 1) * K)
            st["distance"].assign_lazy(lambda sl=sl: derived()["dist"][0, sl] if single else derived()["dist"][:, sl])
            st["normal_vec"].assign_lazy(lambda sl=sl: derived()["nvec"][0, sl] if single else derived()["nvec"][:, sl])
            st["relative_position"].assign_lazy(lambda i=i: derived()["rel"][0, i] if single else derived()["rel"][:, i])
