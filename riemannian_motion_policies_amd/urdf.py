"""URDF -> flat kinematic table ("kinematic compiler", host side, setup time only).

This is the setup stage of the hot path: it turns a URDF into the constant tables the
HIP kernels index.  It restates the *semantics* of the reference's setup code

  * helper/urdf_parsing.py:57-97   (find base link, breadth-first attach joints, ids in
                                    creation order)
  * helper/urdf_parsing.py:134-147 (backward paths root -> element, by joint name)
  * kinematics.py:163-209          (frame names, padded chains, q re-ordering, T_constant
                                    from rpy/xyz with R = R_x(roll) @ R_y(pitch) @ R_z(yaw)
                                    -- reference quirk Q7 --, axis, one-hot joint types)

but emits a parent-index tree + depth-first schedule instead of padded chain tables,
because the kernels walk the tree once per robot and never re-multiply a chain.

Nothing here touches the GPU; the table is plain numpy and is serialised into the C
descriptor by `descriptor.py`.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Sequence
from xml.etree import ElementTree

import numpy as np

JOINT_FIXED = 0
JOINT_REVOLUTE = 1
JOINT_PRISMATIC = 2
_TYPE_CODE = {"fixed": JOINT_FIXED, "revolute": JOINT_REVOLUTE, "prismatic": JOINT_PRISMATIC}

_ROBOT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "robots")
PANDA_URDF = os.path.join(_ROBOT_DIR, "panda_kinematics.urdf")
TWO_JOINT_URDF = os.path.join(_ROBOT_DIR, "two_joint_kinematics.urdf")

# PyBullet motor-joint order of the two reference robots (helper/pybullet_helper.py:8-19
# applied to the reference URDFs; SURVEY section 8 header).
PANDA_ORDER = [f"panda_joint{i}" for i in range(1, 8)] + ["panda_finger_joint1", "panda_finger_joint2"]
TWO_JOINT_ORDER = ["joint_1", "joint_2"]


def _floats(text: str | None, n: int = 3) -> List[float]:
    if text is None:
        return [0.0] * n
    vals = [float(t) for t in text.split()]
    if len(vals) != n:
        raise ValueError(f"expected {n} numbers, got {text!r}")
    return vals


def _rot_x(a: np.float32) -> np.ndarray:
    c, s = np.cos(a, dtype=np.float32), np.sin(a, dtype=np.float32)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float32)


def _rot_y(a: np.float32) -> np.ndarray:
    c, s = np.cos(a, dtype=np.float32), np.sin(a, dtype=np.float32)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32)


def _rot_z(a: np.float32) -> np.ndarray:
    c, s = np.cos(a, dtype=np.float32), np.sin(a, dtype=np.float32)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)


def rotation_from_rpy_reference_order(rpy: Sequence[float]) -> np.ndarray:
    """fp32 R = R_x(roll) @ R_y(pitch) @ R_z(yaw)  (kinematics.py:123-127, quirk Q7).

    The URDF standard is R_z @ R_y @ R_x; the reference multiplies the other way round.
    Both agree whenever at most one angle is non-zero, which holds for every joint of the
    two reference robots.  The reference's order is reproduced on purpose.
    """
    r, p, y = (np.float32(v) for v in rpy)
    return ((_rot_x(r) @ _rot_y(p)).astype(np.float32) @ _rot_z(y)).astype(np.float32)


@dataclass
class _Elem:
    name: str
    link_name: str
    joint_type: str = "fixed"
    rpy: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    xyz: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    axis: List[float] = field(default_factory=lambda: [0.0, 0.0, 0.0])
    has_collision: bool = False
    parent: int = -1  # element id of the parent (0 = root element)


@dataclass
class KinematicTable:
    """Constant tables of one robot type.  Frame index = reference frame order."""

    frame_names: List[str]
    parent: np.ndarray  # int32 [F], parent frame index, -1 = child of the base link
    joint_type: np.ndarray  # int32 [F], JOINT_*
    q_index: np.ndarray  # int32 [F], index into the caller's q vector, -1 = not actuated
    axis: np.ndarray  # float32 [F,3], joint axis in the joint frame
    T_const: np.ndarray  # float32 [F,4,4], parent-link -> joint frame at q = 0
    has_collision: np.ndarray  # bool [F]
    order: List[str]  # caller's joint order (names), len = n_dof
    link_names: List[str]
    limits_lower: np.ndarray  # float32 [F] (nan when the joint has no <limit>)
    limits_upper: np.ndarray  # float32 [F]

    @property
    def n_frames(self) -> int:
        return len(self.frame_names)

    @property
    def n_dof(self) -> int:
        return len(self.order)

    def frame_index(self, name: str) -> int:
        try:
            return self.frame_names.index(name)
        except ValueError:
            raise KeyError(f"unknown frame {name!r}; known: {self.frame_names}") from None

    # -- views that mirror the reference's own tables (used by the golden-table test) ----
    def backward_paths(self) -> List[List[str]]:
        """Root -> frame joint-name paths (helper/urdf_parsing.py:134-147)."""
        paths = []
        for i in range(self.n_frames):
            path, j = [], i
            while j >= 0:
                path.insert(0, self.frame_names[j])
                j = int(self.parent[j])
            paths.append(path)
        return paths

    def q_reordering(self) -> List[int]:
        """kinematics.py:197: index into q, or n_dof for "append 0"."""
        return [int(k) if k >= 0 else self.n_dof for k in self.q_index]

    def ancestor_dof_mask(self, frame: int) -> int:
        """Bit j set  <=>  dof j moves `frame` (its joint is the frame or an ancestor)."""
        mask, j = 0, frame
        while j >= 0:
            if self.q_index[j] >= 0 and self.joint_type[j] != JOINT_FIXED:
                mask |= 1 << int(self.q_index[j])
            j = int(self.parent[j])
        return mask

    def depth_first_schedule(self):
        """Visit order + save/restore slots for a one-pass tree walk.

        Returns (order, restore_slot, save_slot, n_slots): frames are visited in `order`
        (depth-first pre-order, children in reference order).  The walker keeps ONE
        running frame state; `restore_slot[k] >= 0` means "before visiting order[k], reload
        the state saved in that slot" (the parent is not the frame visited just before),
        `restore_slot[k] == -2` means "start from the base", -1 means "continue from the
        previous frame".  `save_slot[k] >= 0` means "after visiting, save the state" (the
        frame has a child that is not visited immediately afterwards).
        """
        F = self.n_frames
        children = [[] for _ in range(F)]
        roots = []
        for i in range(F):
            (roots if self.parent[i] < 0 else children[int(self.parent[i])]).append(i)
        order: List[int] = []
        stack = list(reversed(roots))
        while stack:
            i = stack.pop()
            order.append(i)
            stack.extend(reversed(children[i]))
        pos = {f: k for k, f in enumerate(order)}
        restore = [-1] * F
        save = [-1] * F
        free_at: List[int] = []  # per slot: schedule position after which it is free
        slot_of = {}
        for k, f in enumerate(order):
            p = int(self.parent[f])
            if p < 0:
                restore[k] = -2
            elif k > 0 and order[k - 1] == p:
                restore[k] = -1
            else:
                restore[k] = slot_of[p]
            # does f need saving?  yes if some child is not the next frame in the order
            late = [c for c in children[f] if pos[c] != k + 1]
            if late:
                last_use = max(pos[c] for c in late)
                for s, free in enumerate(free_at):
                    if free < k:
                        free_at[s] = last_use
                        slot_of[f] = s
                        break
                else:
                    free_at.append(last_use)
                    slot_of[f] = len(free_at) - 1
                save[k] = slot_of[f]
        return order, restore, save, len(free_at)


def compile_urdf(urdf_filepath: str, order: Sequence[str]) -> KinematicTable:
    """Parse `urdf_filepath` and build the table for joint order `order`.

    Frame order reproduces the reference exactly: breadth-first over links starting at the
    base link, joints taken in document order (helper/urdf_parsing.py:74-97), frames =
    all non-root elements in creation order (kinematics.py:169-171).
    """
    root = ElementTree.parse(urdf_filepath).getroot()
    links = root.findall("link")
    joints = root.findall("joint")
    if not links or not joints:
        raise ValueError(f"{urdf_filepath}: no <link>/<joint> elements")

    child_links = {j.find("child").attrib["link"] for j in joints}
    base = next((l for l in links if l.attrib["name"] not in child_links), None)
    if base is None:
        raise ValueError("URDF has no base link (every link is the child of a joint)")
    link_by_name = {l.attrib["name"]: l for l in links}

    elems: List[_Elem] = [_Elem(name="<ROOT>", link_name=base.attrib["name"])]
    limits: List[tuple] = [(np.nan, np.nan)]
    todo = [0]
    while todo:
        leaf = todo.pop(0)
        for j in joints:
            if j.find("parent").attrib["link"] != elems[leaf].link_name:
                continue
            jtype = j.attrib["type"]
            if jtype not in _TYPE_CODE:
                raise NotImplementedError(
                    f"joint {j.attrib['name']!r}: type {jtype!r} is not supported "
                    "(the reference handles fixed / revolute / prismatic only, kinematics.py:205-209)")
            origin = j.find("origin")
            axis_el = j.find("axis")
            child = link_by_name[j.find("child").attrib["link"]]
            lim = j.find("limit")
            elems.append(_Elem(
                name=j.attrib["name"],
                link_name=child.attrib["name"],
                joint_type=jtype,
                rpy=_floats(origin.attrib.get("rpy") if origin is not None else None),
                xyz=_floats(origin.attrib.get("xyz") if origin is not None else None),
                axis=(_floats(axis_el.attrib.get("xyz")) if (axis_el is not None and jtype != "fixed")
                      else [0.0, 0.0, 0.0]),
                has_collision=child.find("collision") is not None,
                parent=leaf,
            ))
            limits.append((float(lim.attrib["lower"]), float(lim.attrib["upper"]))
                          if lim is not None and "lower" in lim.attrib else (np.nan, np.nan))
            todo.append(len(elems) - 1)

    frames = elems[1:]
    F = len(frames)
    order = list(order)
    names = [e.name for e in frames]
    for o in order:
        if o not in names:
            raise KeyError(f"joint {o!r} from `order` is not in the URDF")
    T_const = np.zeros((F, 4, 4), dtype=np.float32)
    for i, e in enumerate(frames):
        T_const[i, :3, :3] = rotation_from_rpy_reference_order(e.rpy)
        T_const[i, :3, 3] = np.asarray(e.xyz, dtype=np.float32)
        T_const[i, 3, 3] = 1.0
    q_index = np.array([order.index(e.name) if e.name in order else -1 for e in frames], dtype=np.int32)
    jt = np.array([_TYPE_CODE[e.joint_type] for e in frames], dtype=np.int32)
    for i, e in enumerate(frames):
        if jt[i] != JOINT_FIXED and q_index[i] < 0:
            # reference: q' = gather([q, 0], reorder) -> a movable joint missing from `order`
            # is evaluated at q = 0 (kinematics.py:197,218-219); keep that behaviour.
            pass
    return KinematicTable(
        frame_names=names,
        parent=np.array([e.parent - 1 for e in frames], dtype=np.int32),
        joint_type=jt,
        q_index=q_index,
        axis=np.array([e.axis for e in frames], dtype=np.float32).reshape(F, 3),
        T_const=T_const,
        has_collision=np.array([e.has_collision for e in frames], dtype=bool),
        order=order,
        link_names=[e.link_name for e in frames],
        limits_lower=np.array([l[0] for l in limits[1:]], dtype=np.float32),
        limits_upper=np.array([l[1] for l in limits[1:]], dtype=np.float32),
    )


def fitted_link_capsules(urdf_filepath: str) -> dict:
    """{link name: {"a": [3], "b": [3], "r": float}} of capsules fitted to the collision MESHES of a robot whose kinematics-only
    URDF ships here: `<stem>_link_capsules.json` next to `<stem>_kinematics.urdf` (the Panda's: generated from the vertex sets of
    the reference's collision meshes by tests/golden/make_panda_link_capsules.py -- minimum-volume enclosing capsules, so
    distances to the capsule never exceed distances to the mesh); {} when there is no such file."""
    import json
    stem = os.path.basename(urdf_filepath)
    for suffix in ("_kinematics.urdf", ".urdf"):
        if stem.endswith(suffix):
            stem = stem[: -len(suffix)]
            break
    path = os.path.join(os.path.dirname(urdf_filepath), stem + "_link_capsules.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)["links"]


def link_capsules(urdf_filepath: str, table: KinematicTable, frames: Sequence[str], default_radius: float = 0.06,
                  fitted="auto") -> np.ndarray:
    """Link capsules [len(frames), 8] = (a, radius, b, 0) in FRAME coordinates for the closest-point stage with link
    geometry (Engine.closest_points(link_capsules=), include/rmp2.h rmp2_closest_points_links): the link that moves with
    each frame as a capsule.  The reference asks PyBullet for the closest points on the link's collision SHAPE
    (simulation.py:462-484); here that shape is reduced to a capsule:
      * <cylinder length radius>: the cylinder's axis (local z of the collision origin), shortened by the radius at both ends;
      * <box size>: the box's longest edge as the axis, radius = half the larger of the two other edges, shortened likewise;
      * <sphere radius>: a capsule of zero length;
      * <mesh> / no primitive: the capsule FITTED to the link's collision mesh where one ships with the robot (`fitted`:
        "auto" = fitted_link_capsules(urdf_filepath) -- the Panda's --, a dict of the same shape, or None); otherwise the segment
        from the frame origin to the origin of its first child frame (the next joint), radius `default_radius`."""
    root = ElementTree.parse(urdf_filepath).getroot()
    link_by_name = {l.attrib["name"]: l for l in root.findall("link")}
    fitted = fitted_link_capsules(urdf_filepath) if fitted == "auto" else (fitted or {})
    out = np.zeros((len(frames), 8), dtype=np.float32)
    for i, fr in enumerate(frames):
        f = table.frame_index(fr)
        link = link_by_name.get(table.link_names[f])
        col = link.find("collision") if link is not None else None
        geom = col.find("geometry") if col is not None else None
        origin = col.find("origin") if col is not None else None
        xyz = np.asarray(_floats(origin.attrib.get("xyz") if origin is not None else None), dtype=np.float64)
        Rc = rotation_from_rpy_reference_order(_floats(origin.attrib.get("rpy") if origin is not None else None)).astype(np.float64)
        prim = None
        if geom is not None:
            for tag in ("cylinder", "box", "sphere"):
                if geom.find(tag) is not None:
                    prim = (tag, geom.find(tag))
        if prim is not None and prim[0] == "cylinder":
            r, L = float(prim[1].attrib["radius"]), float(prim[1].attrib["length"])
            half = max(L / 2.0 - r, 0.0)
            axis = Rc @ np.array([0.0, 0.0, 1.0])
        elif prim is not None and prim[0] == "box":
            size = np.asarray(_floats(prim[1].attrib["size"]), dtype=np.float64)
            k = int(np.argmax(size))
            r = float(np.max(np.delete(size, k))) / 2.0
            half = max(size[k] / 2.0 - r, 0.0)
            axis = Rc @ np.eye(3)[k]
        elif prim is not None and prim[0] == "sphere":
            r, half, axis = float(prim[1].attrib["radius"]), 0.0, np.array([0.0, 0.0, 1.0])
        else:
            cap = fitted.get(table.link_names[f])
            if cap is not None:   # a capsule fitted to the link's collision mesh (robots/*_link_capsules.json)
                out[i] = [cap["a"][0], cap["a"][1], cap["a"][2], cap["r"], cap["b"][0], cap["b"][1], cap["b"][2], 0.0]
                continue
            kids = [c for c in range(table.n_frames) if table.parent[c] == f]
            b = table.T_const[kids[0], :3, 3].astype(np.float64) if kids else np.zeros(3)
            out[i] = [0.0, 0.0, 0.0, default_radius, b[0], b[1], b[2], 0.0]
            continue
        a, b = xyz - half * axis, xyz + half * axis
        out[i] = [a[0], a[1], a[2], r, b[0], b[1], b[2], 0.0]
    return out


def panda_table() -> KinematicTable:
    return compile_urdf(PANDA_URDF, PANDA_ORDER)


def two_joint_table() -> KinematicTable:
    return compile_urdf(TWO_JOINT_URDF, TWO_JOINT_ORDER)
