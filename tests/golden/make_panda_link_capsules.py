"""Generator of riemannian_motion_policies_amd/robots/panda_link_capsules.json -- DATA, like kinematic_tables.json: one
enclosing capsule per Panda link, fitted to the VERTEX SET of the link's collision mesh (the reference's
urdf/franka_panda/meshes/collision/*.obj, placed by the <collision><origin> of urdf/franka_panda/panda.urdf), in LINK
coordinates.  The reference asks PyBullet for the closest points on these meshes (simulation.py:462-484); the GPU stage works on
capsules (include/rmp2.h rmp2_obstacles.link_capsules), so the capsule must CONTAIN the mesh: distances are then never
over-estimated, and the error is bounded by the capsule's excess over the mesh, which this script reports per link
(max / mean distance from the capsule surface to the nearest mesh vertex direction-wise is not needed: `slack` = capsule radius
minus the mesh's largest radial extent around the same axis at the caps).

Fit: minimum-VOLUME enclosing capsule by direct search over the axis (direction + offset; the radius and the two end points
follow in closed form from the vertices), started from the principal axis of the vertex set.

Run in the build container (needs /root/reference; the output is committed):  python tests/golden/make_panda_link_capsules.py
"""
import json
import os
import sys
from xml.etree import ElementTree

import numpy as np
from scipy.optimize import minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_URDF = "/root/reference/urdf/franka_panda/panda.urdf"


def read_obj_vertices(path):
    v = [list(map(float, line.split()[1:4])) for line in open(path) if line.startswith("v ")]
    return np.asarray(v, dtype=np.float64)


def capsule_for_axis(P, c, d):
    """Smallest capsule with axis through c along unit d that contains the points P: radius R = max radial distance; the end
    points as far inside as the points allow (a point at axial t and radial rho needs the segment to reach within
    sqrt(R^2 - rho^2) of t)."""
    t = (P - c) @ d
    rad = np.linalg.norm((P - c) - np.outer(t, d), axis=1)
    R = rad.max()
    reach = np.sqrt(np.maximum(R * R - rad * rad, 0.0))
    lo, hi = (t + reach).min(), (t - reach).max()
    if lo > hi:
        lo = hi = 0.5 * (lo + hi)
    return R, c + lo * d, c + hi * d


def volume(R, a, b):
    return np.pi * R * R * np.linalg.norm(b - a) + 4.0 / 3.0 * np.pi * R ** 3


def fit_capsule(P):
    c0 = P.mean(axis=0)
    _, _, Vt = np.linalg.svd(P - c0, full_matrices=False)
    best = None
    for k in range(3):                         # every principal direction as a start (a flat link's best axis is not its longest)
        d0 = Vt[k]
        # parametrise: direction = normalise(d0 + u e1 + v e2), offset = c0 + s e1 + w e2  (e1, e2 span the plane normal to d0)
        e1 = np.cross(d0, [1.0, 0.0, 0.0] if abs(d0[0]) < 0.9 else [0.0, 1.0, 0.0])
        e1 /= np.linalg.norm(e1)
        e2 = np.cross(d0, e1)

        def cost(x):
            d = d0 + x[0] * e1 + x[1] * e2
            d /= np.linalg.norm(d)
            c = c0 + x[2] * e1 + x[3] * e2
            R, a, b = capsule_for_axis(P, c, d)
            return volume(R, a, b)
        res = minimize(cost, np.zeros(4), method="Nelder-Mead", options=dict(xatol=1e-7, fatol=1e-12, maxiter=4000))
        d = d0 + res.x[0] * e1 + res.x[1] * e2
        d /= np.linalg.norm(d)
        c = c0 + res.x[2] * e1 + res.x[3] * e2
        R, a, b = capsule_for_axis(P, c, d)
        if best is None or volume(R, a, b) < best[0]:
            best = (volume(R, a, b), R, a, b)
    return best[1], best[2], best[3]


def main():
    from riemannian_motion_policies_amd.urdf import rotation_from_rpy_reference_order, _floats
    root = ElementTree.parse(REF_URDF).getroot()
    base = os.path.dirname(REF_URDF)
    out = {"source": "urdf/franka_panda/panda.urdf collision meshes of the reference (vertex sets); generator: "
                     "tests/golden/make_panda_link_capsules.py",
           "convention": "capsule = segment a-b with radius r, in LINK coordinates (= the frame of the joint that moves the link)",
           "links": {}}
    for link in root.findall("link"):
        col = link.find("collision")
        if col is None:
            continue
        mesh = col.find("geometry").find("mesh")
        if mesh is None:
            continue
        path = os.path.join(base, mesh.attrib["filename"].replace("package://", ""))
        V = read_obj_vertices(path)
        origin = col.find("origin")
        xyz = np.asarray(_floats(origin.attrib.get("xyz") if origin is not None else None), dtype=np.float64)
        Rc = rotation_from_rpy_reference_order(_floats(origin.attrib.get("rpy") if origin is not None else None)).astype(np.float64)
        P = V @ Rc.T + xyz                      # mesh vertices in link coordinates
        R, a, b = fit_capsule(P)
        # containment check and the capsule's excess: distance of every vertex to the axis segment
        ab = b - a
        L2 = float(ab @ ab)
        t = np.clip(((P - a) @ ab) / L2, 0.0, 1.0) if L2 > 0 else np.zeros(len(P))
        dist = np.linalg.norm(P - (a + np.outer(t, ab)), axis=1)
        assert dist.max() <= R * (1 + 1e-9), (link.attrib["name"], dist.max(), R)
        # the mesh's own extent: bounding-box diagonal, for scale
        ext = P.max(axis=0) - P.min(axis=0)
        out["links"][link.attrib["name"]] = {
            "a": [round(float(x), 6) for x in a], "b": [round(float(x), 6) for x in b], "r": round(float(R) + 5e-7, 6),
            "n_vertices": int(len(P)), "mesh_extent": [round(float(x), 4) for x in ext],
            "mean_vertex_depth": round(float((R - dist).mean()), 4)}
        print(f"{link.attrib['name']:18s} r = {R:.4f}  length = {np.linalg.norm(ab):.4f}  vertices {len(P):4d}  "
              f"extent {ext.round(3)}  mean depth of the vertices under the surface {float((R - dist).mean()):.4f}")
    dst = os.path.join(ROOT, "riemannian_motion_policies_amd", "robots", "panda_link_capsules.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
