"""Link / obstacle geometry the GPU closest-point code works on, checked on the CPU:
 * robots/panda_link_capsules.json (capsules fitted to the Panda's collision meshes; generator
   tests/golden/make_panda_link_capsules.py): sane, picked up by urdf.link_capsules, and -- where the reference's meshes are at
   hand (this container) -- containing every mesh vertex;
 * the reference's cylinder obstacles (simulation.py:245-261: flat caps) as capsules: the error of that approximation is
   confined to the cap region and bounded by (sqrt(2) - 1) r, attained at the rim."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fitted_panda_capsules_are_used_and_sane():
    from riemannian_motion_policies_amd import configs as Cf, urdf as U
    data = json.load(open(os.path.join(ROOT, "riemannian_motion_policies_amd", "robots", "panda_link_capsules.json")))["links"]
    assert set(data) >= {f"panda_link{i}" for i in range(8)} | {"panda_hand", "panda_leftfinger", "panda_rightfinger"}
    for name, c in data.items():
        length = float(np.linalg.norm(np.subtract(c["b"], c["a"])))
        assert 0.01 <= c["r"] <= 0.12 and length <= 0.3, (name, c)
    t = U.panda_table()
    lc = U.link_capsules(U.PANDA_URDF, t, Cf.CONTROL_POINT_FRAMES)
    stand_in = U.link_capsules(U.PANDA_URDF, t, Cf.CONTROL_POINT_FRAMES, fitted=None)
    assert lc.shape == (8, 8) and not np.allclose(lc, stand_in)
    for i, fr in enumerate(Cf.CONTROL_POINT_FRAMES):
        c = data[t.link_names[t.frame_index(fr)]]
        assert np.allclose(lc[i], [*c["a"], c["r"], *c["b"], 0.0], atol=1e-6)
    assert np.allclose(stand_in[:, 3], 0.06)          # the default-radius stand-in is still what `fitted=None` gives


@pytest.mark.skipif(not os.path.exists("/root/reference/urdf/franka_panda/panda.urdf"), reason="the reference's meshes are not at hand")
def test_fitted_panda_capsules_contain_the_collision_meshes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_panda_link_capsules.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from xml.etree import ElementTree
    from riemannian_motion_policies_amd.urdf import rotation_from_rpy_reference_order, _floats
    data = json.load(open(os.path.join(ROOT, "riemannian_motion_policies_amd", "robots", "panda_link_capsules.json")))["links"]
    root = ElementTree.parse(mk.REF_URDF).getroot()
    seen = 0
    for link in root.findall("link"):
        col = link.find("collision")
        mesh = col.find("geometry").find("mesh") if col is not None else None
        if mesh is None:
            continue
        V = mk.read_obj_vertices(os.path.join(os.path.dirname(mk.REF_URDF), mesh.attrib["filename"].replace("package://", "")))
        origin = col.find("origin")
        xyz = np.asarray(_floats(origin.attrib.get("xyz") if origin is not None else None), dtype=np.float64)
        Rc = rotation_from_rpy_reference_order(_floats(origin.attrib.get("rpy") if origin is not None else None)).astype(np.float64)
        P = V @ Rc.T + xyz
        c = data[link.attrib["name"]]
        a, b = np.asarray(c["a"]), np.asarray(c["b"])
        ab = b - a
        t = np.clip(((P - a) @ ab) / max(float(ab @ ab), 1e-30), 0.0, 1.0)
        dist = np.linalg.norm(P - (a + np.outer(t, ab)), axis=1)
        assert dist.max() <= c["r"] + 1e-6, (link.attrib["name"], dist.max(), c["r"])     # contains every vertex (hence the hull)
        assert dist.max() >= c["r"] - 1e-3, "the capsule is tight: some vertex touches its surface"
        seen += 1
    assert seen == 11


def _dist_cylinder(p, r, h):
    """Exact distance of points p [n, 3] to a solid flat-capped cylinder of radius r and height h, axis z, centred at 0."""
    rho = np.hypot(p[:, 0], p[:, 1])
    dr, dz = rho - r, np.abs(p[:, 2]) - 0.5 * h
    outside = np.hypot(np.maximum(dr, 0.0), np.maximum(dz, 0.0))
    return np.where((dr <= 0) & (dz <= 0), np.maximum(dr, dz), outside)


def _dist_capsule(p, r, half):
    z = np.clip(p[:, 2], -half, half)
    return np.linalg.norm(p - np.stack([np.zeros_like(z), np.zeros_like(z), z], axis=1), axis=1) - r


@pytest.mark.parametrize("r,h", [(0.05, 0.6), (0.1, 0.3), (0.15, 0.5)])
def test_cylinder_as_capsule_error_is_confined_to_the_caps(r, h):
    """The reference's obstacles are cylinders with FLAT caps (simulation.py:245-261; 06_cluttered_environment.py:39-52); the
    GPU primitives are spheres and capsules.  A cylinder enters as the capsule INSCRIBED in it (same radius, axis shortened by r
    at both ends -- urdf.link_capsules does the same for cylinder links): the capsule lies inside the cylinder, so the distance is
    never under-estimated, it is EXACT for every point whose nearest cylinder point lies on the lateral surface between the two
    shortened ends, and elsewhere the over-estimate is at most (sqrt(2) - 1) r -- reached on the diagonal through the rim."""
    rng = np.random.default_rng(3)
    p = rng.uniform(-1.0, 1.0, size=(200000, 3)) * [3 * r + 0.3, 3 * r + 0.3, 0.5 * h + 3 * r + 0.3]
    d_cyl, d_cap = _dist_cylinder(p, r, h), _dist_capsule(p, r, 0.5 * h - r)
    outside = d_cyl > 0
    err = (d_cap - d_cyl)[outside]
    assert err.min() >= -1e-12
    assert err.max() <= (np.sqrt(2.0) - 1.0) * r + 1e-12
    beside = outside & (np.abs(p[:, 2]) <= 0.5 * h - r)           # abeam of the shortened axis: lateral surface, exact
    assert np.abs((d_cap - d_cyl)[beside]).max() <= 1e-12
    # the bound is attained: a point far out on the 45-degree diagonal through the rim
    q = np.array([[r + 10.0, 0.0, 0.5 * h + 10.0]])
    assert abs((_dist_capsule(q, r, 0.5 * h - r) - _dist_cylinder(q, r, h))[0] - (np.sqrt(2.0) - 1.0) * r) < 1e-3 * r
    # in units of the leaf: ObstacleAvoidance's length scales are 0.01 m (repulsion) and 0.2 m (metric radius): for the
    # reference's exp-06 cylinders (r = 0.05) the worst over-estimate is 2.1 cm, and only for approaches over a cap's rim
    assert (np.sqrt(2.0) - 1.0) * 0.05 < 0.0208
