"""Drives rmp2_validate (the host-side program compiler, no GPU) with random and broken descriptors.
python tools/fuzz_validate.py [n]   (under the sanitizer build: tools/asan_compile_program.sh)"""
import ctypes as C, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from riemannian_motion_policies_amd import descriptor as D, urdf
from test_gpu_random_robots import _write_urdf

lib = C.CDLL(os.environ.get("RMP2_LIB", os.path.join(ROOT, "riemannian_motion_policies_amd", "librmp2_hip.so")))
lib.rmp2_validate.argtypes = [C.POINTER(D.Desc)]
lib.rmp2_last_error.restype = C.c_char_p
lib.rmp2_last_error.argtypes = [C.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(2024)
tmp = tempfile.mkdtemp()
codes = {}
kinds_id = [(D.LEAF_JOINT_DAMPING, [1.0, 0.005, 0.3]), (D.LEAF_JOINT_VELOCITY_CAP, [0.5, 0.15, 5.0, 0.05]),
            (D.LEAF_CSPACE_BIASING, [0.005, 1.0, 2.0, 0.5, 1e-4]), (D.LEAF_CONFIG_SPACE_BIASING, [0.01, 0.1, 0.05]),
            (D.LEAF_JOINT_LIMIT_AVOIDANCE, [0.3, 1.0])]
for it in range(n):
    path = os.path.join(tmp, "r.urdf")
    links = int(rng.integers(1, 33))
    movable = _write_urdf(path, rng, links, branch_prob=float(rng.choice([0.0, 0.1, 0.3, 0.6])))
    order = [m for m in movable if rng.random() < 0.85][:16]
    if not order:
        continue
    try:
        t = urdf.compile_urdf(path, order)
    except Exception:
        continue
    nd, F = t.n_dof, t.n_frames
    specs = []
    for _ in range(int(rng.integers(0, 47))):
        r = rng.random()
        if r < 0.25:
            specs.append(D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, int(rng.integers(0, F)),
                                    [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02], goal_len=3))
        elif r < 0.6:
            specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, int(rng.integers(0, F)),
                                    [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]))
        elif r < 0.7:
            specs.append(D.LeafSpec(D.LEAF_COLLISION_AVOIDANCE, D.TASKMAP_FK_POINT, int(rng.integers(0, F)),
                                    [0.27, 0.3, 1.0, 0.3, 1.1, 1e5]))
        else:
            k, p = kinds_id[int(rng.integers(0, len(kinds_id)))]
            specs.append(D.LeafSpec(k, D.TASKMAP_IDENTITY, -1, p, vec_a=np.full(nd, -2.0), vec_b=np.full(nd, 2.0)))
    d = D.build_desc(t, specs, solve=str(rng.choice(["auto", "pinv"])))
    # a third of the descriptors are broken on purpose: the compiler must reject them, never read or write out of bounds
    r = rng.random()
    if r < 0.08:
        d.robot.parent[int(rng.integers(0, F))] = int(rng.integers(-3, 40))
    elif r < 0.14:
        d.robot.q_index[int(rng.integers(0, F))] = int(rng.integers(-2, 40))
    elif r < 0.2:
        d.robot.joint_type[int(rng.integers(0, F))] = int(rng.integers(-1, 5))
    elif r < 0.25 and d.n_leaves:
        d.leaves[int(rng.integers(0, d.n_leaves))].frame = int(rng.integers(-5, 60))
    elif r < 0.29 and d.n_leaves:
        d.leaves[int(rng.integers(0, d.n_leaves))].taskmap = int(rng.integers(-1, 6))
    elif r < 0.32:
        d.n_leaves = int(rng.integers(-2, 60))
    elif r < 0.34:
        d.robot.n_frames = int(rng.integers(-2, 40))
    elif r < 0.36:
        d.robot.n_dof = int(rng.integers(-1, 20))
    elif r < 0.38 and d.n_leaves:
        d.leaves[int(rng.integers(0, d.n_leaves))].goal_offset = int(rng.integers(-3, 200))
    rc = lib.rmp2_validate(C.byref(d))
    codes[rc] = codes.get(rc, 0) + 1
    if rc != 0:
        assert lib.rmp2_last_error(None), "an error code without a message"
print(f"{sum(codes.values())} descriptors through rmp2_validate: return codes {dict(sorted(codes.items()))}")
assert codes.get(0, 0) > 0 and len(codes) > 1
