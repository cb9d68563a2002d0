"""One-off stress: random tree robots (3..30 links, branches) with random RMP sets, every kernel mapping vs the oracle.
python tools/stress_random_robots.py [n_seeds]"""
import os, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
from riemannian_motion_policies_amd import descriptor as D, urdf
from riemannian_motion_policies_amd.engine import Engine
from test_gpu_random_robots import _write_urdf

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = {}
tmp = tempfile.mkdtemp()
done = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(5000 + seed)
    path = os.path.join(tmp, f"r{seed}.urdf")
    ok = False
    for _ in range(50):
        movable = _write_urdf(path, rng, int(rng.integers(3, 31)), branch_prob=float(rng.choice([0.0, 0.15, 0.3])))
        order = [m for m in movable if rng.random() < 0.8][:9]
        if not order:
            continue
        t = urdf.compile_urdf(path, order)
        if t.depth_first_schedule()[3] <= 2:
            ok = True
            break
    if not ok:
        continue
    n, F = t.n_dof, t.n_frames
    frames = rng.choice(F, size=min(F, int(rng.integers(1, 6))), replace=False)
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, int(frames[0]),
                        [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02], goal_len=3),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.005, 0.3]),
             D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, [0.5, 0.15, 5.0, 0.05]),
             D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, [0.3, 1.0], vec_a=np.full(n, -2.0), vec_b=np.full(n, 2.0))]
    for fr in frames:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, int(fr),
                                [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]))
    R = 70
    q = rng.uniform(-1.5, 1.5, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.5, 0.5, (R, 3)).astype(np.float32)
    sph = np.concatenate([rng.uniform(-1, 1, (7, 3)) + [0, 0, 8.0], rng.uniform(0.05, 0.1, (7, 1))], axis=1).astype(np.float32)
    desc = D.build_desc(t, specs)
    ref = O.step(desc, q, qd, goal, spheres=sph)
    for kern in ("hex", "quad", "lane"):
        os.environ["RMP2_KERNEL"] = kern
        eng = Engine(desc, 0)
        out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                       obstacles=eng.obstacles(spheres=torch.from_numpy(sph)))
        torch.cuda.synchronize()
        err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1) / np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))
        worst[kern] = max(worst.get(kern, 0.0), float(err.max()))
        if err.max() > 2e-5:
            print(f"seed {seed} kernel {kern}: F={F} n={n} rel err {err.max():.2e}  <-- OUT OF TOLERANCE")
    done += 1
print(f"{done} robots; worst scaled error per mapping: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
