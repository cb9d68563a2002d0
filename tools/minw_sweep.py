"""Sweep the quad kernel's register cap (RMP2_QUAD_MINW = waves per SIMD the build leaves room for) over fleet sizes.
Each (minw, R) runs in THIS process: the cap is read per handle at rmp2_create.
usage: minw_sweep.py <config3|config3r|config3j|config2> [R ...]     (config3j = config 3 + a JointLimitAvoidance leaf: the general,
non-symmetric form of the system)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
which = sys.argv[1] if len(sys.argv) > 1 else "config3"
sizes = [int(x) for x in sys.argv[2:]] or [16384, 24576, 32768, 40960, 49152, 57344, 65536, 81920, 98304, 131072, 196608, 262144]
dev = torch.device("cuda", 0)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).to(dev)
def timed(launch, n=200):
    for _ in range(30): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"{which}: us per step;  rows = RMP2_QUAD_MINW, columns = robots")
print("      " + " ".join(f"{R:>8d}" for R in sizes))
for minw in ("2", "3", "4", "auto"):
    if minw == "auto": os.environ.pop("RMP2_QUAD_MINW", None)
    else: os.environ["RMP2_QUAD_MINW"] = minw
    os.environ["RMP2_KERNEL"] = "quad"
    row = []
    for R in sizes:
        if which == "config3j":
            from riemannian_motion_policies_amd import descriptor as D
            from riemannian_motion_policies_amd.urdf import panda_table
            t = panda_table()
            specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"), Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3),
                     D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, Cf.JOINT_VELOCITY_CAP_PARAMS),
                     D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, Cf.JOINT_DAMPING_PARAMS),
                     D.LeafSpec(D.LEAF_CSPACE_BIASING, D.TASKMAP_IDENTITY, -1, Cf.CSPACE_BIASING_PARAMS, vec_a=Cf.CSPACE_BIASING_GOAL),
                     D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, Cf.JOINT_LIMIT_PARAMS, vec_a=Cf.PANDA_Q_LOW, vec_b=Cf.PANDA_Q_HIGH)]
            for fr in Cf.CONTROL_POINT_FRAMES:
                specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, t.frame_index(fr), Cf.OBSTACLE_AVOIDANCE_PARAMS))
            desc = D.build_desc(t, specs)
        else:
            _, desc = (Cf.config2 if which == "config2" else Cf.config3)()
        eng = Engine(desc, 0)
        s = Cf.sample_panda_states(np.random.default_rng(1), R)
        q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
        if which == "config2":
            obs = None
        elif which == "config3r":
            off, idx = Cf.sample_ragged(np.random.default_rng(3), R)
            obs = eng.obstacles(spheres=sph, csr_offset=torch.from_numpy(off), csr_index=torch.from_numpy(idx))
        else:
            obs = eng.obstacles(spheres=sph)
        launch, _ = eng.bind(q, qd, goal, obstacles=obs)
        row.append(timed(launch))
        del eng, launch
    print(f"{minw:>5s} " + " ".join(f"{t:8.1f}" for t in row), flush=True)
