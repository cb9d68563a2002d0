"""Sweep the robot -> lane mappings (RMP2_KERNEL = hex | quad | lane) over fleet sizes for one RMP set.
usage: dispatch_sweep.py <tj5|config3r|config3|config2|exp05|exp05tj> [R ...]     (tj5 = TwoJoint half of config 5, ragged lists;
exp05 / exp05tj = attached-point leaves, TaskmapRelative4x4 + CollisionAvoidance, on the Panda / the TwoJoint, 4 pairs per leaf)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
which = sys.argv[1] if len(sys.argv) > 1 else "tj5"
sizes = [int(x) for x in sys.argv[2:]] or [4096, 8192, 12288, 16384, 20480, 24576, 32768, 49152, 65536]
dev = torch.device("cuda", 0)
def timed(launch, n=200):
    for _ in range(30): launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"{which}: us per step;  rows = RMP2_KERNEL, columns = robots")
print("      " + " ".join(f"{R:>8d}" for R in sizes))
for kern in ("hex", "quad", "lane", "auto"):
    if kern == "auto": os.environ.pop("RMP2_KERNEL", None)
    else: os.environ["RMP2_KERNEL"] = kern
    row = []
    for R in sizes:
        try:
            if which in ("exp05", "exp05tj"):
                from riemannian_motion_policies_amd import descriptor as D
                _, desc = Cf.exp05_panda() if which == "exp05" else Cf.exp05_two_joint()
                s = (Cf.sample_panda_states if which == "exp05" else Cf.sample_two_joint_states)(np.random.default_rng(1), R)
                sph = None
            elif which == "tj5":
                _, desc = Cf.config5_two_joint()
                s = Cf.sample_two_joint_states(np.random.default_rng(1), R)
                sph = Cf.sample_spheres(np.random.default_rng(7)); sph[:, :2] *= 2.0; sph[:, 2] = 0.1
            else:
                _, desc = (Cf.config2 if which == "config2" else Cf.config3)()
                s = Cf.sample_panda_states(np.random.default_rng(1), R)
                sph = Cf.sample_spheres(np.random.default_rng(7))
            eng = Engine(desc, 0)
            q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
            spt = torch.from_numpy(sph).to(dev) if sph is not None else None
            if which in ("exp05", "exp05tj"):
                rel, nv, dist = Cf.sample_point_pairs(np.random.default_rng(2), R, len(D.distance_leaf_indices(desc)), 4)
                obs = eng.obstacles(p_link=torch.from_numpy(rel).to(dev), p_obs=torch.from_numpy(nv).to(dev), dist=torch.from_numpy(dist).to(dev))
            elif which == "config2": obs = None
            elif which == "config3": obs = eng.obstacles(spheres=spt)
            else:
                off, idx = Cf.sample_ragged(np.random.default_rng(3), R)
                obs = eng.obstacles(spheres=spt, csr_offset=torch.from_numpy(off), csr_index=torch.from_numpy(idx))
            launch, _ = eng.bind(q, qd, goal, obstacles=obs)
            row.append(timed(launch))
            del eng, launch
        except Exception as e:
            row.append(float("nan"))
    print(f"{kern:>5s} " + " ".join(f"{t:8.1f}" for t in row), flush=True)
