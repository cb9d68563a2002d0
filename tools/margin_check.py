"""Tolerance margin of the table modes against the golden config-3 accelerations: the plain table mode and the link path with
zero-length capsules (the route RmpCore.evaluate takes after update_distances), worst relative error against 1e-5."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, urdf as U
from riemannian_motion_policies_amd.engine import Engine
g = np.load(os.path.join(ROOT, "tests", "golden", "config3.npz"))
table, desc = Cf.config3()
eng = Engine(desc, 0)
q, qd, goal = (torch.from_numpy(g[k]).cuda() for k in ("q", "qd", "goal"))
sp = torch.from_numpy(g["spheres"]).cuda()
mag = np.maximum(1.0, np.abs(g["qdd"]).max(axis=1))
for name, obs in (("plain table", eng.obstacles(spheres=sp)),
                  ("link path, zero-length capsules", eng.obstacles(spheres=sp, link_capsules=torch.zeros(8, 8, device="cuda")))):
    out = eng.step(q, qd, goal, obstacles=obs).cpu().numpy()
    err = np.abs(out - g["qdd"]).max(axis=1) / mag
    print(f"{name:34s}: worst relative error vs golden {err.max():.2e} (tolerance 1e-5), median {np.median(err):.2e}")
