"""Time the closest-point stage (rmp2_closest_points / rmp2_closest_points_links) on its own and the explicit-pair control step it
feeds: the reference's data flow calculate_distances -> Datamanager -> RmpCore.evaluate (simulation.py:462-484,
data_management.py:16-31, rmp.py:133-155) for a whole fleet.   usage: python tools/closest_stage_timing.py [R] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, urdf as U  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
table, desc = Cf.config3()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
lc = torch.from_numpy(U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES)).to(dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


for prim, tab_np in (("spheres", Cf.sample_spheres(np.random.default_rng(7), Cf.N_SPHERES)),
                     ("capsules", Cf.sample_capsules(np.random.default_rng(7), Cf.N_SPHERES))):
    tab = eng.obstacles(spheres=torch.from_numpy(tab_np).to(dev))
    P = 8 * Cf.N_SPHERES
    out_bytes = R * P * 24
    for name, caps in (("frame origins", None), ("link capsules", lc)):
        pl, po = eng.closest_points(q, tab, link_capsules=caps)
        us = timed(lambda: eng.closest_points(q, tab, link_capsules=caps))
        obst = eng.obstacles(p_link=pl, p_obs=po)
        out = torch.empty_like(q)
        launch, _ = eng.bind(q, qd, goal, obstacles=obst, out=out)
        us_step = timed(launch)
        if caps is not None:   # the same pairs formed inside the step (rmp2_obstacles.link_capsules)
            launch_f, _ = eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(tab_np).to(dev), link_capsules=caps), out=out)
            us_f = timed(launch_f)
            print(f"R={R} {prim:8s} link capsules, fused into the step: {us_f:7.1f} us = {R / us_f * 1e-3:6.3f} G robot steps/s   ({eng.last_kernel()})")
        print(f"R={R} {prim:8s} {name:14s}: stage {us:8.1f} us ({out_bytes / us / 1e6:6.2f} TB/s of pair arrays written)   "
              f"explicit-pair step {us_step:7.1f} us   stage + step {us + us_step:8.1f} us = {R / (us + us_step) * 1e-3:6.3f} G robot steps/s")
