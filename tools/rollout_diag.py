"""Where does the fused rollout's time per control step go?  For K control steps from the seed-1 states: us per control step of the
rollout, and the time / status histogram of a PLAIN step at the state the rollout ends in (does the fleet's motion make the step
itself slower: more in-range pairs, robots on the careful path?).   python tools/rollout_diag.py [R]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, descriptor as D  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
_, desc = Cf.config3()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q0, qd0, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
obs = eng.obstacles(spheres=sph)
frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]


def plain(q, qd):
    out = torch.empty_like(q)
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    launch, _ = eng.bind(q, qd, goal, obstacles=obs, out=out)
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        launch()
    e1.record()
    torch.cuda.synchronize()
    eng.step(q, qd, goal, obstacles=obs, status=st)
    torch.cuda.synchronize()
    T = eng.forward_kinematics(q)[:, frames, :3, 3]
    d = torch.linalg.norm(T[:, :, None, :] - sph[None, None, :, :3], dim=-1) - sph[None, None, :, 3]
    inr = (d < 0.5).sum(dim=(1, 2)).float()
    vals, counts = torch.unique(st, return_counts=True)
    return e0.elapsed_time(e1) * 10, inr.mean().item(), dict(zip(vals.tolist(), counts.tolist())), qd.abs().max().item()


us, inr, hist, vmax = plain(q0.clone(), qd0.clone())
print(f"K=0 (seed states): plain step {us:6.2f} us, in-range pairs per robot {inr:5.1f}, status {hist}, max|qd| {vmax:.2f}")
for K in (1, 2, 5, 10, 20, 50):
    ts = []
    for rep in range(5):
        q, qd = q0.clone(), qd0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(q, qd, goal, n_control_steps=K, substeps=10, dt=0.01, obstacles=obs)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    us, inr, hist, vmax = plain(q, qd)
    print(f"K={K:3d}: rollout {np.median(ts):8.1f} us = {np.median(ts) / K:6.2f} us per control step; plain step at the end state {us:6.2f} us, "
          f"in-range pairs per robot {inr:5.1f}, status {hist}, max|qd| {vmax:.2f}, finite {bool(torch.isfinite(q).all())}")
