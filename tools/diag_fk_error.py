"""Diagnostic: position error of the engine's forward kinematics against the fp64 kinematics, next to the fp32 C oracle's, on the
config-3 perf states (what feeds the distance x = |p - c| - r, whose absolute error near-contact accelerations amplify by 1 / rstd)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from riemannian_motion_policies_amd import configs as Cf, descriptor as D
from riemannian_motion_policies_amd.engine import Engine
_, desc = Cf.config3()
s = Cf.sample_panda_states(np.random.default_rng(1), 65536)
q = s["q"][:8192]
eng = Engine(desc, 0)
Tg = eng.forward_kinematics(torch.from_numpy(q)).cpu().numpy().astype(np.float64)
T64 = O.forward_kinematics(desc, q, precision="f64")
T32 = O.forward_kinematics(desc, q, precision="f32").astype(np.float64)
frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
for name, T in (("engine", Tg), ("C oracle f32", T32)):
    e = np.abs(T[:, frames][:, :, :3, 3] - T64[:, frames][:, :, :3, 3]).max(axis=2)      # [R, frames]
    print(f"{name:14s} position error of the 8 control-point frames: median {np.median(e):.2e} p90 {np.percentile(e, 90):.2e} p99 {np.percentile(e, 99):.2e} max {e.max():.2e}  per frame p90 {[float(f'{x:.2e}') for x in np.percentile(e, 90, axis=0)]}")
