"""Experiment: does the ORDER of the robots in the fleet arrays change the config-3 step time?
The slowest quad of a wave sets the wave's pair-loop trips (20.0 per wave-step against 13.6 with perfect balance); robots are
independent (rmp.py:133-155), so the fleet owner may keep them in any order.  Orders tried:
  given        as sampled
  sorted       by in-range pair count, contiguous (similar robots share a wave; heavy waves share a workgroup / CU)
  dealt        sorted into waves of 16, then the waves dealt so that every workgroup of W waves holds one wave of each load
               quantile, the quantile rotating with the workgroup index (wave slot <-> SIMD mapping is the hardware's)
  dealt_wg     sorted into workgroups of W waves (similar waves share a workgroup), workgroups dealt round-robin over quantiles
usage: python tools/experiments/load_order.py [R] [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, descriptor as D  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
table, desc = Cf.config3()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
spheres = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7), Cf.N_SPHERES)).to(dev)
q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
radius = 0.5   # metric_modulation_radius of the config-3 ObstacleAvoidance leaves
T = eng.forward_kinematics(q)[:, frames, :3, 3]                                  # [R, 8, 3]
d = torch.linalg.norm(T[:, :, None, :] - spheres[None, None, :, :3], dim=-1) - spheres[None, None, :, 3]
inr = d < radius                                                                 # [R, 8, K]
load = inr.sum(dim=(1, 2))
# trips of a quad for one frame = ceil(in-range / 4); the wave's trips = max over its 16 robots, summed over frames
per_frame = (inr.sum(dim=2) + 3) // 4                                            # [R, 8]
print(f"R={R}: in-range pairs per robot mean {load.float().mean():.1f} sd {load.float().std():.1f} max {load.max().item()}")


def wave_trips(perm):
    pf = per_frame[perm].reshape(-1, 16, per_frame.shape[1])
    return pf.max(dim=1).values.sum(dim=1).float()


def run(name, perm, W=4):
    qq, qqd, gg = q[perm].contiguous(), qd[perm].contiguous(), goal[perm].contiguous()
    out = torch.empty_like(qq)
    launch, _ = eng.bind(qq, qqd, gg, obstacles=eng.obstacles(spheres=spheres), out=out)
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    wt = wave_trips(perm)
    wg = wt.reshape(-1, W).sum(dim=1)
    print(f"{name:10s} {e0.elapsed_time(e1) / steps * 1e3:7.2f} us   trips/wave mean {wt.mean():5.2f} max {wt.max():4.0f}   "
          f"per workgroup of {W}: mean {wg.mean():6.2f} max {wg.max():4.0f}   kernel {eng.last_kernel()}")
    return out, perm


ident = torch.arange(R, device=dev)
# the sort key: the wave's trips are sum over frames of max over robots -> robots that are similar FRAME BY FRAME belong
# together; the scalar total is the first approximation
order = torch.argsort(load, stable=True)
n_waves = R // 16
for W in (4,):
    run("given", ident, W)
    run("sorted", order, W)
    waves = order.reshape(n_waves, 16)                      # wave k = k-th lightest
    n_wg = n_waves // W
    # dealt: workgroup b takes sorted waves b, n_wg + b, 2 n_wg + b, ... (one per quantile), rotated by b
    idx = torch.arange(n_wg, device=dev)[:, None] + n_wg * torch.arange(W, device=dev)[None, :]
    rot = (torch.arange(W, device=dev)[None, :] + torch.arange(n_wg, device=dev)[:, None]) % W
    idx = torch.gather(idx, 1, rot)
    run("dealt", waves[idx.reshape(-1)].reshape(-1), W)
    # dealt_wg: workgroups of similar waves, workgroups interleaved light/heavy
    wgs = waves.reshape(n_wg, W * 16)
    half = n_wg // 2
    inter = torch.stack([torch.arange(half, device=dev), n_wg - 1 - torch.arange(half, device=dev)], dim=1).reshape(-1)
    run("dealt_wg", wgs[inter].reshape(-1), W)
    rnd = torch.randperm(n_wg, device=dev)
    run("wg_shuffle", wgs[rnd].reshape(-1), W)

# ---- the bound of ANY balancing between the robots of a wave: every wave holds 16 copies of ONE robot (no imbalance between its
# quads at all; what remains is the quad's own ceil(pairs of a frame / 4)) ----
rep = torch.arange(R // 16, device=dev).repeat_interleave(16)
run("replicated", rep)
pf_rep = per_frame[rep].reshape(-1, 16, per_frame.shape[1])
print(f"replicated: trips per wave-step = sum over frames of ceil(pairs / 4) of the one robot: mean {pf_rep[:, 0].sum(dim=1).float().mean():.2f}; "
      f"pairs / 4 without the ceil: {load[rep].float().mean() / 4:.2f}")
