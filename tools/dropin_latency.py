"""Wall-clock cost of ONE control step through the reference's class surface at R = 1 -- the way the reference's scripts call it
(experiments/franka_panda/06_cluttered_environment.py:120-131: calculate_distances -> Datamanager.update -> RmpCore.evaluate ->
.numpy()), for the experiment-06 policy set with 8 control-point frames x K obstacles.
  host tuples   Datamanager.update(q, tuples) with host tuples as PyBullet delivers them, evaluate(q, qd) on numpy arrays
  device        Datamanager.update_device(core, q, spheres) + evaluate on device tensors, result copied back per step
usage: python tools/dropin_latency.py [K] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "compat"))
import data_management  # noqa: E402
import kinematics  # noqa: E402
import rmp  # noqa: E402
import rmp2  # noqa: E402
import taskmap  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf, urdf  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 7          # exp. 06 has 7 cylinders
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300

fkine = kinematics.UrdfForwardKinematic(urdf_filepath=urdf.PANDA_URDF, order=urdf.PANDA_ORDER)
dm = data_management.Datamanager(fkine)
core = rmp.RmpCore()
ee = taskmap.chain_taskmaps([taskmap.TaskmapByForwardKinematic(fkine, frame='panda_grasptarget_hand'), taskmap.TaskmapFrom4x4ToPosition()])
target = rmp2.TargetAttractor(goal=[0.2, -0.2, 0.5], accel_p_gain=0.3, accel_d_gain=0.6, accel_norm_eps=0.075,
                              metric_alpha_length_scale=0.05, min_metric_alpha=0.03, max_metric_scalar=1, min_metric_scalar=0.5,
                              proximity_metric_boost_scalar=1., proximity_metric_boost_length_scale=0.02, taskmap=ee, name='attractor')
core.add_rmp(target)
core.add_rmp(rmp2.JointVelocityCap(max_velocity=0.5, velocity_damping_region=0.15, damping_gain=5.0, metric_weight=0.05))
core.add_rmp(rmp2.JointDamping(accel_d_gain=1, metric_scalar=0.005, inertia=0.3))
core.add_rmp(rmp2.CSpaceBiasing(goal=Cf.CSPACE_BIASING_GOAL, metric_scalar=0.005, position_gain=1, damping_gain=2,
                                robust_position_term_thresh=0.5, inertia=0.0001))
for frame in Cf.CONTROL_POINT_FRAMES:
    tm = taskmap.chain_taskmaps([taskmap.TaskmapByForwardKinematic(fkine, frame),
                                 taskmap.TaskmapJointFrame4x4ToDistance(dm[frame]['pos_on_link_in_base_frame'],
                                                                        dm[frame]['pos_on_obstacle_in_base_frame'])])
    core.add_rmp(rmp2.ObstacleAvoidance(margin=0., damping_gain=50, damping_std_dev=0.04, damping_robustness_eps=0.01,
                                        damping_velocity_gate_length_scale=0.01, repulsion_gain=800, repulsion_std_dev=0.01,
                                        metric_modulation_radius=0.5, metric_scalar=1, metric_exploder_std_dev=0.02,
                                        metric_exploder_eps=0.001, taskmap=tm, name=f'collision_avoidance_for_{frame}'))

rng = np.random.default_rng(3)
s = Cf.sample_panda_states(rng, 1)
q, qd = s["q"][0], s["qd"][0]
spheres = Cf.sample_spheres(rng, K)
spheres[:, 2] += np.float32(0.6)
# the tuples PyBullet would deliver: (frame, p_link, p_obs, normal, distance, description) per frame and obstacle
org = np.stack([fkine.forward(q[None], fr)[0, :3, 3] for fr in Cf.CONTROL_POINT_FRAMES])
tuples = []
for c, fr in enumerate(Cf.CONTROL_POINT_FRAMES):
    for b in range(K):
        diff = org[c] - spheres[b, :3]
        d = np.linalg.norm(diff)
        tuples.append((fr, org[c], spheres[b, :3] + spheres[b, 3] * diff / d, diff / d, d - spheres[b, 3], ""))


def loop(body):
    for _ in range(20):
        body()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        body()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


def host_body():
    dm.update(q, tuples)
    return core.evaluate(q, qd).numpy()


def host_eval_only():
    return core.evaluate(q, qd).numpy()


dev = torch.device("cuda", 0)
qt, qdt, spt = torch.from_numpy(q).to(dev), torch.from_numpy(qd).to(dev), torch.from_numpy(spheres).to(dev)


def device_body():
    dm.update_device(core, qt, spt)
    return core.evaluate(qt, qdt).cpu().numpy()


def device_eval_only():
    return core.evaluate(qt, qdt).cpu().numpy()


ref = host_body()
us_host = loop(host_body)
us_host_eval = loop(host_eval_only)
got = device_body()
us_dev = loop(device_body)
us_dev_eval = loop(device_eval_only)
print(f"experiment-06 set, R = 1, {len(Cf.CONTROL_POINT_FRAMES)} frames x {K} obstacles = {len(tuples)} pairs; {steps} steps each")
print(f"  host tuples : Datamanager.update + evaluate + .numpy()   {us_host:8.1f} us per control step = {1e6 / us_host:7.0f} steps/s   (evaluate alone {us_host_eval:7.1f} us)")
print(f"  device      : update_device + evaluate + .cpu()          {us_dev:8.1f} us per control step = {1e6 / us_dev:7.0f} steps/s   (evaluate alone {us_dev_eval:7.1f} us)")
print(f"  max |qdd_device - qdd_host| = {np.abs(got - ref).max():.2e}")

# ---- the same loop body for a FLEET, device-resident (GPU time per control step, CUDA events) ----
Rf = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
sf = Cf.sample_panda_states(np.random.default_rng(4), Rf)
qf, qdf = torch.from_numpy(sf["q"]).to(dev), torch.from_numpy(sf["qd"]).to(dev)
target.goal = torch.from_numpy(sf["goal"]).to(dev)
lc = torch.from_numpy(urdf.link_capsules(urdf.PANDA_URDF, fkine.table, Cf.CONTROL_POINT_FRAMES)).to(dev)


def fleet_loop(body, n=20):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        body()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def fleet_body():
    dm.update_device(core, qf, spt, link_capsules=lc)
    return core.evaluate(qf, qdf)


def fleet_body_all_fields():
    dm.update_device(core, qf, spt, link_capsules=lc)
    out = core.evaluate(qf, qdf)
    for fr in Cf.CONTROL_POINT_FRAMES:
        dm[fr]["distance"].value, dm[fr]["normal_vec"].value, dm[fr]["relative_position"].value
    return out


us_f = fleet_loop(fleet_body)
us_fa = fleet_loop(fleet_body_all_fields)
print(f"fleet of {Rf} robots, device-resident, link capsules, {K} obstacles: update_device + evaluate {us_f:8.1f} us per control step "
      f"= {Rf / us_f * 1e-3:6.3f} G robot steps/s;  with the three derived Datamanager fields read every step {us_fa:8.1f} us")
