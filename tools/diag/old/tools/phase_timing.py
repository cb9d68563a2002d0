"""Kernel-time breakdown by RMP-set ablation (GPU).  python tools/phase_timing.py [R]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, descriptor as D
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.urdf import panda_table
from riemannian_motion_policies_amd.rmp import _null_table

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = panda_table()
tgt = D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"), Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3)
jla = D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, Cf.JOINT_LIMIT_PARAMS, vec_a=Cf.PANDA_Q_LOW, vec_b=Cf.PANDA_Q_HIGH)
damp = D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, Cf.JOINT_DAMPING_PARAMS)
sets = {
    "D: no frames, damping only (load+LU+store)": (_null_table(9), [damp]),
    "C: panda walk, damping only": (t, [damp]),
    "B: + target attractor": (t, [tgt, damp]),
    "A: + joint limit (config2)": (t, [tgt, jla, damp]),
    "config3 (32 spheres)": None,
}
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
out = torch.empty_like(q)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
for name, spec in sets.items():
    if spec is None:
        _, desc = Cf.config3()
    else:
        desc = D.build_desc(spec[0], spec[1])
    eng = Engine(desc, 0)
    obs = eng.obstacles(spheres=sph) if spec is None else None
    g = goal if desc.goal_floats else None
    for _ in range(20):
        eng.step(q, qd, g, obstacles=obs, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); eng.step(q, qd, g, obstacles=obs, out=out); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    print(f"R={R:6d} {name:48s} median {ts[50]:8.1f} us   min {ts[0]:8.1f} us")
