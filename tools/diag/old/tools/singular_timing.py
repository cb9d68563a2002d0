"""Kernel time of a rank-deficient set (Panda, target attractor only: rank 3 of 9) in AUTO vs PINV mode."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, descriptor as D
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.urdf import panda_table
t = panda_table()
tgt = D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"), Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
for solve in ("auto", "pinv"):
    eng = Engine(D.build_desc(t, [tgt], solve), 0)
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    for _ in range(3):
        eng.step(q, qd, goal, status=st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        eng.step(q, qd, goal, status=st)
    b.record(); torch.cuda.synchronize()
    print(f"R={R} target-only solve={solve}: {a.elapsed_time(b) / 10 * 1e3:.1f} us/step; status bits: pinv_path={int(((st & 4) != 0).sum())} rank_drop={int(((st & 2) != 0).sum())}")
