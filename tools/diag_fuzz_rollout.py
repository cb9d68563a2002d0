#!/usr/bin/env python3
"""Replay one fuzz seed: the plain step against the first control step of the fused rollout, robot by robot
(test infrastructure; python tools/diag_fuzz_rollout.py SEED)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F
import oracle as O
import torch
from riemannian_motion_policies_amd import descriptor as D
from riemannian_motion_policies_amd.engine import Engine

seed = int(sys.argv[1])
c = F.draw_case(seed)
t, specs, solve, kernel, R, n, desc, q, qd, goal, kw, eng_kw = (c[k] for k in ("table", "specs", "solve", "kernel", "R", "n", "desc", "q", "qd", "goal", "kw", "eng_kw"))
print(c["robot_kind"], n, "dof", solve, kernel or "default", R, c["obs_label"], [(s.kind, s.taskmap, s.frame) for s in specs])
if kernel: os.environ["RMP2_KERNEL"] = str(kernel)
eng = Engine(desc, 0)
os.environ.pop("RMP2_KERNEL", None)
dev = {k: (torch.from_numpy(np.ascontiguousarray(v)) if k != "pair_counts" else v) for k, v in eng_kw.items()}
obstacles = eng.obstacles(**dev) if eng_kw else None
tq, tqd = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
tg = None if goal is None else torch.from_numpy(goal).cuda()
st = torch.zeros(R, dtype=torch.int32, device="cuda")
got = eng.step(tq, tqd, tg, obstacles=obstacles, status=st).cpu().numpy().astype(np.float64)
print("step:", eng.last_kernel())
st2 = torch.zeros(R, dtype=torch.int32, device="cuda")
first = eng.rollout(tq.clone(), tqd.clone(), tg, obstacles=obstacles, n_control_steps=1, substeps=1, dt=0.004, status=st2).cpu().numpy().astype(np.float64)
print("rollout:", eng.last_kernel())
ref = O.step(desc, q, qd, goal, **kw)
np.set_printoptions(linewidth=220, precision=4)
d = np.abs(first - got).max(axis=1) / np.maximum(1.0, np.abs(got).max(axis=1))
e_step = np.abs(got - ref["qdd64"]).max(axis=1) / np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))
dsel = np.where(e_step <= 1e-5, np.nan_to_num(d), -1.0)
worst = np.argsort(-dsel)[:3]
print('north-star robots', int((e_step <= 1e-5).sum()), 'of', R)
for r in worst:
    sv = np.linalg.svd(ref["M"][r], compute_uv=False)
    print("robot", r, "rollout-vs-step", d[r], "step-vs-oracle", e_step[r], "status step", int(st[r]), "rollout", int(st2[r]))
    print("   sv_rel", sv / sv[0]); print("   M diag", np.diag(ref["M"][r])); print("   f", ref["f"][r])
    print("   step   ", got[r]); print("   rollout", first[r]); print("   oracle ", ref["qdd64"][r])
