#!/usr/bin/env python3
"""One robot of one fuzz seed against the fp32 envelope (test infrastructure): the engine's q-double-dot, the fp32-leaf and fp64 oracle
values, and the distance from the fp64 value of each of the 17 fp32 draws of oracle.fp32_envelope.   python tools/diag_envelope.py SEED ROBOT"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F
import oracle as O
import torch
from riemannian_motion_policies_amd.engine import Engine

seed, robot = int(sys.argv[1]), int(sys.argv[2])
c = F.draw_case(seed)
desc, q, qd, goal, kw, kernel, R, n = (c[k] for k in ("desc", "q", "qd", "goal", "kw", "kernel", "R", "n"))
np.set_printoptions(linewidth=200, precision=6)
if kernel:
    os.environ["RMP2_KERNEL"] = kernel
eng = Engine(desc, 0)
dev = {k: (torch.from_numpy(np.ascontiguousarray(v)) if k not in ("pair_counts", "primitive") else v) for k, v in c["eng_kw"].items()}
out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), None if goal is None else torch.from_numpy(goal), obstacles=eng.obstacles(**dev) if dev else None)
torch.cuda.synchronize()
got = out.cpu().numpy().astype(np.float64)[robot]
sl = slice(robot, robot + 1)
one = {k: (v[sl] if isinstance(v, np.ndarray) and v.shape[:1] == (R,) else v) for k, v in kw.items()}
if "csr_offset" in kw:
    print("ragged lists: the oracle runs on the whole case")
    one, sl = kw, slice(None)
ref = O.step(desc, q[sl], qd[sl], None if goal is None else goal[sl], **one)["qdd64"]
tru = O.step(desc, q[sl], qd[sl], None if goal is None else goal[sl], precision="f64", **one)["qdd64"]
i = robot if sl == slice(None) else 0
print(c["robot_kind"], n, "dof", c["solve"], kernel or "default", R, c["obs_label"], eng.last_kernel()[:60])
print("engine ", got); print("fp32 or", ref[i]); print("fp64 or", tru[i])
print("|engine - fp64|", np.abs(got - tru[i]).max(), " |fp32 oracle - fp64|", np.abs(ref[i] - tru[i]).max())
for s in range(4):
    env = O.fp32_envelope(desc, q[sl], qd[sl], None if goal is None else goal[sl], seed=s, **one)
    print(f"envelope (17 draws, seed {s})", env[i])
env64 = O.fp32_envelope(desc, q[sl], qd[sl], None if goal is None else goal[sl], samples=128, seed=9, **one)
print("envelope (129 draws)", env64[i])
