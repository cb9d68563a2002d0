#!/usr/bin/env python3
"""(x, xd, J, c) of one frame of one robot of a fuzz seed: engine (rmp2_differentiate) against the oracle in fp32 and fp64
(test infrastructure).   python tools/diag_frame_kinematics.py SEED ROBOT FRAME"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F
import oracle as O
import torch
from riemannian_motion_policies_amd.engine import Engine
seed, robot, frame = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
c = F.draw_case(seed)
desc, q, qd = c["desc"], c["q"], c["qd"]
np.set_printoptions(linewidth=200, precision=9)
eng = Engine(desc, 0)
x, xd, J, cc = (t.cpu().numpy().astype(np.float64)[robot] for t in eng.differentiate(torch.from_numpy(q), torch.from_numpy(qd), frame))
x32, xd32, J32, c32 = (a[robot].astype(np.float64) for a in O.differentiate(desc, q, qd, frame, "f32"))
x64, xd64, J64, c64 = (a[robot] for a in O.differentiate(desc, q, qd, frame, "f64"))
pos = [3, 7, 11]
for name, e, a32, a64 in (("p", x, x32, x64), ("v", xd, xd32, xd64), ("a_bias", cc, c32, c64)):
    print(name, "fp64", a64[pos], " engine - fp64", e[pos] - a64[pos], " fp32 oracle - fp64", a32[pos] - a64[pos])
print("J rows (position) engine - fp64\n", J[pos] - J64[pos], "\n fp32 oracle - fp64\n", J32[pos] - J64[pos], "\n J fp64\n", J64[pos])
print("q", q[robot], "qd", qd[robot])
