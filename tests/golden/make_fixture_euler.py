"""Generate tests/golden/euler.npz: (x, xd, J, c) of the chain [FK(frame), TaskmapFrom4x4ToEuler] (SURVEY 8(a) row
a12; taskmap.py:57-67, kinematics.py:74-96) for three Panda frames, by the nested-autograd restatement
(oracle/torch_autodiff_oracle.py).  16 seeded Panda states (default_rng(12)), re-drawn until |cos(theta_y)| >= 0.2 for
every frame (the Jacobian of Euler angles grows like 1/cos(theta_y); near gimbal lock it pins nothing).

    python tests/golden/make_fixture_euler.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch_autodiff_oracle as TA  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402

FRAMES = (3, 8, 11)
R = 16


def main():
    gold = json.load(open(os.path.join(HERE, "kinematic_tables.json")))
    fk = TA.UrdfForwardKinematicTorch(gold["panda"])
    rng = np.random.default_rng(12)
    maps = {fr: TA.chain_taskmaps([TA.TaskmapByForwardKinematic(fk, fk.frame_names[fr]), TA.TaskmapFrom4x4ToEuler()])
            for fr in FRAMES}
    qs, qds, out = [], [], {fr: [] for fr in FRAMES}
    while len(qs) < R:
        s = Cf.sample_panda_states(rng, 1)
        res = {fr: maps[fr].differentiate(torch.tensor(s["q"]), torch.tensor(s["qd"])) for fr in FRAMES}
        if min(abs(np.cos(float(res[fr][0][0, 1]))) for fr in FRAMES) < 0.2:
            continue
        qs.append(s["q"][0])
        qds.append(s["qd"][0])
        for fr in FRAMES:
            out[fr].append([t.numpy()[0] for t in res[fr]])
    data = {"q": np.stack(qs), "qd": np.stack(qds), "frames": np.array(FRAMES)}
    for fr in FRAMES:
        for k, name in enumerate(("x", "xd", "J", "c")):
            data[f"f{fr}_{name}"] = np.stack([o[k] for o in out[fr]])
    np.savez_compressed(os.path.join(HERE, "euler.npz"), **data)
    print({k: v.shape for k, v in data.items()})


if __name__ == "__main__":
    main()
