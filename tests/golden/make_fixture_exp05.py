"""Generate tests/golden/exp05.npz: the TwoJoint experiment-05 set (SURVEY 8(a) rows a11 + a21:
TaskmapRelative4x4 + CollisionAvoidance, experiments/two_joint_robot/05_obstacle_avoidance.py:44-61) and the same
leaf on the Panda's 8 control-point frames.

Expected outputs come from oracle/torch_autodiff_oracle.py (nested-autograd restatement run one robot at a time,
like the reference).  Inputs: NumPy default_rng(5); 32 robots per set; B = 2 pairs per frame; the Datamanager
fields 'relative_position', 'normal_vec', 'distance' are drawn by configs.sample_point_pairs.  TwoJoint states are
re-drawn until cond(M) <= 100 (as for config 1: no damping leaf in this set).

    python tests/golden/make_fixture_exp05.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle as O  # noqa: E402
import torch_autodiff_oracle as TA  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd import descriptor as D  # noqa: E402

R, B = 32, 2
GOLD = json.load(open(os.path.join(HERE, "kinematic_tables.json")))


def main():
    rng = np.random.default_rng(5)
    out = {}
    for key, (tab, desc), fkkey, sampler in (("tj", Cf.exp05_two_joint(), "two_joint", Cf.sample_two_joint_states),
                                             ("pd", Cf.exp05_panda(), "panda", Cf.sample_panda_states)):
        fk = TA.UrdfForwardKinematicTorch(GOLD[fkkey])
        dl = D.distance_leaf_indices(desc)
        s = sampler(rng, R)
        rel, nv, dist = Cf.sample_point_pairs(rng, R, len(dl), B)
        dist[0, 0] = 1.2          # beyond r = 1.1: spline weight cut to 0 (rmp.py:305)
        dist[1, :] = 1.25         # a robot whose every pair is cut: M comes from the other leaves only
        if key == "tj":
            for _ in range(200):
                Mx = O.step(desc, s["q"], s["qd"], s["goal"], p_link=rel, p_obs=nv, dist=dist, precision="f64")["M"]
                bad = np.array([np.linalg.cond(m) > 100.0 for m in Mx])
                if not bad.any():
                    break
                fresh = sampler(rng, int(bad.sum()))
                for k in s:
                    s[k][bad] = fresh[k]
        leaves = TA.leaves_from_desc(desc, tab.frame_names)
        n = desc.robot.n_dof
        qdd, M, f = np.zeros((R, n)), np.zeros((R, n, n)), np.zeros((R, n))
        for r in range(R):
            pairs = {li: (rel[r, k * B:(k + 1) * B], nv[r, k * B:(k + 1) * B], dist[r, k * B:(k + 1) * B])
                     for k, li in enumerate(dl)}
            qdd[r], M[r], f[r] = TA.evaluate_one(fk, leaves, s["q"][r], s["qd"][r], s["goal"][r], pairs)
        for k, v in s.items():
            out[f"{key}_{k}"] = v
        out.update({f"{key}_rel": rel, f"{key}_nvec": nv, f"{key}_dist": dist, f"{key}_qdd": qdd, f"{key}_M": M,
                    f"{key}_f": f})
        print(key, "max |qdd|", np.abs(qdd).max(), "max cond", max(np.linalg.cond(m) for m in M))
    np.savez_compressed(os.path.join(HERE, "exp05.npz"), **out)


if __name__ == "__main__":
    main()
