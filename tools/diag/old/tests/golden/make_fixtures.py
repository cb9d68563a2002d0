"""Generate the committed golden vectors tests/golden/config*.npz.

The reference itself cannot run here (TensorFlow / PyBullet absent), so the expected outputs
come from oracle/torch_autodiff_oracle.py -- the op-for-op autograd restatement of the
reference, run the way the reference runs (one robot per call, FK re-differentiated per RMP,
explicit closest-point pairs).  Inputs follow SURVEY section 8(d): NumPy default_rng(0),
R = 64 per config; config-3/5 robots are re-drawn until every control point keeps >= 0.05 m
surface distance to every sphere (keeps |qdd| = O(1), where an absolute 1e-5 is meaningful).

    python tests/golden/make_fixtures.py          # ~1-2 min on 8 cores
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

import oracle as O  # noqa: E402
import torch_autodiff_oracle as TA  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd import descriptor as D  # noqa: E402

R = 64
MIN_CLEARANCE = 0.05
GOLD = json.load(open(os.path.join(HERE, "kinematic_tables.json")))


def control_point_origins(desc, table, q):
    """fp64 FK (C oracle, double build) of the distance leaves' frames -> [R, C, 3] fp32 inputs."""
    T = O.forward_kinematics(desc, q, precision="f64")
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    return T[:, frames][:, :, :3, 3].astype(np.float32)


def clearance(origins, spheres):
    d = np.linalg.norm(origins[:, :, None, :].astype(np.float64) - spheres[None, None, :, :3], axis=-1)
    return (d - spheres[None, None, :, 3]).min(axis=(1, 2))


def redraw_until_clear(rng, sampler, desc, table, spheres, n):
    s = sampler(rng, n)
    for _ in range(200):
        bad = clearance(control_point_origins(desc, table, s["q"]), spheres) < MIN_CLEARANCE
        if not bad.any():
            return s
        fresh = sampler(rng, int(bad.sum()))
        for k in s:
            s[k][bad] = fresh[k]
    raise RuntimeError("could not find collision-free states")


def run_torch(fk, desc, table, s, pairs_fn=None):
    leaves = TA.leaves_from_desc(desc, table.frame_names)
    n = desc.robot.n_dof
    qdd, M, f = np.zeros((len(s["q"]), n)), np.zeros((len(s["q"]), n, n)), np.zeros((len(s["q"]), n))
    for r in range(len(s["q"])):
        pairs = pairs_fn(r) if pairs_fn else None
        qdd[r], M[r], f[r] = TA.evaluate_one(fk, leaves, s["q"][r], s["qd"][r], s["goal"][r], pairs)
    return qdd, M, f


def main():
    rng = np.random.default_rng(0)
    fk_two = TA.UrdfForwardKinematicTorch(GOLD["two_joint"])
    fk_panda = TA.UrdfForwardKinematicTorch(GOLD["panda"])

    # ---- config 1: TwoJoint target-only; robot 0 is the rank-1 start pose q = [0, 0] -------
    t1, d1 = Cf.config1()
    s = Cf.sample_two_joint_states(rng, R + 1)
    # keep cond(J^T A J) <= 100: near the arm's kinematic singularity sin(q2) = 0 the metric is (nearly)
    # rank deficient and the reference's own fp32 -> fp64 pinv result is rounding noise;
    # the EXACTLY singular start pose is covered separately by robot 0 below.
    for _ in range(200):
        Mx = O.step(d1, s["q"], s["qd"], s["goal"], precision="f64")["M"]
        bad = np.array([np.linalg.cond(m) > 100.0 for m in Mx])
        if not bad.any():
            break
        fresh = Cf.sample_two_joint_states(rng, int(bad.sum()))
        for k in s:
            s[k][bad] = fresh[k]
    s["q"][0] = 0.0
    s["qd"][0] = 0.0
    s["goal"][0] = [1.4, -1.4, 0.1]  # experiments/two_joint_robot/01_target_rmp_only.py:28
    qdd, M, f = run_torch(fk_two, d1, t1, s)
    np.savez_compressed(os.path.join(HERE, "config1.npz"), **s, qdd=qdd, M=M, f=f)
    print("config1", np.abs(qdd).max())

    # ---- config 2: Panda target + joint-limit + damping ---------------------------------
    t2, d2 = Cf.config2()
    s = Cf.sample_panda_states(rng, R)
    # push a few robots into the joint-limit band so the non-symmetric metric is exercised
    s["q"][:8, 3] = np.float32(Cf.PANDA_Q_LOW[3] + 0.1 * rng.uniform(0.2, 0.9, 8) * (Cf.PANDA_Q_HIGH[3] - Cf.PANDA_Q_LOW[3]))
    s["q"][4:12, 5] = np.float32(Cf.PANDA_Q_HIGH[5] - 0.1 * rng.uniform(0.2, 0.9, 8) * (Cf.PANDA_Q_HIGH[5] - Cf.PANDA_Q_LOW[5]))
    qdd, M, f = run_torch(fk_panda, d2, t2, s)
    # sub-step vectors: FK of all frames and differentiate() of three frames (autodiff)
    T = np.stack([np.stack([fk_panda.forward(torch.tensor(s["q"][r:r + 1]), fr)[0].numpy()
                            for fr in fk_panda.frame_names]) for r in range(16)])
    diff = {}
    for fr in (3, 9, 11):
        outs = [fk_panda.differentiate(torch.tensor(s["q"][r:r + 1]), torch.tensor(s["qd"][r:r + 1]),
                                       fk_panda.frame_names[fr]) for r in range(16)]
        for k, name in enumerate(("x", "xd", "J", "c")):
            diff[f"diff{fr}_{name}"] = np.concatenate([o[k].numpy() for o in outs])
    np.savez_compressed(os.path.join(HERE, "config2.npz"), **s, qdd=qdd, M=M, f=f, fk_T=T, **diff)
    print("config2", np.abs(qdd).max())

    # ---- config 3: Panda cluttered, 8 control points x 32 spheres ----------------------------
    t3, d3 = Cf.config3()
    spheres = Cf.sample_spheres(rng)
    s = redraw_until_clear(rng, Cf.sample_panda_states, d3, t3, spheres, R)
    origins = control_point_origins(d3, t3, s["q"])
    p_link, p_obs = Cf.pairs_from_spheres(origins, spheres)
    dl = D.distance_leaf_indices(d3)
    K = spheres.shape[0]

    def pairs3(r):
        return {li: (p_link[r, k * K:(k + 1) * K], p_obs[r, k * K:(k + 1) * K]) for k, li in enumerate(dl)}
    qdd, M, f = run_torch(fk_panda, d3, t3, s, pairs3)
    np.savez_compressed(os.path.join(HERE, "config3.npz"), **s, spheres=spheres, origins=origins, qdd=qdd, M=M, f=f)
    print("config3", np.abs(qdd).max(), "clearance", clearance(origins, spheres).min())

    # ---- config 5: mixed fleet, ragged obstacle lists (32 TwoJoint + 32 Panda) ---------------
    out = {}
    for key, (tab, desc), fk, sampler in (("tj", Cf.config5_two_joint(), fk_two, Cf.sample_two_joint_states),
                                          ("pd", Cf.config3(), fk_panda, Cf.sample_panda_states)):
        sph = Cf.sample_spheres(rng)
        if key == "tj":  # the planar arm lives at z ~ 0.1, radius <= 2: spread the spheres there
            sph[:, :2] *= 2.0
            sph[:, 2] = 0.1 + 0.3 * rng.uniform(-1, 1, len(sph)).astype(np.float32)
        st = redraw_until_clear(rng, sampler, desc, tab, sph, 32)
        if key == "tj":  # no damping leaf in this set: keep cond(M) <= 100 as for config 1
            for _ in range(200):
                Mx = O.step(desc, st["q"], st["qd"], st["goal"], spheres=sph, precision="f64")["M"]
                bad = np.array([np.linalg.cond(m) > 100.0 for m in Mx])
                if not bad.any():
                    break
                fresh = redraw_until_clear(rng, sampler, desc, tab, sph, int(bad.sum()))
                for k2 in st:
                    st[k2][bad] = fresh[k2]
        off, idx = Cf.sample_ragged(rng, 32, len(sph))
        lists = [idx[off[r]:off[r + 1]] for r in range(32)]
        lists[0] = lists[0][:0]        # k_r = 0 edge case: robot 0 sees no obstacle
        lists[1] = np.arange(len(sph), dtype=np.int32)  # k_r = K edge case: robot 1 sees all
        off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int32)
        idx = np.concatenate(lists).astype(np.int32)
        orig = control_point_origins(desc, tab, st["q"])
        dli = D.distance_leaf_indices(desc)

        def pairs5(r, orig=orig, sph=sph, off=off, idx=idx, dli=dli):
            sel = sph[idx[off[r]:off[r + 1]]]
            pl, po = Cf.pairs_from_spheres(orig[r:r + 1], sel) if len(sel) else (np.zeros((1, 0, 3)), np.zeros((1, 0, 3)))
            k = len(sel)
            return {li: (pl[0, c * k:(c + 1) * k], po[0, c * k:(c + 1) * k]) for c, li in enumerate(dli)}
        qdd, M, f = run_torch(fk, desc, tab, st, pairs5)
        for k2, v in st.items():
            out[f"{key}_{k2}"] = v
        out.update({f"{key}_spheres": sph, f"{key}_origins": orig, f"{key}_csr_offset": off, f"{key}_csr_index": idx,
                    f"{key}_qdd": qdd, f"{key}_M": M, f"{key}_f": f})
        print("config5", key, np.abs(qdd).max())
    np.savez_compressed(os.path.join(HERE, "config5.npz"), **out)


if __name__ == "__main__":
    main()
