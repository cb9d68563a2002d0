#!/usr/bin/env python3
"""Replay one fuzz seed leaf by leaf: the engine's and the oracle's (M, f) of ONE robot for each leaf of the set alone
(test infrastructure; python tools/diag_fuzz_leaf.py SEED ROBOT)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as F
import oracle as O
import torch
from riemannian_motion_policies_amd import descriptor as D
from riemannian_motion_policies_amd.engine import Engine

seed, robot = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
with tempfile.TemporaryDirectory() as tmp:
    kind, t, lo, hi = F.draw_robot(rng, tmp)
specs, ok = F.draw_specs(rng, t, lo, hi)
solve = str(rng.choice(["auto", "pinv"], p=[0.6, 0.4])); kernel = rng.choice(["", "hex", "quad", "lane"], p=[0.4, 0.2, 0.25, 0.15]); R = int(rng.choice(F.FLEET_SIZES))
desc = D.build_desc(t, specs, solve); n = t.n_dof; span = hi - lo
q = rng.uniform(lo + 0.05 * span, hi - 0.05 * span, (R, n)).astype(np.float32); qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
if rng.random() < 0.3: qd *= 5.0
goal = rng.uniform(-0.8, 0.8, (R, desc.goal_floats)).astype(np.float32) if desc.goal_floats else None
kw, lab = F.draw_obstacles(rng, O, desc, q, ok)
print(kind, n, "dof", solve, kernel, R, lab)
np.set_printoptions(linewidth=220, precision=3)
dl = D.distance_leaf_indices(desc)
sl = slice(robot, robot + 1)
subsets = [[int(x) for x in a.split(",")] for a in sys.argv[3:]]
for sub in subsets:
    sps = [specs[i] for i in sub]
    dS = D.build_desc(t, sps, solve)
    gS = None
    if dS.goal_floats:
        gS = np.concatenate([goal[:, desc.leaves[i].goal_offset:desc.leaves[i].goal_offset + specs[i].goal_len] for i in sub if specs[i].goal_len], axis=1)
    dlS = [i for i in sub if specs[i].taskmap in (2, 3)]
    kS = dict(kw) if dlS else {}
    if kernel: os.environ["RMP2_KERNEL"] = str(kernel)
    eng = Engine(dS, 0)
    os.environ.pop("RMP2_KERNEL", None)
    dev = {k: (torch.from_numpy(np.ascontiguousarray(v)) if k != "pair_counts" else v) for k, v in kS.items()}
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda"); f = torch.empty((R, n), dtype=torch.float64, device="cuda")
    eng.step(torch.from_numpy(q), torch.from_numpy(qd), None if gS is None else torch.from_numpy(np.ascontiguousarray(gS)), obstacles=eng.obstacles(**dev) if kS else None, M=M, f=f)
    torch.cuda.synchronize()
    ref = O.step(dS, q, qd, gS, **kS)
    print("subset", sub, eng.last_kernel()[:60]); print("  M engine row1", M[robot].cpu().numpy()[1]); print("  M oracle row1", ref["M"][robot][1]); print("  M engine diag", np.diag(M[robot].cpu().numpy())); print("  M oracle diag", np.diag(ref["M"][robot])); print("  f engine", f[robot].cpu().numpy()); print("  f oracle", ref["f"][robot])
for i, sp in enumerate(specs if not subsets else []):
    d1 = D.build_desc(t, [sp], solve)
    k1 = {}
    if sp.taskmap in (2, 3) and kw:
        k1 = dict(kw)
        if "pair_counts" in kw:      # this leaf's pairs only
            j = dl.index(i); b = int(np.sum(kw["pair_counts"][:j])); c = kw["pair_counts"][j]
            k1 = {k: (v[:, b:b + c] if k in ("p_link", "p_obs", "dist") else v) for k, v in kw.items()}
            k1["pair_counts"] = [c]
    g1 = None
    if sp.goal_len:
        off = desc.leaves[i].goal_offset
        g1 = goal[:, off:off + sp.goal_len]
    if kernel: os.environ["RMP2_KERNEL"] = str(kernel)
    eng = Engine(d1, 0)
    os.environ.pop("RMP2_KERNEL", None)
    dev = {k: (torch.from_numpy(np.ascontiguousarray(v)) if k != "pair_counts" else v) for k, v in k1.items()}
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda"); f = torch.empty((R, n), dtype=torch.float64, device="cuda")
    try:
        eng.step(torch.from_numpy(q), torch.from_numpy(qd), None if g1 is None else torch.from_numpy(np.ascontiguousarray(g1)),
                 obstacles=eng.obstacles(**dev) if k1 else None, M=M, f=f)
    except Exception as e:
        print("leaf", i, (sp.kind, sp.taskmap, sp.frame), "engine:", e); continue
    torch.cuda.synchronize()
    ref = O.step(d1, q[sl], qd[sl], None if g1 is None else g1[sl], **({k: (v[sl] if k in ("p_link", "p_obs", "dist") else v) for k, v in k1.items()} if "p_link" in k1 else
                                                               ({**k1, "csr_offset": np.array([0, k1["csr_offset"][robot + 1] - k1["csr_offset"][robot]], np.int32),
                                                                 "csr_index": k1["csr_index"][k1["csr_offset"][robot]:k1["csr_offset"][robot + 1]]} if "csr_offset" in k1 else k1)))
    print("leaf", i, (sp.kind, sp.taskmap, sp.frame), eng.last_kernel()[:40])
    print("   M diag engine", np.diag(M[robot].cpu().numpy())); print("   M diag oracle", np.diag(ref["M"][0]))
    print("   f engine", f[robot].cpu().numpy()); print("   f oracle", ref["f"][0])
    if sp.taskmap == 2 and "spheres" in k1:
        T = O.forward_kinematics(d1, q[sl], "f64"); p = T[0, sp.frame, :3, 3]
        tab = k1["spheres"]
        idx = k1["csr_index"][k1["csr_offset"][robot]:k1["csr_offset"][robot + 1]] if "csr_offset" in k1 else np.arange(len(tab))
        if tab.shape[1] == 8:
            a, b, rad = tab[idx, 0:3].astype(np.float64), tab[idx, 4:7].astype(np.float64), tab[idx, 3].astype(np.float64)
            ab = b - a; tt = np.clip(np.einsum("kc,kc->k", p - a, ab) / np.maximum(np.einsum("kc,kc->k", ab, ab), 1e-30), 0, 1)
            dist = np.linalg.norm(p - (a + tt[:, None] * ab), axis=1) - rad
        else:
            dist = np.linalg.norm(p - tab[idx, :3], axis=1) - tab[idx, 3]
        print("   list", idx.tolist(), "distances", np.sort(dist)[:6], "params r?", sp.params)
