"""Diagnostic: per-phase shader-clock stamps of the quad kernel (build with -DRMP2_STAMPS into
tools/diag/librmp2_stamps.so, run with RMP2_LIB pointing at it).  Prints median cycles per phase."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RMP2_LIB", os.path.join(ROOT, "tools", "diag", "librmp2_stamps.so"))
from riemannian_motion_policies_amd import configs as Cf, descriptor as D
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.urdf import panda_table
from riemannian_motion_policies_amd.rmp import _null_table

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = panda_table()
damp = D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, Cf.JOINT_DAMPING_PARAMS)
sets = {"D no-frames damping": D.build_desc(_null_table(9), [damp]), "C walk damping": D.build_desc(t, [damp]),
        "config2": Cf.config2()[1], "config3": Cf.config3()[1], "config3 link geometry": Cf.config3()[1],
        "config3 explicit pairs": Cf.config3()[1]}
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
names = ["prologue", "phase1+walk", "fk leaves", "identity leaves", "LU", "store"]
for name, desc in sets.items():
    eng = Engine(desc, 0)
    obs = eng.obstacles(spheres=sph) if name == "config3" else None
    if name == "config3 link geometry":   # the links' capsules against the same table, closest points formed inside the step
        from riemannian_motion_policies_amd import urdf as U
        lc = torch.from_numpy(U.link_capsules(U.PANDA_URDF, t, Cf.CONTROL_POINT_FRAMES)).cuda()
        obs = eng.obstacles(spheres=sph, link_capsules=lc)
    if name == "config3 explicit pairs":   # interface B: the pairs of the same table as [R, 256, 3] arrays
        pl_, po_ = eng.closest_points(q, eng.obstacles(spheres=sph))
        obs = eng.obstacles(p_link=pl_, p_obs=po_)
    g = goal if desc.goal_floats else None
    n_blocks = (R + 3) // 4 if os.environ.get("RMP2_HEX_WAVES") == "1" else (R + 15) // 16
    buf = torch.zeros(R * 9 + 16 * n_blocks, dtype=torch.float64, device="cuda")  # f rows, then 16 stamps per block
    f = buf[: R * 9].view(R, 9)
    for _ in range(5):
        eng.step(q, qd, g, obstacles=obs, f=f)
    torch.cuda.synchronize()
    per_block = 16  # quad: 16 robots per wave-block; hex: 4 waves x 4 robots per block
    if os.environ.get("RMP2_KERNEL") == "hex" and os.environ.get("RMP2_HEX_WAVES") == "1":
        per_block = 4
    st = buf.cpu().numpy().view(np.uint64)[R * 9 : R * 9 + ((R + per_block - 1) // per_block) * 16].reshape(-1, 16).astype(np.int64)
    seg = np.median(st[:, 8:13], axis=0)
    d = np.diff(st[:, :7], axis=1)
    med = np.median(d, axis=0)
    tot = np.median(st[:, 6] - st[:, 0])
    print(f"R={R} {name:22s} total {tot:8.0f} cyc | " + " ".join(f"{n}={m:.0f}" for n, m in zip(names, med)))
    if st[:, 13].any():  # quad kernel, culled sphere mode: pair-loop trips of the wave and in-range pairs of its first robot
        print(" " * 12 + f"pair loop: {st[:, 13].mean():.1f} trips per wave-step (max {st[:, 13].max()}), "
              f"{st[:, 14].mean():.1f} in-range pairs per robot-step (of 240)")
    if seg.any():  # quad kernel only: the FK-leaf loop by segment, summed over the frames
        print(" " * 12 + "fk leaves by segment: " + " ".join(f"{n}={m:.0f}" for n, m in zip(
            ["fetch(frame,leaf head)", "target leaf / quad sums", "cull+pair trips", "my columns", "pull-back"], seg)))
