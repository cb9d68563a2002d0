"""Generic robots: random kinematic trees (revolute / prismatic / fixed joints, arbitrary axes and
origins, up to two simultaneously open branch points) with random RMP sets, HIP vs the CPU oracle.
Exercises everything the two reference robots do not: non-z axes, multi-axis rpy (quirk Q7 order),
save/restore slots, unactuated movable joints, leaves on several frames."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _write_urdf(path, rng, n_links, branch_prob):
    names, parents = ["base"], {}
    joints = []
    for i in range(1, n_links + 1):
        # parent: mostly the previous link (chain), sometimes an earlier one (branch)
        p = i - 1 if (i == 1 or rng.random() > branch_prob) else int(rng.integers(max(0, i - 4), i))
        jt = rng.choice(["revolute", "prismatic", "fixed"], p=[0.6, 0.2, 0.2])
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        if rng.random() < 0.4:
            axis = np.eye(3)[rng.integers(3)] * rng.choice([-1, 1])
        rpy = rng.uniform(-1.5, 1.5, 3) * (rng.random(3) < 0.5)
        xyz = rng.uniform(-0.3, 0.3, 3)
        joints.append((f"j{i}", jt, f"l{p}" if p else "base", f"l{i}", rpy, xyz, axis))
    with open(path, "w") as f:
        f.write('<robot name="rnd">\n<link name="base"/>\n')
        for i in range(1, n_links + 1):
            f.write(f'<link name="l{i}"><collision><geometry/></collision></link>\n')
        for name, jt, par, ch, rpy, xyz, axis in joints:
            f.write(f'<joint name="{name}" type="{jt}"><parent link="{par}"/><child link="{ch}"/>'
                    f'<origin rpy="{rpy[0]} {rpy[1]} {rpy[2]}" xyz="{xyz[0]} {xyz[1]} {xyz[2]}"/>'
                    + (f'<axis xyz="{axis[0]} {axis[1]} {axis[2]}"/>' if jt != "fixed" else "") + '</joint>\n')
        f.write('</robot>\n')
    return [j[0] for j in joints if j[1] != "fixed"]


@pytest.mark.parametrize("seed", range(8))
def test_random_tree_robot(tmp_path, seed, hip_lib):
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D, urdf
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(1000 + seed)
    path = str(tmp_path / "rnd.urdf")
    for _ in range(50):  # re-draw until the tree fits the engine's limits (<= 9 dof, <= 2 slots)
        movable = _write_urdf(path, rng, int(rng.integers(3, 13)), branch_prob=0.25)
        order = [m for m in movable if rng.random() < 0.9][:9]      # some movable joints stay unactuated (q = 0)
        if not order:
            continue
        t = urdf.compile_urdf(path, order)
        if t.depth_first_schedule()[3] <= 2:
            break
    n, F = t.n_dof, t.n_frames
    frames = rng.choice(F, size=min(F, 3), replace=False)
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, int(frames[0]),
                        [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02], goal_len=3),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.005, 0.3]),
             D.LeafSpec(D.LEAF_CSPACE_BIASING, D.TASKMAP_IDENTITY, -1, [0.005, 1.0, 2.0, 0.5, 0.0001],
                        vec_a=rng.uniform(-0.5, 0.5, n)),
             D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, int(frames[-1]), [0.1, 0.5, 0.1], goal_len=3)]
    for fr in frames:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, int(fr),
                                [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]))
    R = 300
    q = rng.uniform(-1.0, 1.0, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.5, 0.5, (R, 6)).astype(np.float32)
    sph = np.concatenate([rng.uniform(-1, 1, (5, 3)) + [0, 0, 3.0], rng.uniform(0.05, 0.1, (5, 1))], axis=1).astype(np.float32)
    for solve in ("auto", "pinv"):
        desc = D.build_desc(t, specs, solve)
        eng = Engine(desc, 0)
        Tg = eng.forward_kinematics(torch.from_numpy(q)).cpu().numpy()
        assert np.abs(Tg - O.forward_kinematics(desc, q)).max() < 2e-6
        fr = int(frames[0])
        got = [x.cpu().numpy() for x in eng.differentiate(torch.from_numpy(q[:32]), torch.from_numpy(qd[:32]), fr)]
        want = O.differentiate(desc, q[:32], qd[:32], fr)
        for a, b, nm in zip(got, want, "x xd J c".split()):
            assert np.abs(a - b).max() < 5e-6, (nm, np.abs(a - b).max())
        out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                       obstacles=eng.obstacles(spheres=torch.from_numpy(sph)))
        ref = O.step(desc, q, qd, goal, spheres=sph)
        err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1)
        tol = ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))
        assert (err <= tol).all(), f"seed {seed} {solve}: worst {err.max():.2e}, slots {t.depth_first_schedule()[3]}, n={n}, F={F}"


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
def test_deep_chain_more_frames_than_lanes(tmp_path, kernel, hip_lib):
    """A 30-frame robot (9 actuated joints, the other movable joints held at q = 0, leaves on the deepest frames so that
    nothing is pruned): more frames than the 16 lanes a robot has in the hex kernel (two frames per lane, five
    pointer-jumping rounds), deep save/restore-free chain for the walk kernels."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D, urdf
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(77)
    path = str(tmp_path / "deep.urdf")
    for _ in range(50):
        movable = _write_urdf(path, rng, 30, branch_prob=0.0)
        if len(movable) >= 20:      # fixed leaf-less frames are folded away: >= 20 frames survive, > 16 lanes
            break
    order = movable[::2][:9]
    t = urdf.compile_urdf(path, order)
    n, F = t.n_dof, t.n_frames
    assert F == 30 and n == 9 and len(movable) >= 20
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, F - 1,
                        [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02], goal_len=3),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.005, 0.3]),
             D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, F - 1,
                        [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]),
             D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, F - 9,
                        [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001])]
    desc = D.build_desc(t, specs)
    old = os.environ.get("RMP2_KERNEL")
    os.environ["RMP2_KERNEL"] = kernel
    try:
        eng = Engine(desc, 0)
    finally:
        if old is None:
            del os.environ["RMP2_KERNEL"]
        else:
            os.environ["RMP2_KERNEL"] = old
    R = 130
    q = rng.uniform(-0.6, 0.6, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.5, 0.5, (R, 3)).astype(np.float32)
    sph = np.concatenate([rng.uniform(-1, 1, (6, 3)) + [0, 0, 6.0], rng.uniform(0.05, 0.1, (6, 1))], axis=1).astype(np.float32)
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda")
    out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                   obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), M=M)
    torch.cuda.synchronize()
    ref = O.step(desc, q, qd, goal, spheres=sph)
    scale = np.maximum(1.0, np.abs(ref["M"]).max())
    assert np.abs(M.cpu().numpy() - ref["M"]).max() < 5e-6 * scale
    err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1)
    assert (err <= 2 * ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))).all(), err.max()


def test_twelve_dof_robot(tmp_path, hip_lib):
    """10 .. 16 actuated dofs (the ABI's RMP2_MAX_DOF) run on the hex mapping's N = 16 template at every fleet size,
    with every leaf kind and both resolves (strict pseudo-inverse, attached-point leaves, sets without an inertia leaf)."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import _native, descriptor as D, urdf
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(1234)
    path = str(tmp_path / "twelve.urdf")
    for _ in range(100):
        movable = _write_urdf(path, rng, 18, branch_prob=0.1)
        if len(movable) >= 12:
            t = urdf.compile_urdf(path, movable[:12])
            if t.depth_first_schedule()[3] <= 2:
                break
    n, F = t.n_dof, t.n_frames
    assert n == 12
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, F - 1,
                        [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02], goal_len=3),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.005, 0.3]),
             D.LeafSpec(D.LEAF_JOINT_LIMIT_AVOIDANCE, D.TASKMAP_IDENTITY, -1, [0.3, 1.0], vec_a=np.full(n, -2.0), vec_b=np.full(n, 2.0)),
             D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, [0.5, 0.15, 5.0, 0.05]),
             D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, F - 1,
                        [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]),
             D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, F // 2,
                        [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001])]
    desc = D.build_desc(t, specs)
    eng = Engine(desc, 0)
    sph = np.concatenate([rng.uniform(-1, 1, (6, 3)) + [0, 0, 6.0], rng.uniform(0.05, 0.1, (6, 1))], axis=1).astype(np.float32)
    for R in (5, 9000):          # one block / beyond the hex mapping's usual fleet range
        q = rng.uniform(-1.2, 1.2, (R, n)).astype(np.float32)
        q[: min(R, 3), 0] = 1.95                                   # inside a joint-limit band
        qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
        goal = rng.uniform(-0.5, 0.5, (R, 3)).astype(np.float32)
        M = torch.empty((R, n, n), dtype=torch.float64, device="cuda")
        out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                       obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), M=M)
        torch.cuda.synchronize()
        sub = slice(0, min(R, 400))
        ref = O.step(desc, q[sub], qd[sub], goal[sub], spheres=sph)
        assert np.abs(M.cpu().numpy()[sub] - ref["M"]).max() < 5e-6 * max(1.0, np.abs(ref["M"]).max())
        err = np.abs(out.cpu().numpy()[sub] - ref["qdd64"]).max(axis=1)
        assert (err <= ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))).all(), err.max()
    # ---- what round 1 refused beyond 9 dofs now runs on the hex mapping ------------------------------------------------
    R = 130
    q = rng.uniform(-1.2, 1.2, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.5, 0.5, (R, 3)).astype(np.float32)
    tq, tqd, tg = (torch.from_numpy(x) for x in (q, qd, goal))

    def check(out, ref, what, scale=1.0):
        err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1)
        assert (err <= scale * ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))).all(), f"{what}: {err.max():.3e}"

    # (i) solve = "pinv": the strict pseudo-inverse (rmp.py:153, the reference's only resolve) on every robot
    d_pinv = D.build_desc(t, specs, "pinv")
    e_pinv = Engine(d_pinv, 0)
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    out = e_pinv.step(tq, tqd, tg, obstacles=e_pinv.obstacles(spheres=torch.from_numpy(sph)), status=st)
    torch.cuda.synchronize()
    assert "strict" in e_pinv.last_kernel() and not st.cpu().numpy().any()
    check(out, O.step(d_pinv, q, qd, goal, spheres=sph), "12 dof, solve = pinv")
    # (ii) a set WITHOUT an inertia leaf (identity-map TargetPolicy supplies a full-rank metric, but no leaf the
    # dispatcher counts as inertia): was refused for n_dof > 9
    tp = D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_IDENTITY, -1, [0.1, 1.0, 0.1], goal_len=n)
    d_ni = D.build_desc(t, [specs[0], tp])
    e_ni = Engine(d_ni, 0)
    goal2 = np.concatenate([goal, np.clip(q + rng.uniform(-0.4, 0.4, q.shape), -2, 2).astype(np.float32)], axis=1)
    out = e_ni.step(tq, tqd, torch.from_numpy(goal2))
    torch.cuda.synchronize()
    ref = O.step(d_ni, q, qd, goal2)
    ok = np.array([np.linalg.cond(m) < 100.0 for m in ref["M"]])   # (as for config 1: beyond, fp32 leaves decide the digits)
    assert ok.mean() > 0.5
    err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1)
    assert (err[ok] <= ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))[ok]).all(), err[ok].max()
    if not ok.all():   # the ill-conditioned robots: backward error against the oracle's system / fp32 resolution (oracle.accuracy_gate)
        rest = ~ok
        verdict = O.accuracy_gate(out.cpu().numpy()[rest], {k: ref[k][rest] for k in ("qdd64", "M", "f")},
                                  spread=O.fp32_resolution(d_ni, q[rest], qd[rest], goal2[rest]))
        assert verdict["ok"].all(), f"no inertia leaf, cond >= 100: {O.gate_summary(verdict)}"
    # (iii) attached-point leaves (CollisionAvoidance on [FK, TaskmapRelative4x4, 4x4 -> position]): a Jacobian per pair
    ca = [0.1 * np.e, 0.3, 1.0, 0.3, 1.1, 1e5]
    sp_pt = specs[:2] + [D.LeafSpec(D.LEAF_COLLISION_AVOIDANCE, D.TASKMAP_FK_POINT, fr, ca) for fr in (F - 1, F // 2, 1)]
    d_pt = D.build_desc(t, sp_pt)
    e_pt = Engine(d_pt, 0)
    B = 3
    rel = rng.uniform(-0.15, 0.15, (R, 3 * B, 3)).astype(np.float32)
    nv = rng.normal(size=(R, 3 * B, 3))
    nv = (nv / np.linalg.norm(nv, axis=-1, keepdims=True)).astype(np.float32)
    dist = rng.uniform(0.05, 1.3, (R, 3 * B)).astype(np.float32)
    out = e_pt.step(tq, tqd, tg, obstacles=e_pt.obstacles(p_link=torch.from_numpy(rel), p_obs=torch.from_numpy(nv),
                                                          dist=torch.from_numpy(dist)))
    torch.cuda.synchronize()
    assert "hex" in e_pt.last_kernel()
    check(out, O.step(d_pt, q, qd, goal, p_link=rel, p_obs=nv, dist=dist), "12 dof, attached-point leaves")


@pytest.mark.parametrize("case", ["nine_dof_many_goals", "sixteen_dof"])
def test_descriptor_limits(tmp_path, case, hip_lib):
    """The ABI's limits exercised: RMP2_MAX_LEAVES = 48 leaves, more than 16 goal floats per robot (the hex mapping
    stages at most 16: such sets run on the quad mapping), RMP2_MAX_DOF = 16 actuated dofs."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import descriptor as D, urdf
    from riemannian_motion_policies_amd.engine import Engine
    rng = np.random.default_rng(4321)
    path = str(tmp_path / "lim.urdf")
    want = 16 if case == "sixteen_dof" else 9
    for _ in range(200):
        movable = _write_urdf(path, rng, 24, branch_prob=0.0)
        if len(movable) >= want:
            t = urdf.compile_urdf(path, movable[:want])
            break
    n, F = t.n_dof, t.n_frames
    assert n == want
    att = [0.3, 0.6, 0.075, 0.05, 0.03, 1.0, 0.5, 1.0, 0.02]
    oa = [0.0, 50.0, 0.04, 0.01, 0.01, 800.0, 0.01, 0.5, 1.0, 0.02, 0.001]
    n_goals = 6 if case == "nine_dof_many_goals" else 5
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, F - 1 - 2 * i, att, goal_len=3) for i in range(n_goals)]
    specs += [D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.005, 0.3]),
              D.LeafSpec(D.LEAF_JOINT_VELOCITY_CAP, D.TASKMAP_IDENTITY, -1, [0.5, 0.15, 5.0, 0.05])]
    while len(specs) < D.MAX_LEAVES:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, int(rng.integers(0, F)), oa))
    desc = D.build_desc(t, specs)
    assert desc.n_leaves == 48 and desc.goal_floats == 3 * n_goals
    eng = Engine(desc, 0)
    R = 70
    q = rng.uniform(-1.0, 1.0, (R, n)).astype(np.float32)
    qd = rng.uniform(-0.1, 0.1, (R, n)).astype(np.float32)
    goal = rng.uniform(-0.5, 0.5, (R, 3 * n_goals)).astype(np.float32)
    sph = np.concatenate([rng.uniform(-1, 1, (4, 3)) + [0, 0, 9.0], rng.uniform(0.05, 0.1, (4, 1))], axis=1).astype(np.float32)
    out = eng.step(torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(goal),
                   obstacles=eng.obstacles(spheres=torch.from_numpy(sph)))
    torch.cuda.synchronize()
    ref = O.step(desc, q, qd, goal, spheres=sph)
    err = np.abs(out.cpu().numpy() - ref["qdd64"]).max(axis=1)
    assert (err <= 2 * ATOL * np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))).all(), err.max()
