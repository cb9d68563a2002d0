"""Every kernel mapping (hex: 16 lanes per robot, quad: 4, lane: 1) against the CPU oracle on the same inputs.

The dispatcher picks one mapping by fleet size (csrc/rmp2_hip.hip dispatch_solve), so a default run at a given size
exercises one of them; here each is forced through RMP2_KERNEL (read at rmp2_create) for the BASELINE sets, the
mixed-fleet sets, a padded 7-dof arm and random tree robots.  Tolerance as in test_gpu_parity.py.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
D_STATUS_NONFINITE = 1   # include/rmp2.h RMP2_STATUS_NONFINITE

ATOL = 1e-5


@pytest.fixture(scope="module")
def torch_mod(hip_lib):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _engine_env(desc, **env):
    """Engine created with diagnostic environment knobs (read at rmp2_create) set, restored afterwards."""
    from riemannian_motion_policies_amd.engine import Engine
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Engine(desc, 0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _engine(desc, kernel):
    from riemannian_motion_policies_amd.engine import Engine
    old = os.environ.get("RMP2_KERNEL")
    os.environ["RMP2_KERNEL"] = kernel
    try:
        return Engine(desc, 0)
    finally:
        if old is None:
            del os.environ["RMP2_KERNEL"]
        else:
            os.environ["RMP2_KERNEL"] = old


def _check(got, ref, what, mask=None):
    err = np.abs(np.asarray(got, np.float64) - ref).max(axis=-1)
    tol = ATOL * np.maximum(1.0, np.abs(ref).max(axis=-1))
    if mask is not None:
        err, tol = err[mask], tol[mask]
    assert (err <= tol).all(), f"{what}: worst {err.max():.3e}"


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
@pytest.mark.parametrize("R", [3, 130])
def test_config2_all_mappings(torch_mod, kernel, R):
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    rng = np.random.default_rng(7 + R)
    s = Cf.sample_panda_states(rng, R)
    s["q"][: min(R, 8), 3] = np.float32(Cf.PANDA_Q_LOW[3] + 0.02)          # inside the joint-limit band (quirk Q2)
    _, desc = Cf.config2()
    eng = _engine(desc, kernel)
    n = 9
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda")
    f = torch.empty((R, n), dtype=torch.float64, device="cuda")
    qdd = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), M=M, f=f)
    torch.cuda.synchronize()
    ref = O.step(desc, s["q"], s["qd"], s["goal"])
    assert np.abs(M.cpu().numpy() - ref["M"]).max() < 5e-6 and np.abs(f.cpu().numpy() - ref["f"]).max() < 5e-6
    _check(qdd.cpu().numpy(), ref["qdd64"], f"config2 {kernel} R={R}")


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
@pytest.mark.parametrize("mode", ["spheres", "pairs", "ragged"])
def test_config3_all_mappings_and_obstacle_modes(torch_mod, golden_dir, kernel, mode):
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = _engine(desc, kernel)
    q, qd, goal = (torch.from_numpy(g[k]) for k in ("q", "qd", "goal"))
    R = g["q"].shape[0]
    if mode == "spheres":
        obs = eng.obstacles(spheres=torch.from_numpy(g["spheres"]))
        ref = g["qdd"]
    elif mode == "pairs":
        pl, po = Cf.pairs_from_spheres(g["origins"], g["spheres"])
        obs = eng.obstacles(p_link=torch.from_numpy(pl), p_obs=torch.from_numpy(po))
        ref = g["qdd"]
    else:
        off, idx = Cf.sample_ragged(np.random.default_rng(3), R, len(g["spheres"]))
        obs = eng.obstacles(spheres=torch.from_numpy(g["spheres"]), csr_offset=torch.from_numpy(off),
                            csr_index=torch.from_numpy(idx))
        ref = O.step(desc, g["q"], g["qd"], g["goal"], spheres=g["spheres"], csr_offset=off, csr_index=idx)["qdd64"]
    qdd = eng.step(q, qd, goal, obstacles=obs)
    torch.cuda.synchronize()
    _check(qdd.cpu().numpy(), ref, f"config3 {kernel} {mode}")


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
def test_two_joint_template_and_status_paths(torch_mod, golden_dir, kernel):
    """N = 2 template of every mapping (a TwoJoint set WITH an inertia leaf: sets without one are routed to the
    strict pseudo-inverse kernel whatever RMP2_KERNEL says), ragged sphere lists, and the non-finite status path."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.urdf import two_joint_table
    g = np.load(os.path.join(golden_dir, "config5.npz"))
    t = two_joint_table()
    specs = [D.LeafSpec(D.LEAF_TARGET_POLICY, D.TASKMAP_FK_POSITION, t.frame_index("link_23"), Cf.TARGET_POLICY_PARAMS,
                        goal_len=3, name="target"),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, Cf.JOINT_DAMPING_PARAMS, name="joint_damping")]
    for fr in Cf.TWO_JOINT_CONTROL_POINT_FRAMES:
        specs.append(D.LeafSpec(D.LEAF_OBSTACLE_AVOIDANCE, D.TASKMAP_FK_DISTANCE, t.frame_index(fr),
                                Cf.OBSTACLE_AVOIDANCE_PARAMS, name=f"collision_avoidance_for_{fr}"))
    desc = D.build_desc(t, specs)
    eng = _engine(desc, kernel)
    kw = dict(spheres=g["tj_spheres"], csr_offset=g["tj_csr_offset"], csr_index=g["tj_csr_index"])
    obs = eng.obstacles(**{k: torch.from_numpy(v) for k, v in kw.items()})
    qdd = eng.step(torch.from_numpy(g["tj_q"]), torch.from_numpy(g["tj_qd"]), torch.from_numpy(g["tj_goal"]), obstacles=obs)
    torch.cuda.synchronize()
    ref = O.step(desc, g["tj_q"], g["tj_qd"], g["tj_goal"], **kw)
    _check(qdd.cpu().numpy(), ref["qdd64"], f"two-joint + damping {kernel}")
    # non-finite input: the mapping's careful path must flag it, not hang, and leave the other robots alone
    q = g["tj_q"].copy()
    q[0, 0] = np.nan
    st = torch.zeros(q.shape[0], dtype=torch.int32, device="cuda")
    out = eng.step(torch.from_numpy(q), torch.from_numpy(g["tj_qd"]), torch.from_numpy(g["tj_goal"]), obstacles=obs,
                   status=st)
    torch.cuda.synchronize()
    assert st.cpu().numpy()[0] & 1 and not np.isfinite(out.cpu().numpy()[0]).all()
    _check(out.cpu().numpy()[1:], ref["qdd64"][1:], f"two-joint + damping {kernel}, robots next to a NaN one")


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
@pytest.mark.parametrize("key", ["tj", "tjd", "pdi", "p04"])
def test_exp04_sets_all_mappings(torch_mod, golden_dir, kernel, key):
    """Identity-map TargetPolicy + JointLimitAvoidance (04_driving_into_jointlimits.py:46-52) on the TwoJoint and on the
    Panda, and TargetPolicy(FK) + ConfigurationSpaceBiasing (04_nullspace_control.py:41-52), through every mapping,
    against the autograd vectors.  `tj` (the script's own set) has no inertia leaf: whatever RMP2_KERNEL says it is
    resolved by the strict pseudo-inverse kernel -- the reference's only resolve."""
    torch = torch_mod
    from riemannian_motion_policies_amd import configs as Cf
    g = np.load(os.path.join(golden_dir, "exp04.npz"))
    _, desc = {"tj": lambda: Cf.exp04_two_joint(), "tjd": lambda: Cf.exp04_two_joint(with_damping=True),
               "pdi": Cf.exp04_panda_identity_target, "p04": Cf.panda04_nullspace}[key]()
    eng = _engine(desc, kernel)
    R, n = g[f"{key}_q"].shape
    M = torch.empty((R, n, n), dtype=torch.float64, device="cuda")
    f = torch.empty((R, n), dtype=torch.float64, device="cuda")
    qdd = eng.step(torch.from_numpy(g[f"{key}_q"]), torch.from_numpy(g[f"{key}_qd"]), torch.from_numpy(g[f"{key}_goal"]),
                   M=M, f=f)
    torch.cuda.synchronize()
    if key != "tj":
        want = {"hex": "hex", "quad": "quad", "lane": "one lane"}[kernel]
        assert want in eng.last_kernel(), eng.last_kernel()
    assert np.abs(M.cpu().numpy() - g[f"{key}_M"]).max() < 5e-6 and np.abs(f.cpu().numpy() - g[f"{key}_f"]).max() < 5e-6
    _check(qdd.cpu().numpy(), g[f"{key}_qdd"], f"exp04 {key} {kernel}")


@pytest.mark.parametrize("kernel", ["hex", "quad"])
def test_culling_is_exact_and_hardware_approximations_are_bounded(torch_mod, kernel):
    """Unrestricted performance inputs (robots touching and penetrating spheres included), 4 096 robots:
    (i) the culled sphere mode against the explicit-pair mode, which evaluates every pair: a culled pair has metric exactly
        0 (rmp2.py:191-195), so the two may differ by summation order only;
    (ii) the fast mappings (hardware rcp / rsq / exp2, their own FK operation order) against the fp32 oracle and against the
        lane-per-robot kernel (libm formulas, the oracle's operation order).  The reference algorithm amplifies 1e-7 m of
        control-point position -- the fp32 rounding of FK itself -- by exp(-x / 0.01) * 800: measured on the worst robot of
        this fleet (0.052 m clearance), the whole 1.3e-5 sits in f, M agrees to 4e-7 and cond(M) = 4; the fp64 and fp32
        evaluations of the ORACLE differ by 8e-6 there.  Gates: 1e-5 from 0.08 m clearance, 3e-5 in [0.05, 0.08) m, and
        near contact (< 0.05 m) the fast kernels' error distribution within 4x the accurate kernel's."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    rng = np.random.default_rng(33)
    R = 4096
    s = Cf.sample_panda_states(rng, R)
    sph = Cf.sample_spheres(np.random.default_rng(7))
    _, desc = Cf.config3()
    q, qd, goal = (torch.from_numpy(s[k]) for k in ("q", "qd", "goal"))
    eng = _engine(desc, kernel)
    fast = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(sph))).cpu().numpy()
    T = O.forward_kinematics(desc, s["q"], "f64")
    frames = [desc.leaves[i].frame for i in range(desc.n_leaves) if desc.leaves[i].taskmap == 2]
    org = T[:, frames][:, :, :3, 3]
    pl, po = Cf.pairs_from_spheres(org.astype(np.float32), sph)
    pairs = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=torch.from_numpy(pl), p_obs=torch.from_numpy(po))).cpu().numpy()
    lane = _engine(desc, "lane")
    acc = lane.step(q, qd, goal, obstacles=lane.obstacles(spheres=torch.from_numpy(sph))).cpu().numpy()
    torch.cuda.synchronize()
    ref = O.step(desc, s["q"], s["qd"], s["goal"], spheres=sph, precision="f64")["qdd64"]   # yardstick: the algorithm in fp64
    ref32 = O.step(desc, s["q"], s["qd"], s["goal"], spheres=sph)["qdd64"]                  # the reference's fp32 arithmetic
    fin = np.isfinite(ref).all(axis=1) & np.isfinite(fast).all(axis=1) & np.isfinite(acc).all(axis=1)
    assert fin.mean() > 0.99
    mag = np.maximum(1.0, np.abs(ref).max(axis=1))
    clr = (np.linalg.norm(org[:, :, None, :] - sph[None, None, :, :3], axis=-1) - sph[None, None, :, 3]).min(axis=(1, 2))
    clear, band, near = fin & (clr >= 0.08), fin & (clr >= 0.05) & (clr < 0.08), fin & (clr < 0.05) & (clr > 0.0)
    assert clear.sum() > 500 and band.sum() > 100 and near.sum() > 100
    # (i) culled vs every-pair evaluation
    e_cp = np.abs(fast - pairs).max(axis=1) / mag
    assert (e_cp[clear | band] <= 2 * ATOL).all(), f"{kernel}: culled vs explicit pairs, clear robots {e_cp[clear | band].max():.2e}"
    assert (e_cp[near] <= 1e-3).mean() > 0.98 and np.median(e_cp[near]) <= 1e-5, f"{kernel}: {np.median(e_cp[near]):.2e}"
    # (ii) hardware approximations against the accurate formulas
    e_fast, e_acc = np.abs(fast - ref).max(axis=1) / mag, np.abs(acc - ref).max(axis=1) / mag
    e32 = np.abs(fast - ref32).max(axis=1) / mag
    assert (e32[clear] <= ATOL).all(), f"{kernel}: robots with >= 0.08 m clearance, worst {e32[clear].max():.2e}"
    assert (e32[band] <= 3 * ATOL).all(), f"{kernel}: robots with 0.05 .. 0.08 m clearance, worst {e32[band].max():.2e}"
    assert np.median(e_fast[near]) <= 4 * max(np.median(e_acc[near]), 1e-6), (np.median(e_fast[near]), np.median(e_acc[near]))
    assert np.quantile(e_fast[near], 0.95) <= 4 * max(np.quantile(e_acc[near], 0.95), 1e-5), \
        (np.quantile(e_fast[near], 0.95), np.quantile(e_acc[near], 0.95))
    # (iii) and EVERY robot -- penetrating ones included -- is bounded by oracle.accuracy_gate: none is exempted
    ref_sys = O.step(desc, s["q"], s["qd"], s["goal"], spheres=sph)
    verdict = O.accuracy_gate(fast, ref_sys, spread=O.fp32_resolution(desc, s["q"], s["qd"], s["goal"], spheres=sph))
    assert verdict["ok"].all(), f"{kernel}: {O.gate_summary(verdict)}"


def test_quad_register_caps_agree_bitwise_at_fleet_sizes(torch_mod):
    """BASELINE-size property (no oracle at this size): the throughput quad kernel is built three times -- 256, 168 and 128
    registers = two, three, four waves per SIMD -- and `launch_quad` picks two or three from the fleet size.  The builds
    run the same arithmetic in the same order, so their outputs must be IDENTICAL bit for bit, at a size where the rule
    picks three waves (49 152 robots, one round of three) and at the BASELINE size (65 536, two rounds of two); the
    fused rollout too (priorities are reset every control step)."""
    torch = torch_mod
    from riemannian_motion_policies_amd import configs as Cf
    _, desc = Cf.config3()
    sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
    for R in (49152, 65536):
        s = Cf.sample_panda_states(np.random.default_rng(R), R)
        q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
        outs, stats = {}, {}
        for cap in ("auto", 2, 3, 4):
            eng = _engine_env(desc, RMP2_KERNEL="quad") if cap == "auto" else _engine_env(desc, RMP2_KERNEL="quad", RMP2_QUAD_MINW=cap)
            st = torch.zeros(R, dtype=torch.int32, device="cuda")
            outs[cap] = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=sph), status=st).clone()
            stats[cap] = st.clone()
            assert "quad" in eng.last_kernel()
        torch.cuda.synchronize()
        assert torch.isfinite(outs["auto"]).all()
        for cap in (2, 3, 4):
            assert torch.equal(outs[cap], outs["auto"]), f"R={R}: the {cap}-wave build differs from the dispatched one"
            assert torch.equal(stats[cap], stats["auto"])
    # rollout: 3 control steps fused, two caps
    R = 49152
    s = Cf.sample_panda_states(np.random.default_rng(5), R)
    res = []
    for cap in (2, 3):
        eng = _engine_env(desc, RMP2_KERNEL="quad", RMP2_QUAD_MINW=cap)
        q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
        qdd = eng.rollout(q, qd, goal, obstacles=eng.obstacles(spheres=sph), n_control_steps=3, substeps=4, dt=0.002)
        res.append((q.clone(), qd.clone(), qdd.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_symmetric_form_against_the_general_form_and_its_careful_path(torch_mod):
    """The quad kernel's symmetric form (template flag SYM, sets without a JointLimitAvoidance leaf: the system stays
    block-upper through the identity leaves and the elimination) against the general form (RMP2_QUAD_SYM=0) of the same
    kernel: same arithmetic up to the order of the fp64 eliminations, so the fp32 results agree far inside the parity
    tolerance.  Then the one place where the symmetric form needs the FULL matrix on a rare path -- the careful solver of a
    flagged robot, fed by the mirror step: a Panda with a lone target attractor (rank <= 3 of 9: every robot is flagged)
    stepped through the fused-rollout entry (which keeps such sets on the elimination mappings), against the general
    form bit for bit."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.urdf import panda_table
    _, desc = Cf.config3()
    R = 12288                                   # quad by default dispatch, latency build
    s = Cf.sample_panda_states(np.random.default_rng(3), R)
    sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    outs = []
    for symenv in ("1", "0"):
        eng = _engine_env(desc, RMP2_KERNEL="quad", RMP2_QUAD_SYM=symenv)
        M = torch.zeros((R, 9, 9), dtype=torch.float64, device="cuda")
        outs.append((eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=sph), M=M).clone(), M.clone()))
    torch.cuda.synchronize()
    a, b = outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy()
    scale = np.maximum(1.0, np.abs(b).max(axis=1))
    diff = np.abs(a - b).max(axis=1)
    # unrestricted states: ~2 % of the robots touch or penetrate a sphere (|qdd| ~ 1e3, metric entries spanning ten orders
    # of magnitude); there the ORDER of the fp64 eliminations shows at 1e-5 relative in either form (measured: both forms
    # sit equally far from the oracle, 1e-5 .. 1e-4).  Everywhere else the two forms agree to fp32 rounding.
    calm = scale <= 50.0
    assert calm.mean() > 0.9 and (diff[calm] <= 2e-6 * scale[calm]).all(), diff[calm].max()
    assert (diff <= 1e-3 * scale).all(), (diff / scale).max()
    # the exported metric: the symmetric form's is EXACTLY symmetric (round 4: its upper triangle is the matrix -- inside a
    # diagonal block (i, j) and (j, i) are accumulated separately, equal up to fp32 rounding, and the elimination, which reads
    # M[i][k] from the pivot row below the diagonal blocks, needs them equal); the general form keeps both roundings
    Ms, Mg = outs[0][1], outs[1][1]
    iu = torch.triu_indices(9, 9, device="cuda")
    assert torch.equal(Ms[:, iu[0], iu[1]], Mg[:, iu[0], iu[1]]), "the upper triangles of the two forms' exported metric differ"
    assert torch.equal(Ms, Ms.transpose(1, 2)), "the symmetric form's exported metric is not symmetric"
    assert ((Mg - Mg.transpose(1, 2)).abs().amax(dim=(1, 2)) <= 1e-6 * Mg.abs().amax(dim=(1, 2))).all()
    # (inside the diagonal 4 x 4 blocks M[i][j] and M[j][i] are formed by different lanes, (S c_i) . c_j and (S c_j) . c_i in
    # fp32: symmetric to rounding; the off-diagonal blocks are copies)
    Ms = outs[0][1]
    asym = (Ms - Ms.transpose(1, 2)).abs().amax(dim=(1, 2)) / Ms.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    assert float(asym.max()) < 1e-6, float(asym.max())
    # careful path of the symmetric form
    t = panda_table()
    lone = D.build_desc(t, [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, t.frame_index("panda_grasptarget_hand"),
                                       Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3, name="attractor")])
    R2 = 8704                                   # beyond the hex cut: the rollout entry launches the quad kernel
    s2 = Cf.sample_panda_states(np.random.default_rng(4), R2)
    q2, qd2, g2 = (torch.from_numpy(s2[k]).cuda() for k in ("q", "qd", "goal"))
    res = []
    for symenv in ("1", "0"):
        eng = _engine_env(lone, RMP2_KERNEL="quad", RMP2_QUAD_SYM=symenv)
        st = torch.zeros(R2, dtype=torch.int32, device="cuda")
        qdd = eng.rollout(q2.clone(), qd2.clone(), g2, n_control_steps=1, substeps=1, dt=0.0, status=st)   # dt = 0: q, qd unchanged
        torch.cuda.synchronize()
        assert "quad" in eng.last_kernel()
        res.append((qdd.clone(), st.clone()))
    # a rank-3 system: WHAT the pseudo-inverse returns is pinned elsewhere (golden config 1, exp-04 sets, the strict-pinv
    # tests) and is decided by rounding noise in the six "zero" singular values; the two forms' matrices differ by exactly such
    # noise since round 4 (the symmetric form's diagonal blocks are made exactly symmetric), so what is pinned here is that the
    # careful solver of EITHER form returns a solution of the consistent system (f lies in the range of M) and the same status
    Mx = torch.zeros((R2, 9, 9), dtype=torch.float64, device="cuda")
    fx = torch.zeros((R2, 9), dtype=torch.float64, device="cuda")
    _engine_env(lone, RMP2_KERNEL="lane").step(q2, qd2, g2, M=Mx, f=fx)
    torch.cuda.synchronize()
    for qdd, _ in res:
        x = qdd.double()
        resid = (torch.einsum("rij,rj->ri", Mx, x) - fx).abs().amax(dim=1)
        scale_ = (torch.einsum("rij,rj->ri", Mx.abs(), x.abs()) + fx.abs()).amax(dim=1)
        assert (resid <= 1e-4 * scale_).all(), (resid / scale_).max().item()
    assert torch.equal(res[0][1], res[1][1])
    assert torch.isfinite(res[0][0]).all() and torch.isfinite(res[1][0]).all()
    assert (res[0][1].cpu().numpy() != 0).all(), "every robot of a rank-deficient set must report the pseudo-inverse path"


@pytest.mark.parametrize("kernel", ["hex", "quad"])
@pytest.mark.parametrize("K", [32, 48, 64, 80])
def test_ragged_lists_as_membership_masks_and_their_fallback(torch_mod, golden_dir, kernel, K):
    """Ragged obstacle lists over a table of K <= 64 spheres run as a 32/64-bit membership mask (pair_loop_culled<MEMBER>,
    no list walk in the frame loop); K > 64 keeps the CSR walk; a list that repeats an index cannot be a mask -- the
    reference would count that obstacle twice -- and must take the list walk too.  All against the oracle's list walk."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = _engine(desc, kernel)
    R = g["q"].shape[0]
    rng = np.random.default_rng(100 + K)
    sph = np.concatenate([g["spheres"], Cf.sample_spheres(rng, 64)])[:K]
    sph[32:, :2] *= np.float32(1.6)    # the extra spheres stand further out
    # parity is stated for states with >= 0.05 m of clearance: robots an extra sphere comes closer to
    # carry |qdd| of 1e2 and more, where fp32 rounding alone exceeds the bound
    d = np.linalg.norm(g["origins"][:, :, None, :] - sph[None, None, :, :3], axis=-1) - sph[None, None, :, 3]
    clear = d.min(axis=(1, 2)) >= 0.05
    assert clear.sum() >= R // 2
    off, idx = Cf.sample_ragged(rng, R, K)
    q, qd, goal = (torch.from_numpy(g[k]) for k in ("q", "qd", "goal"))
    for dup in (False, True):
        idx2 = idx.copy()
        if dup:   # robot 5 lists its first obstacle twice (same list length: the last entry is overwritten)
            a, b = off[5], off[6]
            if b - a >= 2:
                idx2[b - 1] = idx2[a]
        obs = eng.obstacles(spheres=torch.from_numpy(sph), csr_offset=torch.from_numpy(off), csr_index=torch.from_numpy(idx2))
        kw = dict(spheres=sph, csr_offset=off, csr_index=idx2)
        ref = O.step(desc, g["q"], g["qd"], g["goal"], **kw)
        qdd = eng.step(q, qd, goal, obstacles=obs)
        torch.cuda.synchronize()
        _check(qdd.cpu().numpy(), ref["qdd64"], f"ragged K={K} {kernel} dup={dup}", mask=clear)
        if (~clear).any():      # ... and the robots an extra sphere comes close to are bounded too (oracle.accuracy_gate)
            nc = ~clear
            verdict = O.accuracy_gate(qdd.cpu().numpy()[nc], {k: ref[k][nc] for k in ("qdd64", "M", "f")},
                                      spread=O.fp32_resolution(desc, g["q"][nc], g["qd"][nc], g["goal"][nc], spheres=sph,
                                                               csr_offset=np.concatenate([[0], np.cumsum(np.diff(off)[nc])]).astype(np.int32),
                                                               csr_index=np.concatenate([idx2[off[r]:off[r + 1]] for r in np.nonzero(nc)[0]]).astype(np.int32)))
            assert verdict["ok"].all(), f"ragged K={K} {kernel} dup={dup}: {O.gate_summary(verdict)}"


def test_two_by_two_closed_form_pseudo_inverse(torch_mod, golden_dir):
    """The quad mapping resolves 2-dof robots by the closed-form 2 x 2 pseudo-inverse (rmp2_solve.h pinv_solve_2x2): full
    rank, exactly rank one (the TwoJoint start pose q = [0, 0], quirk Q3: J = [[0,0],[2,1],[0,0]]) and a NaN state,
    against the oracle's SVD-style pseudo-inverse, with the status bits."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    _, desc = Cf.config1()
    eng = _engine(desc, "quad")
    s = Cf.sample_two_joint_states(np.random.default_rng(21), 300)
    s["q"][0] = 0.0                      # rank-1 metric
    s["q"][1] = [0.3, np.pi]             # folded arm: singular Jacobian again
    s["q"][2, 0] = np.nan
    st = torch.zeros(300, dtype=torch.int32, device="cuda")
    qdd = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), status=st)
    torch.cuda.synchronize()
    assert "quad" in eng.last_kernel()
    got, stc = qdd.cpu().numpy(), st.cpu().numpy()
    ref = O.step(desc, s["q"], s["qd"], s["goal"])
    cond = np.linalg.cond(ref["M"][3:])
    # (a set without an inertia leaf: fp32 rounding of the leaves is amplified by cond(M), so the 1e-5 bound is stated for
    # cond(M) <= 100 as everywhere else -- DESIGN.md section 2; robot 0: exact rank 1, the pseudo-inverse is well defined)
    ok = np.concatenate([[True, False, False], cond < 100])
    _check(got, ref["qdd64"], "2x2 closed form", mask=ok)
    # ... and the ill-conditioned rest is not exempted: backward error against the oracle's system, or the robot's own fp32
    # resolution (oracle.accuracy_gate); the folded arm and the NaN robot are asserted through their status words below
    rest = ~ok
    rest[1:3] = False
    verdict = O.accuracy_gate(got[rest], {k: ref[k][rest] for k in ("qdd64", "M", "f")},
                              spread=O.fp32_resolution(desc, s["q"][rest], s["qd"][rest], s["goal"][rest]))
    assert verdict["ok"].all(), f"2x2 closed form, cond >= 100: {O.gate_summary(verdict)}"
    assert stc[0] & 2 and stc[0] & 4, "rank drop of the start pose must be reported"
    assert stc[2] & 1 and not np.isfinite(got[2]).all()
    assert (stc[3:][cond < 1e4] == 0).all() and (cond < 100).sum() > 100


def test_strict_pseudo_inverse_at_fleet_size_takes_two_kernels(torch_mod):
    """solve = "pinv" (the reference's only resolve, rmp.py:153-154) for a fleet: the quad mapping up to the combined metric and
    force, then rmp2_pinv_kernel for every robot.  Same numbers as the oracle's pseudo-inverse and as the lane-per-robot strict
    kernel (which small fleets and RMP2_KERNEL=lane keep); a rank-deficient set (target only: rank <= 3 of 9) goes the same
    way under solve = "auto"; a robot with a NaN state resolves to NaN, as tf.linalg.pinv of a NaN matrix does."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    R = 5000
    rng = np.random.default_rng(31)
    s = Cf.sample_panda_states(rng, R)
    sph = Cf.sample_spheres(rng)
    sph[:, 2] += np.float32(0.5)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    table, desc = Cf.config3("pinv")
    from riemannian_motion_policies_amd.engine import Engine
    eng = Engine(desc, 0)
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    M = torch.zeros(R, 9, 9, dtype=torch.float64, device="cuda")
    # (config 3 is a symmetric set with an inertia leaf: by default its strict step is ONE certifying launch -- the test below;
    # RMP2_STRICT_CERTIFY=0 keeps the two kernels, which non-symmetric and rank-deficient sets always take)
    eng = _engine_env(desc, RMP2_STRICT_CERTIFY="0")
    got = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), status=st)
    assert "rmp2_pinv_kernel" in eng.last_kernel()
    # a non-symmetric set (JointLimitAvoidance scales columns, quirk Q2): certified from U and the largest multiplier -- one
    # launch as well, the same q-double-dot as its all-Jacobi two-kernel step to a last-place flip, and the oracle's
    _, d2 = Cf.config2("pinv")
    for reps, mapping in ((1, "hex"), (4, "quad")):     # 5 000 robots: the hex mapping certifies; 20 000: the quad mapping
        qq, qqd, gg = (x.repeat(reps, 1) for x in (q, qd, goal))
        e2 = Engine(d2, 0)
        j2 = _engine_env(d2, RMP2_STRICT_CERTIFY="0", **({"RMP2_KERNEL": "hex"} if mapping == "hex" else {}))
        s2 = torch.zeros(R * reps, dtype=torch.int32, device="cuda")
        g2 = e2.step(qq, qqd, gg, status=s2)
        assert "certified" in e2.last_kernel() and mapping in e2.last_kernel(), e2.last_kernel()
        w2 = j2.step(qq, qqd, gg)
        assert ("strict pseudo-inverse" if mapping == "hex" else "rmp2_pinv_kernel") in j2.last_kernel(), j2.last_kernel()
        torch.cuda.synchronize()
        ulp2 = np.spacing(np.abs(w2.cpu().numpy()).max(axis=1, keepdims=True).astype(np.float32))
        assert (np.abs(g2.cpu().numpy() - w2.cpu().numpy()) <= 2.0 * ulp2).all()
        assert ((s2.cpu().numpy() & D.STATUS_JACOBI) != 0).mean() < 0.01
        _check(g2[:256].cpu().numpy(), O.step(d2, s["q"][:256], s["qd"][:256], s["goal"][:256])["qdd64"], f"config 2, strict, certified ({mapping})")
    lane = _engine(desc, "lane")
    want_lane = lane.step(q, qd, goal, obstacles=lane.obstacles(spheres=torch.from_numpy(sph)))
    assert "STRICT" in lane.last_kernel()
    torch.cuda.synchronize()
    n = 512
    ref = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=sph)
    # (parity is asserted for robots clear of contact, as everywhere: >= 0.05 m between every control point and every sphere)
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    org = O.forward_kinematics(desc, s["q"], precision="f64")[:, frames][:, :, :3, 3]
    clr = (np.linalg.norm(org[:, :, None, :] - sph[None, None, :, :3], axis=-1) - sph[None, None, :, 3]).min(axis=(1, 2))
    clear = clr >= 0.05
    assert clear[:n].sum() > n // 3
    _check(got[:n].cpu().numpy(), ref["qdd64"], "two-kernel strict step vs oracle", mask=clear[:n])
    near = ~clear[:n]           # ... and the robots near contact are bounded too: every one passes oracle.accuracy_gate
    verdict = O.accuracy_gate(got[:n].cpu().numpy()[near], {k: ref[k][near] for k in ("qdd64", "M", "f")},
                              spread=O.fp32_resolution(desc, s["q"][:n][near], s["qd"][:n][near], s["goal"][:n][near], spheres=sph))
    assert verdict["ok"].all(), f"two-kernel strict step, near contact: {O.gate_summary(verdict)}"
    _check(got.cpu().numpy(), want_lane.cpu().numpy().astype(np.float64), "two-kernel strict step vs lane strict kernel", mask=clear)
    assert (st[torch.from_numpy(clear).cuda()] == 0).all()
    # a caller who asks for the combined metric (debug output, robot index slowest) gets it copied out of the two kernels'
    # exchange buffer (round 3: such a call fell back to the lane kernel)
    got2 = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), M=M)
    torch.cuda.synchronize()
    assert "rmp2_pinv_kernel" in eng.last_kernel() and torch.equal(got2, got)
    Mn, c = M[:n].cpu().numpy(), clear[:n]
    assert (np.abs(Mn - ref["M"]).max(axis=(1, 2))[c] <= 1e-5 * np.abs(ref["M"]).max(axis=(1, 2))[c]).all()
    # a NaN state: NaN out, flagged, neighbours untouched
    qb = q.clone()
    qb[7, 3] = float("nan")
    gb = eng.step(qb, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), status=st)
    torch.cuda.synchronize()
    assert torch.isnan(gb[7]).all() and (st[7] & D.STATUS_NONFINITE) and torch.equal(gb[:7], got[:7]) and torch.equal(gb[8:], got[8:])
    gl = lane.step(qb, qd, goal, obstacles=lane.obstacles(spheres=torch.from_numpy(sph)))
    assert torch.isnan(gl[7]).all()
    # rank-deficient set under AUTO: a lone target attractor on the Panda (rank <= 3): every robot takes the pseudo-inverse
    spec = D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, table.frame_index("panda_grasptarget_hand"),
                      Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3)
    d1 = D.build_desc(table, [spec])
    e1 = Engine(d1, 0)
    g1 = e1.step(q, qd, goal, status=st)
    assert "rmp2_pinv_kernel" in e1.last_kernel()
    f1 = torch.zeros(R, 9, dtype=torch.float64, device="cuda")
    e1.step(q, qd, goal, M=M, f=f1)          # (the system itself: through the lane kernel's debug outputs)
    torch.cuda.synchronize()
    # WHAT the pseudo-inverse of such a system returns is decided by rounding noise in its six "zero" singular values (fp32
    # pull-backs: ~1e-8 of the largest, far above TensorFlow's cutoff of 2e-14) and differs between any two implementations;
    # pinned here: the system is consistent (f = J^T A e lies in the range of M = J^T A J), so the result solves it
    x = g1.double()
    res = (torch.einsum("rij,rj->ri", M, x) - f1).abs().max(dim=1).values
    scale = (torch.einsum("rij,rj->ri", M.abs(), x.abs()) + f1.abs()).max(dim=1).values
    assert torch.isfinite(g1).all() and (res <= 1e-4 * scale).all(), (res / scale).max().item()
    r1 = O.step(d1, s["q"][:n], s["qd"][:n], s["goal"][:n])
    assert np.abs(M[:n].cpu().numpy() - r1["M"]).max() <= 1e-5 * np.abs(r1["M"]).max()



@pytest.mark.parametrize("R", [64, 65536])
def test_strict_step_certifies_full_rank_and_keeps_the_jacobi_for_the_rest(torch_mod, R):
    """solve = "pinv" is the reference's ONLY resolve (rmp.py:153-154: tf.linalg.pinv with rcond = 10 n eps).  Where every
    singular value of M lies above that cutoff, pinv(M) IS inv(M): the quad mapping's elimination certifies that per robot
    (comparison-matrix bound on the LDL^T factor, rmp2_quad.h) and only uncertified robots take the Jacobi pseudo-inverse.
    Same numbers as the Jacobi pseudo-inverse on EVERY robot (RMP2_STRICT_CERTIFY=0: the two-kernel step on the same combined
    systems), to fp32 rounding of the output; nearly every robot of the unrestricted perf fleet is certified; robots whose
    metric is made nearly singular are NOT certified, take the Jacobi pseudo-inverse and agree with the all-Jacobi step."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    from riemannian_motion_policies_amd.engine import Engine
    s = Cf.sample_panda_states(np.random.default_rng(1), R)          # the bench's perf inputs: near-contact robots included
    sph = Cf.sample_spheres(np.random.default_rng(7))
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    table, desc = Cf.config3("pinv")
    # small fleets take the 16-lanes-per-robot mapping, whose Gauss-Jordan certifies from its own pivot rows; fleets the quad
    # mapping.  The all-Jacobi counterpart on the SAME systems: the hex mapping's strict careful path / the two kernels
    mapping = "hex" if R <= 8192 else "quad"
    eng = Engine(desc, 0)
    jac = _engine_env(desc, RMP2_STRICT_CERTIFY="0", **({"RMP2_KERNEL": "hex"} if mapping == "hex" else {}))
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    got = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=torch.from_numpy(sph)), status=st)
    assert "certified" in eng.last_kernel() and mapping in eng.last_kernel(), eng.last_kernel()
    want = jac.step(q, qd, goal, obstacles=jac.obstacles(spheres=torch.from_numpy(sph)))
    assert ("strict pseudo-inverse" if mapping == "hex" else "rmp2_pinv_kernel") in jac.last_kernel(), jac.last_kernel()
    torch.cuda.synchronize()
    g, w, stc = got.cpu().numpy(), want.cpu().numpy(), st.cpu().numpy()
    fin = np.isfinite(w).all(axis=1)
    assert np.array_equal(fin, np.isfinite(g).all(axis=1))
    # the two resolves differ in fp64 by ~eps cond(M): after rounding to fp32 the outputs are identical up to a last-place flip
    ulp = np.spacing(np.abs(w[fin]).max(axis=1, keepdims=True).astype(np.float32))
    assert (np.abs(g[fin] - w[fin]) <= 2.0 * ulp).all(), f"worst {(np.abs(g[fin] - w[fin]) / ulp).max():.1f} ulp of the row maximum"
    assert (g[fin] == w[fin]).mean() > 0.99
    jacobi = (stc & D.STATUS_JACOBI) != 0
    assert jacobi.mean() < 0.01, f"{jacobi.sum()} of {R} robots not certified"
    assert not (stc & D.STATUS_PINV_PATH).any()
    n = min(R, 256)
    ref = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=sph)
    verdict = O.accuracy_gate(g[:n], ref, spread=O.fp32_resolution(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=sph))
    assert verdict["ok"].all(), O.gate_summary(verdict)
    print(f"R={R}: {int(jacobi.sum())} robots through the Jacobi pseudo-inverse; {O.gate_summary(verdict)}")
    # singular metrics under a certifying handle: a lone target attractor (rank <= 3 of 9 up to fp32 noise; the finger joints do
    # not move its frame at all) plus a damping leaf whose inertia is 1e-30 -- rows 7 and 8 of M are zero but for that diagonal.  The
    # certificate must refuse every robot (a pivot below 1e-11 max|M|), the Jacobi pseudo-inverse must drop those singular values and
    # return a solution of the consistent system
    m = min(R, 4096)
    specs = [D.LeafSpec(D.LEAF_TARGET_ATTRACTOR, D.TASKMAP_FK_POSITION, table.frame_index("panda_grasptarget_hand"),
                        Cf.TARGET_ATTRACTOR_PARAMS, goal_len=3),
             D.LeafSpec(D.LEAF_JOINT_DAMPING, D.TASKMAP_IDENTITY, -1, [1.0, 0.0, 1e-30])]
    dn = D.build_desc(table, specs, "pinv")
    en = Engine(dn, 0)
    jn = _engine_env(dn, RMP2_STRICT_CERTIFY="0", **({"RMP2_KERNEL": "hex"} if m <= 8192 and mapping == "hex" else {}))
    stn = torch.zeros(m, dtype=torch.int32, device="cuda")
    gn = en.step(q[:m], qd[:m], goal[:m], status=stn)
    assert "certified" in en.last_kernel()
    Mn = torch.zeros((m, 9, 9), dtype=torch.float64, device="cuda")
    fn = torch.zeros((m, 9), dtype=torch.float64, device="cuda")
    wn = jn.step(q[:m], qd[:m], goal[:m], M=Mn, f=fn)
    assert "pseudo-inverse" in jn.last_kernel() or "rmp2_pinv_kernel" in jn.last_kernel()
    torch.cuda.synchronize()
    stn = stn.cpu().numpy()
    assert ((stn & D.STATUS_JACOBI) != 0).all(), "a metric with rows of 1e-30 must not be certified"
    assert ((stn & D.STATUS_RANK_DROP) != 0).all(), "the zero singular values must be reported as dropped"
    assert torch.isfinite(gn).all() and (gn[:, 7:] == 0).all() and (wn[:, 7:] == 0).all(), "dropped directions resolve to 0"
    # WHAT the pseudo-inverse returns on the rank-3-plus-noise block is decided by that noise (and differs between any two
    # Jacobi implementations); pinned: both results solve the consistent system
    for x in (gn.double(), wn.double()):
        resid = (torch.einsum("rij,rj->ri", Mn, x) - fn).abs().amax(dim=1)
        scale_ = (torch.einsum("rij,rj->ri", Mn.abs(), x.abs()) + fn.abs()).amax(dim=1)
        assert (resid <= 1e-4 * scale_).all(), (resid / scale_).max().item()


@pytest.mark.parametrize("R", [8209, 40000])
def test_explicit_pairs_streamed_by_lds_dma(torch_mod, R):
    """Interface B at fleet size (the reference's Datamanager layout, data_management.py:8-37 -> taskmap.py:115-138): with
    RMP2_EXPLICIT_GLDS=1 the plain two-wave build streams the pair arrays half a leaf ahead with global_load_lds_dwordx4 and
    evaluates the in-range pairs compacted per quad (rmp2_quad.h pair_loop_explicit_glds).  The same pairs as the register-load
    form, summed in another order: equal to fp32 rounding, tail robots included (R is not a multiple of 16); right against the
    oracle; a layout the DMA cannot take (leaf segments not 16-byte aligned) falls back to the register loads by itself."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    _, desc = Cf.config3()
    rng = np.random.default_rng(R)
    s = Cf.sample_panda_states(rng, R)
    sph = Cf.sample_spheres(rng)
    sph[:, 2] += np.float32(0.4)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    eng, reg = _engine_env(desc, RMP2_EXPLICIT_GLDS="1"), Engine(desc, 0)   # (the stream is opt-in: measured no faster)
    pl, po = eng.closest_points(q, eng.obstacles(spheres=torch.from_numpy(sph)))
    assert pl.shape[1] == 256 and pl.data_ptr() % 16 == 0
    a = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=pl, p_obs=po))
    b = reg.step(q, qd, goal, obstacles=reg.obstacles(p_link=pl, p_obs=po))
    torch.cuda.synchronize()
    assert "quad" in eng.last_kernel()
    # (the streamed form evaluates a leaf's in-range pairs in another order -- compacted per quad --: equal to fp32 rounding of the
    # sums, which near contact is the robot's fp32 resolution; every robot is then bounded against the oracle below)
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    calm = np.abs(bn).max(axis=1) <= 50.0
    assert np.isfinite(an).all() == np.isfinite(bn).all()
    assert (np.abs(an - bn).max(axis=1)[calm] <= 2e-6 * np.maximum(1.0, np.abs(bn).max(axis=1))[calm]).all()
    sub = np.unique(np.concatenate([np.arange(48), np.arange(R - 48, R), rng.integers(0, R, 160)]))
    kw = dict(p_link=pl[sub].cpu().numpy(), p_obs=po[sub].cpu().numpy())
    ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw)
    verdict = O.accuracy_gate(a[sub].cpu().numpy(), ref, spread=O.fp32_resolution(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw))
    assert verdict["ok"].all(), O.gate_summary(verdict)
    # a view shifted by one float: the same pairs at addresses the 16-byte DMA cannot take -> the register-load form, same result
    pl1 = torch.empty(pl.numel() + 4, dtype=torch.float32, device="cuda")[1:1 + pl.numel()].view_as(pl)
    pl1.copy_(pl)
    assert pl1.data_ptr() % 16 != 0
    c = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=pl1, p_obs=po))
    torch.cuda.synchronize()
    assert torch.equal(b, c) or torch.equal(torch.nan_to_num(b), torch.nan_to_num(c)), "the fall-back must be the register-load form"


@pytest.mark.parametrize("R,solve", [(16411, "auto"), (40000, "pinv"), (65536, "pinv")])
def test_explicit_pairs_streamed_pair_phase_before_the_pull_back(torch_mod, R, solve):
    """Interface B in its streamed form (round 5; rmp2_quad.h kObsExplicitStream): the pair phase of all leaf frames runs BEFORE any
    pull-back -- pair arrays by LDS-DMA through the frame records' LDS, sums per frame kept in registers, 128 registers and 9.6 KB of
    LDS per wave (sixteen waves per CU) -- and the frame loop then pulls the stored sums back.  Same pairs, same formulae as the
    single-loop form (the default), summed in another order: equal to fp32 rounding of the sums, tail robots included; right against
    the oracle (every robot through the gate); opt-in (RMP2_EXPLICIT_STREAM=1: measured no faster at one round of waves, DESIGN.md
    section 8); a layout the DMA cannot take falls back to the single-loop form by itself."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    _, desc = Cf.config3(solve)
    rng = np.random.default_rng(R)
    s = Cf.sample_panda_states(rng, R)
    sph = Cf.sample_spheres(rng)
    sph[:, 2] += np.float32(0.4)
    q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
    eng = _engine_env(desc, RMP2_EXPLICIT_STREAM="1")
    old = Engine(desc, 0)
    pl, po = eng.closest_points(q, eng.obstacles(spheres=torch.from_numpy(sph)))
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    a = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=pl, p_obs=po), status=st)
    assert "streamed" in eng.last_kernel(), eng.last_kernel()
    b = old.step(q, qd, goal, obstacles=old.obstacles(p_link=pl, p_obs=po))
    assert "streamed" not in old.last_kernel(), old.last_kernel()
    torch.cuda.synchronize()
    an, bn = a.cpu().numpy(), b.cpu().numpy()
    calm = np.abs(bn).max(axis=1) <= 50.0
    assert np.array_equal(np.isfinite(an).all(axis=1), np.isfinite(bn).all(axis=1))
    assert (np.abs(an - bn).max(axis=1)[calm] <= 2e-6 * np.maximum(1.0, np.abs(bn).max(axis=1))[calm]).all()
    sub = np.unique(np.concatenate([np.arange(64), np.arange(R - 64, R), rng.integers(0, R, 384)]))
    kw = dict(p_link=pl[sub].cpu().numpy(), p_obs=po[sub].cpu().numpy())
    ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw)
    truth = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], precision="f64", **kw)["qdd64"]
    verdict = O.accuracy_gate(an[sub], ref, truth=truth, envelope=O.fp32_envelope(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw))
    assert verdict["ok"].all(), O.gate_summary(verdict)
    assert not (st.cpu().numpy() & D_STATUS_NONFINITE).any()
    # a view shifted by one float: addresses the 16-byte DMA cannot take -> the single-loop form, bit for bit
    pl1 = torch.empty(pl.numel() + 4, dtype=torch.float32, device="cuda")[1:1 + pl.numel()].view_as(pl)
    pl1.copy_(pl)
    c = eng.step(q, qd, goal, obstacles=eng.obstacles(p_link=pl1, p_obs=po))
    assert "streamed" not in eng.last_kernel()
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(b), torch.nan_to_num(c))
    # a NaN joint and a NaN pair: the robot answers NaN + status bit, its neighbours are untouched
    q2 = q.clone()
    q2[5, 2] = float("nan")
    pl2 = pl.clone()
    pl2[21, 40, 1] = float("nan")
    d = eng.step(q2, qd, goal, obstacles=eng.obstacles(p_link=pl2, p_obs=po), status=st)
    assert "streamed" in eng.last_kernel()
    torch.cuda.synchronize()
    dn, stn = d.cpu().numpy(), st.cpu().numpy()
    assert np.isnan(dn[5]).all() and np.isnan(dn[21]).all() and (stn[[5, 21]] & D_STATUS_NONFINITE).all()
    keep = np.ones(R, bool)
    keep[[5, 21]] = False
    assert np.array_equal(dn[keep], an[keep])


@pytest.mark.parametrize("kernel", ["hex", "quad", "lane"])
def test_vanishing_pair_weights_stay_finite(torch_mod, kernel):
    """Control points that move AWAY from their obstacles.  The reference's velocity gate `1. - tf.sigmoid(z)`, z = xdot / gate_length
    (rmp2.py:189-194), cancels in fp32; the kernels form it as e^-z / (1 + e^-z), exact to a few ulp, and keep the reference's exact 0
    beyond z = 17.33 where an fp32 sigmoid is 1.  Three kinds of second pair per leaf: the same receding obstacle again (all weights
    exactly 0), an ordinary one beside the point, and one at the edge of the modulation radius with z ~ 16.5 -- gate (1 - x / r)^2 ~ 1e-14
    times 1e-7: weights of ~1e-23, whose SQUARE leaves fp32's range; the rank-one pull-back once normalised such a metric by rsqrt(0)
    (fuzz seeds 33 / 200016 / 300032).  Every robot finite, unflagged, and on the exact evaluation's value."""
    torch = torch_mod
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    R = 96
    s = Cf.sample_panda_states(np.random.default_rng(41), R)
    s["qd"] = (s["qd"] * np.float32(10.0)).astype(np.float32)       # control points at 0.1 .. 1 m/s: z = 10 .. 100
    _, desc = Cf.config3()
    gate_len, radius = Cf.OBSTACLE_AVOIDANCE_PARAMS[4], Cf.OBSTACLE_AVOIDANCE_PARAMS[7]
    rng = np.random.default_rng(42)
    pl, po = [], []
    kind = np.arange(R)[:, None] % 3
    for i in D.distance_leaf_indices(desc):
        x, xd, _, _ = O.differentiate(desc, s["q"], s["qd"], desc.leaves[i].frame, "f64")
        p, v = x[:, [3, 7, 11]], xd[:, [3, 7, 11]]
        vn = np.linalg.norm(v, axis=1, keepdims=True)
        rnd = rng.normal(size=(R, 3))
        vh = np.where(vn > 1e-6, v / np.maximum(vn, 1e-30), rnd / np.linalg.norm(rnd, axis=1, keepdims=True))   # (the base links do not move)
        side = np.cross(vh, rng.normal(size=(R, 3)))
        side /= np.linalg.norm(side, axis=1, keepdims=True)
        behind = p - rng.uniform(0.1, 0.4, (R, 1)) * vh          # the point leaves this obstacle along its own velocity
        beside = p + 0.3 * side                                   # an ordinary pair: xdot = 0
        cos = np.minimum(1.0, 16.5 * gate_len / np.maximum(vn, 1e-30))
        nh = cos * vh + np.sqrt(1.0 - cos * cos) * side           # n . v = 16.5 gate_length where the point is fast enough
        edge = p - radius * (1.0 - 2e-7) * nh                     # x = r (1 - 2e-7): inside the modulation radius by a few fp32 roundings
        pl.append(np.repeat(p[:, None, :], 2, axis=1))
        po.append(np.stack([behind, np.where(kind == 0, behind, np.where(kind == 1, beside, edge))], axis=1))
    pl, po = (np.concatenate(a, axis=1).astype(np.float32) for a in (pl, po))
    eng = _engine(desc, kernel)
    st = torch.zeros(R, dtype=torch.int32, device="cuda")
    qdd = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]),
                   obstacles=eng.obstacles(p_link=torch.from_numpy(pl), p_obs=torch.from_numpy(po)), status=st)
    torch.cuda.synchronize()
    got = qdd.cpu().numpy()
    assert np.isfinite(got).all() and not (st.cpu().numpy() & D.STATUS_NONFINITE).any()
    exact = O.step(desc, s["q"], s["qd"], s["goal"], precision="f64", p_link=pl, p_obs=po)["qdd64"]
    # (joint velocities of 5 rad/s make a few robots ill-conditioned whatever the obstacles do: robots an fp32 evaluation of the
    #  reference's formulae itself resolves to 2e-6 are the ones held to the north star here)
    c32 = O.step(desc, s["q"], s["qd"], s["goal"], precision="f32", p_link=pl, p_obs=po)["qdd64"]
    well = np.abs(c32 - exact).max(axis=1) <= 2e-6 * np.maximum(1.0, np.abs(exact).max(axis=1))
    assert well.sum() >= 0.9 * R and all(well[np.arange(R) % 3 == k].sum() >= 0.8 * R / 3 for k in range(3))
    _check(got, exact, f"vanishing pair weights {kernel}", mask=well)
