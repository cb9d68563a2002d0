"""ctypes front-end of the CPU oracle (oracle/librmp2_oracle.so).

TEST INFRASTRUCTURE -- see the header of rmp2_oracle.c.  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from riemannian_motion_policies_amd import descriptor as D  # noqa: E402  (struct layouts only)

_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "librmp2_oracle.so")
    src = os.path.join(_HERE, "rmp2_oracle.c")
    hdr = os.path.join(_ROOT, "include", "rmp2.h")
    stale = (not os.path.exists(so)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(so) for p in (src, hdr))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for p in ("f32", "f64"):
            assert getattr(_LIB, f"orc_sizeof_desc_{p}")() == C.sizeof(D.Desc), "rmp2_desc layout mismatch"
    return _LIB


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def make_obstacles(desc: D.Desc, *, spheres=None, p_link=None, p_obs=None, pair_counts=None, csr_offset=None,
                   csr_index=None, dist=None, primitive=None):
    """Host-pointer `rmp2_obstacles`; returns (struct, keepalive)."""
    o = D.Obstacles()
    keep = []
    if p_link is not None:
        p_link = np.ascontiguousarray(p_link, dtype=np.float32)
        p_obs = np.ascontiguousarray(p_obs, dtype=np.float32)
        assert p_link.shape == p_obs.shape and p_link.ndim == 3 and p_link.shape[2] == 3
        o.mode = D.OBS_EXPLICIT_PAIRS
        o.n_pairs = p_link.shape[1]
        dl = D.distance_leaf_indices(desc)
        if pair_counts is None:
            assert o.n_pairs % max(len(dl), 1) == 0
            pair_counts = [o.n_pairs // max(len(dl), 1)] * len(dl)
        assert len(pair_counts) == len(dl) and sum(pair_counts) == o.n_pairs
        acc, k = 0, 0
        for i in range(desc.n_leaves + 1):
            o.pair_begin[i] = acc
            if i < desc.n_leaves and i in dl:
                acc += pair_counts[k]
                k += 1
        o.p_link = p_link.ctypes.data
        o.p_obs = p_obs.ctypes.data
        keep += [p_link, p_obs]
        if dist is not None:
            dist = np.ascontiguousarray(dist, dtype=np.float32)
            assert dist.shape == p_link.shape[:2]
            o.dist = dist.ctypes.data
            keep.append(dist)
    elif spheres is not None:
        spheres = np.ascontiguousarray(spheres, dtype=np.float32)
        assert spheres.ndim == 2 and spheres.shape[1] in (4, 8)
        o.primitive = D.PRIM_CAPSULE if spheres.shape[1] == 8 else D.PRIM_SPHERE
        if primitive == "cylinder":   # (centre, radius, unit axis, half height): the reference's flat-capped cylinders
            assert spheres.shape[1] == 8
            o.primitive = D.PRIM_CYLINDER
        o.n_spheres = spheres.shape[0]
        o.spheres = spheres.ctypes.data
        keep.append(spheres)
        if csr_offset is not None:
            csr_offset = np.ascontiguousarray(csr_offset, dtype=np.int32)
            csr_index = np.ascontiguousarray(csr_index, dtype=np.int32)
            o.mode = D.OBS_RAGGED_SPHERES
            o.csr_offset = csr_offset.ctypes.data
            o.csr_index = csr_index.ctypes.data
            keep += [csr_offset, csr_index]
        else:
            o.mode = D.OBS_SHARED_SPHERES
    else:
        o.mode = D.OBS_NONE
    return o, keep


def step(desc: D.Desc, q, qd, goal=None, *, precision="f32", **obstacle_kwargs):
    """RmpCore.evaluate for R robots on the CPU.  Returns dict(qdd fp32, qdd64, M, f, status)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    qd = np.ascontiguousarray(qd, dtype=np.float32)
    assert q.ndim == 2 and q.shape == qd.shape and q.shape[1] == desc.robot.n_dof
    R, n = q.shape
    goal_stride = 0
    if goal is not None:
        goal = np.ascontiguousarray(goal, dtype=np.float32)
        if goal.ndim == 1:
            goal = np.broadcast_to(goal, (R, goal.shape[0])).copy()
        assert goal.shape == (R, desc.goal_floats), (goal.shape, desc.goal_floats)
        goal_stride = desc.goal_floats
    elif desc.goal_floats:
        raise ValueError("this RMP set needs a goal array")
    obs, keep = make_obstacles(desc, **obstacle_kwargs)
    qdd = np.empty((R, n), np.float32)
    qdd64 = np.empty((R, n), np.float64)
    M = np.empty((R, n, n), np.float64)
    f = np.empty((R, n), np.float64)
    status = np.empty(R, np.uint32)
    fn = getattr(lib(), f"orc_step_{precision}")
    rc = fn(C.byref(desc), _ptr(q, C.c_float), _ptr(qd, C.c_float), _ptr(goal, C.c_float), C.c_int(goal_stride),
            C.byref(obs), _ptr(qdd, C.c_float), _ptr(qdd64, C.c_double), _ptr(M, C.c_double), _ptr(f, C.c_double),
            _ptr(status, C.c_uint32), C.c_int(R))
    if rc != 0:
        raise RuntimeError(f"orc_step_{precision} failed: {rc}")
    del keep
    return {"qdd": qdd, "qdd64": qdd64, "M": M, "f": f, "status": status}


def forward_kinematics(desc: D.Desc, q, precision="f32"):
    q = np.ascontiguousarray(q, dtype=np.float32)
    R, F = q.shape[0], desc.robot.n_frames
    dt, ct = (np.float32, C.c_float) if precision == "f32" else (np.float64, C.c_double)
    T = np.empty((R, F, 4, 4), dt)
    getattr(lib(), f"orc_forward_kinematics_{precision}")(C.byref(desc), _ptr(q, C.c_float), _ptr(T, ct), C.c_int(R))
    return T


def differentiate(desc: D.Desc, q, qd, frame: int, precision="f32"):
    q = np.ascontiguousarray(q, dtype=np.float32)
    qd = np.ascontiguousarray(qd, dtype=np.float32)
    R, n = q.shape
    dt, ct = (np.float32, C.c_float) if precision == "f32" else (np.float64, C.c_double)
    x, xd, c = (np.empty((R, 16), dt) for _ in range(3))
    J = np.empty((R, 16, n), dt)
    getattr(lib(), f"orc_differentiate_{precision}")(C.byref(desc), _ptr(q, C.c_float), _ptr(qd, C.c_float),
                                                      C.c_int(frame), _ptr(x, ct), _ptr(xd, ct), _ptr(J, ct),
                                                      _ptr(c, ct), C.c_int(R))
    return x, xd, J, c


def differentiate_euler(desc: D.Desc, q, qd, frame: int, precision="f32"):
    """(x, xd, J, c) of the chain [FK(frame), TaskmapFrom4x4ToEuler] (taskmap.py:57-67)."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    qd = np.ascontiguousarray(qd, dtype=np.float32)
    R, n = q.shape
    dt, ct = (np.float32, C.c_float) if precision == "f32" else (np.float64, C.c_double)
    x, xd, c = (np.empty((R, 3), dt) for _ in range(3))
    J = np.empty((R, 3, n), dt)
    getattr(lib(), f"orc_differentiate_euler_{precision}")(C.byref(desc), _ptr(q, C.c_float), _ptr(qd, C.c_float),
                                                            C.c_int(frame), _ptr(x, ct), _ptr(xd, ct), _ptr(J, ct),
                                                            _ptr(c, ct), C.c_int(R))
    return x, xd, J, c


def pinv_solve(M, f):
    M = np.ascontiguousarray(M, np.float64)
    f = np.ascontiguousarray(f, np.float64)
    n = f.shape[0]
    x = np.empty(n, np.float64)
    dropped = lib().orc_pinv_solve_f64(C.c_int(n), _ptr(M, C.c_double), _ptr(f, C.c_double), _ptr(x, C.c_double))
    return x, dropped


def ulp_spread(desc: D.Desc, q, qd, goal=None, trials: int = 4, seed: int = 0, **obstacle_kwargs):
    """Per robot: how far q-double-dot (fp64 evaluation) moves when every fp32 input -- q, qd, goal, sphere / capsule table,
    explicit pairs -- moves by one fp32 rounding at UNIT scale, `a -> a + s * 2^-23 * max(|a|, 1)` with a random sign s per
    element (an ulp of the value, but at least an ulp of 1.0: the intermediates of the kinematic chain -- sines, cosines,
    positions in metres -- live at unit scale whatever the magnitude of the joint angle); the maximum over `trials` draws,
    inf-norm.  It is the resolution fp32 gives the result, whatever amplifies it (cond(M), exp(-x / 0.01) near contact, 1 / x^2
    repulsion): no fp32 evaluation of the reference algorithm, TensorFlow's included, can promise a robot more than a small
    multiple of it."""
    rng = np.random.default_rng(seed)
    eps = np.float64(2.0 ** -23)

    def jiggle(a):
        a = np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)
        return (a + rng.choice(np.array([-1.0, 1.0]), a.shape) * eps * np.maximum(np.abs(a), 1.0)).astype(np.float32)

    base = step(desc, q, qd, goal, precision="f64", **obstacle_kwargs)["qdd64"]
    spread = np.zeros(len(base))
    for _ in range(trials):
        kw = {k: (jiggle(v) if k in ("spheres", "p_link", "p_obs", "dist") and v is not None else v) for k, v in obstacle_kwargs.items()}
        r = step(desc, jiggle(q), jiggle(qd), None if goal is None else jiggle(goal), precision="f64", **kw)["qdd64"]
        with np.errstate(invalid="ignore"):
            spread = np.fmax(spread, np.abs(r - base).max(axis=1))
    return spread


def fp32_resolution(desc: D.Desc, q, qd, goal=None, trials: int = 4, seed: int = 0, **obstacle_kwargs):
    """What fp32 can resolve of a robot's q-double-dot: the larger of ulp_spread (response of the exact result to one
    unit-scale fp32 rounding of every input) and the distance of this oracle's own REFERENCE-PRECISION evaluation (fp32 leaves,
    fp64 sum and resolve: rmp.py:133-155) from its fp64 evaluation.  The `spread` argument of accuracy_gate."""
    sp = ulp_spread(desc, q, qd, goal, trials=trials, seed=seed, **obstacle_kwargs)
    a = step(desc, q, qd, goal, precision="f32", **obstacle_kwargs)["qdd64"]
    b = step(desc, q, qd, goal, precision="f64", **obstacle_kwargs)["qdd64"]
    with np.errstate(invalid="ignore"):
        return np.fmax(sp, np.abs(a - b).max(axis=1))


def fp32_envelope(desc: D.Desc, q, qd, goal=None, samples: int = 16, seed: int = 0, **obstacle_kwargs):
    """Per robot: how far an fp32 evaluation of the reference's formulae lands from their fp64 evaluation -- the largest of
    `samples` + 1 draws of |C oracle with fp32 leaves - C oracle in fp64|_inf: the plain fp32 evaluation, and `samples` more with
    every fp32 input (q, qd, goal, table / pairs) moved by one unit-scale fp32 rounding of random sign, each of which sends the
    fp32 evaluation down another rounding path (the fp64 yardstick stays on the UNPERTURBED inputs: the draw then also holds the
    response to an input ulp).  One draw is a single sample of a robot's fp32 noise -- two faithful fp32 evaluations of one robot
    differ from each other by several times either one's error, by chance --; the maximum of seventeen is an envelope a further
    faithful evaluation exceeds by a factor 2 about never (tools/accuracy_survey.py prints the control: the plain evaluation against
    the envelope of the others).  What `accuracy_gate(..., truth=, envelope=)` holds the engine to on the robots the absolute 1e-5
    cannot cover; tests/golden/perf_envelope.npz adds the error of the autograd restatement of the reference's own graph."""
    rng = np.random.default_rng(seed)
    eps = np.float64(2.0 ** -23)

    def jiggle(a):
        a = np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)
        return (a + rng.choice(np.array([-1.0, 1.0]), a.shape) * eps * np.maximum(np.abs(a), 1.0)).astype(np.float32)

    base = step(desc, q, qd, goal, precision="f64", **obstacle_kwargs)["qdd64"]
    with np.errstate(invalid="ignore"):
        env = np.abs(step(desc, q, qd, goal, precision="f32", **obstacle_kwargs)["qdd64"] - base).max(axis=1)
        for _ in range(samples):
            kw = {k: (jiggle(v) if k in ("spheres", "p_link", "p_obs", "dist") and v is not None else v) for k, v in obstacle_kwargs.items()}
            r = step(desc, jiggle(q), jiggle(qd), None if goal is None else jiggle(goal), precision="f32", **kw)["qdd64"]
            env = np.fmax(env, np.abs(r - base).max(axis=1))
    return env


def rank_flips(desc: D.Desc, q, qd, goal=None, trials: int = 6, seed: int = 0, **obstacle_kwargs):
    """Per robot: does the NUMBER of non-zero singular values of the reference-precision M change when every fp32 input moves by one
    unit-scale fp32 rounding?  It does when a pair sits within a rounding of a leaf's cutoff radius (rmp2.py:170-174: the gate has a
    double root there -- metric exactly 0 on one side, 1e-12 on the other): a dof whose only metric that is has q-double-dot = f / M =
    O(1..100) in one faithful evaluation and 0 (dropped by the pseudo-inverse) in the next.  tools/fuzz_parity.py counts such robots as
    undetermined at fp32 (seed 510845: one of 1.3 M robots; the engine had the pair in range, both oracle builds out of range)."""
    rng = np.random.default_rng(seed)
    eps = np.float64(2.0 ** -23)

    def jiggle(a):
        a = np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)
        return (a + rng.choice(np.array([-1.0, 1.0]), a.shape) * eps * np.maximum(np.abs(a), 1.0)).astype(np.float32)

    def rank(M):
        with np.errstate(invalid="ignore"):
            sv = np.linalg.svd(np.where(np.isfinite(M), M, 0.0), compute_uv=False)
        return (sv > 1e-18 * np.maximum(sv[:, :1], 1e-300)).sum(axis=1)

    base = rank(step(desc, q, qd, goal, precision="f32", **obstacle_kwargs)["M"])
    flips = np.zeros(len(base), bool)
    for _ in range(trials):
        kw = {k: (jiggle(v) if k in ("spheres", "p_link", "p_obs", "dist") and v is not None else v) for k, v in obstacle_kwargs.items()}
        flips |= rank(step(desc, jiggle(q), jiggle(qd), None if goal is None else jiggle(goal), precision="f32", **kw)["M"]) != base
    return flips


def system_resolution(ref, trials: int = 4, eps: float = 2.0 ** -22, seed: int = 0):
    """Per robot: how far the resolve `pinv(M) f` (TensorFlow's cutoff, rmp.py:153-154) of the oracle's combined system moves when
    every ENTRY of M and f moves by a relative `eps` (two fp32 roundings by default) with a random sign, the maximum over `trials`
    draws, inf-norm.  The reference forms every leaf's J^T A J and J^T A (xdd - c) in fp32 (rmp.py:133-151), so each entry of the
    summed (M, f) carries at least that much relative error in ANY fp32 evaluation; an entrywise-relative perturbation keeps exact
    zeros (a joint no leaf touches stays out of the system).  It is the resolution of the result at the SYSTEM level -- what matters for
    rank-deficient and inconsistent systems (least-squares sensitivity grows with cond^2 times the residual), which the residual
    test B of accuracy_gate cannot judge.  The `system_spread` argument of accuracy_gate."""
    rng = np.random.default_rng(seed)
    M, f = ref["M"], ref["f"]
    R, n = f.shape
    finite = np.isfinite(M).all(axis=(1, 2)) & np.isfinite(f).all(axis=1)
    Mz = np.where(finite[:, None, None], M, 0.0)
    fz = np.where(finite[:, None], f, 0.0)
    rcond = 10.0 * n * np.finfo(np.float64).eps

    def resolve(A, b):
        return np.einsum("rij,rj->ri", np.linalg.pinv(A, rcond=rcond), b)

    base = resolve(Mz, fz)
    spread = np.zeros(R)
    for _ in range(trials):
        dM = 1.0 + eps * rng.choice(np.array([-1.0, 1.0]), Mz.shape)
        df = 1.0 + eps * rng.choice(np.array([-1.0, 1.0]), fz.shape)
        spread = np.fmax(spread, np.abs(resolve(Mz * dM, fz * df) - base).max(axis=1))
    return np.where(finite, spread, 0.0)


# Backward-error bound of clause B.  1e-4 until round 4; the perf fleets put the data under it (profiles/r05_accuracy_survey.txt,
# 2 048 robots per fleet, both resolves): the largest omega over configs 2 / 3 / 3b / 5 is 1.25e-5, the 99th percentile of the robots
# outside clause A below 9e-6 in every fleet -- sphere tables, explicit pairs and ragged lists alike.  Capsule tables (config 3c):
# the 99th percentile is 9.2e-6 as well; three of 2 048 robots, control points INSIDE a capsule, sit above 2e-5 (up to 3.4e-4: the
# clamped nearest point of the axis jumps with a rounding of its parameter there) and are held by the fp32 envelope (clause E)
# instead, as any other robot beyond B.  So one value serves every interface.
ETA = 2e-5


def accuracy_gate(got, ref, eta: float = ETA, atol: float = 1e-5, spread=None, spread_factor: float = 8.0, system_spread=None,
                  truth=None, envelope=None, envelope_factor: float = 2.0):
    """Per-robot accuracy verdict of a computed q-double-dot `got` [R, n] against an oracle result `ref` (the dict of step()).
    EVERY robot gets a bound -- none is exempted for being ill-conditioned or near contact:

      A  (north star)   |got - ref|_inf <= atol * max(1, |ref|_inf)
      B  (backward)     omega = |P (M_ref got - f_ref)|_2 / (|M_ref|_2 |got|_2 + |f_ref|_2) <= eta  (P: projector on the range the
                        oracle's resolve kept -- the identity for a full-rank system)   and
                        |got - ref|_2 <= 4 eta cond_2(M_ref) |ref|_2            (what omega <= eta implies, with slack 2)
                        and |N^T got|_2 <= 1e-3 |ref|_2 when the oracle resolved by a rank-dropping pseudo-inverse (N: the
                        null directions it dropped -- the residual is blind to components there, the minimum norm is not)
      C  (fp32 resolution, only when `spread` = fp32_resolution(...) of the same robots is given)
                        |got - ref|_inf <= spread_factor * spread : within a few times what one fp32 rounding does to the
                        exact result / what the reference-precision oracle itself misses the fp64 result by
      D  (fp32 resolution of the system, only when `system_spread` = system_resolution(ref) is given)
                        |got - ref|_inf <= spread_factor * system_spread : within a few times what two fp32 roundings of every
                        entry of the oracle's (M, f) do to its resolve -- the bound for rank-deficient / inconsistent systems

      E  (fp32 envelope, only when `truth` = the fp64 evaluation's q-double-dot and `envelope` = fp32_envelope(...) of the same robots
                        -- plus, where it exists, the autograd restatement's own error -- are given)
                        |got - truth|_inf <= envelope_factor * envelope : no further from the exact value of the reference's
                        formulae than twice what faithful fp32 evaluations of them are seen to land -- the bound that separates
                        error the kernel ADDS from error any fp32 evaluation has

    B is the statement "got solves a system within relative eta of the oracle's": it is what fp32 leaves can promise a robot
    whose metric is ill-conditioned, and it does not loosen with the condition number -- the forward clause only states its
    consequence.  C covers the robots whose SYSTEM is sensitive (distances of millimetres: an ulp of a position is 1e-4 of the
    distance, and the leaf differentiates exp(-x / 0.01) and 1 / x^2 of it).  Returns dict(a, b, c, d, e, ok: bool arrays -- b
    excludes a, c excludes both, and so on: which clause ADMITTED a robot --; each: the clauses on their own; omega, cond, err_inf)."""
    got = np.asarray(got, np.float64)
    q_ref, M, f = ref["qdd64"], ref["M"], ref["f"]
    with np.errstate(invalid="ignore"):
        err_inf = np.abs(got - q_ref).max(axis=1)
        a = err_inf <= atol * np.maximum(1.0, np.abs(q_ref).max(axis=1))
    finite = np.isfinite(got).all(axis=1) & np.isfinite(M).all(axis=(1, 2)) & np.isfinite(f).all(axis=1) & np.isfinite(q_ref).all(axis=1)
    g = np.where(finite[:, None], got, 0.0)
    Mz = np.where(finite[:, None, None], M, 0.0)
    fz = np.where(finite[:, None], f, 0.0)
    qz = np.where(finite[:, None], q_ref, 0.0)
    U, sv, Vh = np.linalg.svd(Mz)
    # rank the oracle's own resolve kept (TF's cutoff, rmp.py:153-154: 10 n eps_f64 sigma_max)
    n = M.shape[1]
    cutoff = 10.0 * n * np.finfo(np.float64).eps * sv[:, 0]
    kept = (sv > cutoff[:, None]).sum(axis=1)
    # the residual, measured on the RANGE the oracle's resolve kept: a rank-deficient system is in general inconsistent (f has a
    # component outside range(M): the least-squares residual, which the oracle's own answer leaves too), and only the part of
    # M got - f inside the range says anything about got.  For a full-rank system this is the plain residual.
    r_vec = np.einsum("rij,rj->ri", Mz, g) - fz
    r_rng = np.einsum("rji,rj->ri", U, r_vec) * (np.arange(n)[None, :] < kept[:, None])
    res = np.linalg.norm(r_rng, axis=1)
    scale = sv[:, 0] * np.linalg.norm(g, axis=1) + np.linalg.norm(fz, axis=1)
    omega = np.where(scale > 0, res / np.where(scale > 0, scale, 1.0), 0.0)
    smin = np.take_along_axis(sv, np.maximum(kept - 1, 0)[:, None], axis=1)[:, 0]
    cond = np.where(smin > 0, sv[:, 0] / np.where(smin > 0, smin, 1.0), np.inf)
    err2 = np.linalg.norm(g - qz, axis=1)
    ref2 = np.linalg.norm(qz, axis=1)
    fwd = err2 <= 4.0 * eta * cond * np.maximum(ref2, 1e-30)
    # minimum norm: the part of got inside the null space the oracle's resolve dropped (its own answer has none there)
    null_part = np.linalg.norm(np.einsum("rij,rj->ri", Vh, g) * (np.arange(n)[None, :] >= kept[:, None]), axis=1)
    minnorm = (kept == n) | (null_part <= 1e-3 * ref2 + 1e-12)
    b = finite & (omega <= eta) & fwd & minnorm
    c = np.zeros(len(got), bool)
    if spread is not None:
        c = finite & (err_inf <= spread_factor * np.asarray(spread))
    d = np.zeros(len(got), bool)
    if system_spread is not None:
        d = finite & (err_inf <= spread_factor * np.asarray(system_spread))
    e = np.zeros(len(got), bool)
    if truth is not None and envelope is not None:
        with np.errstate(invalid="ignore"):
            e = finite & (np.abs(got - np.asarray(truth, np.float64)).max(axis=1) <= envelope_factor * np.asarray(envelope))
    # a robot the oracle itself resolves to NaN (non-finite state): tf.linalg.pinv of a system holding NaN / Inf is NaN in EVERY
    # entry (rmp2_oracle.c pinv_solve), so the engine must answer non-finite in every dof too -- one NaN joint beside finite
    # ones against an all-NaN reference is a different answer, not the same one
    both_nan = (~np.isfinite(q_ref)).all(axis=1) & (~np.isfinite(got)).all(axis=1)
    return {"a": a, "b": b & ~a, "c": c & ~a & ~b, "d": d & ~a & ~b & ~c, "e": e & ~a & ~b & ~c & ~d, "ok": a | b | c | d | e | both_nan,
            "both_nan": both_nan, "omega": omega, "cond": cond, "err_inf": err_inf,
            # every clause on its own (the exclusive ones above say which clause ADMITTED a robot)
            "each": {"a": a, "b": b, "c": c, "d": d, "e": e}}


def gate_summary(g) -> dict:
    """Counts per admitting branch of an accuracy_gate verdict (for assertion messages and bench.py's result_check)."""
    return {"robots": int(len(g["ok"])), "north_star_1e-5": int(g["a"].sum()), "backward_error": int(g["b"].sum()),
            "input_resolution": int(g["c"].sum()), "system_resolution": int(g["d"].sum()), "fp32_envelope": int(g["e"].sum()), "both_nan": int((g["both_nan"] & ~g["a"]).sum()), "rejected": int((~g["ok"]).sum()),
            "worst_abs_err": float(np.nanmax(g["err_inf"])) if len(g["err_inf"]) else 0.0,
            "worst_omega_beyond_north_star": float(g["omega"][~g["a"]].max()) if (~g["a"]).any() else 0.0}
