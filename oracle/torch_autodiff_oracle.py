"""Autodiff-faithful fp32 PyTorch-CPU restatement of the reference hot path.

TEST INFRASTRUCTURE (see rmp2_oracle.c header).  The reference computes every task-map
derivative with nested TensorFlow GradientTapes; TensorFlow is not installed here, so this
file restates the SAME computation graph with torch.autograd -- forward pass written with the
reference's tensor ops, (x, xd, J, c) obtained by the reference's double-backward
Jacobian-vector-product trick -- one robot per call, FK recomputed per RMP, exactly the way
the reference runs.  It serves two purposes:

  * it pins the analytic C oracle (rmp2_oracle.c): an independent derivation of J, xd,
    c = Jdot qd, every leaf and the resolved qdd  (tests/test_oracle_pins.py), and
  * it generates the committed golden vectors  (tests/golden/make_fixtures.py).

It is labelled "proxy, not TensorFlow": op semantics follow TF 2.10 documentation
(SURVEY tags [TF-doc]); fp32 op ordering inside matmul/reduction kernels differs from
Eigen's at the 1e-7 relative level.

Reference map (file:line):
  reduce_matrix_prod ........................ kinematics.py:12-20
  R_x/R_y/R_z, homogenous_transformation .... kinematics.py:22-71
  rotation_matrix_from_rotation_vector ...... kinematics.py:99-121
  rotation_matrix_from_rpy .................. kinematics.py:123-127
  UrdfForwardKinematic._build/forward/differentiate  kinematics.py:163-270
  jacobian_vector_product / rmp_differentiate helper/rmp_helper.py:3-22,50-60
  soft_norm / directionally_stretched_metric  helper/rmp_helper.py:62-74
  task maps ................................. taskmap.py:13-168
  leaves .................................... rmp2.py:31-226, rmp.py:226-382
  RmpCore.evaluate / _calculate_rmp ......... rmp.py:133-180
"""
from __future__ import annotations

import numpy as np
import torch

torch.set_num_threads(1)
F32 = torch.float32


# ---------------------------------------------------------------------------------------
# kinematics.py
def reduce_matrix_prod(all_T):  # [L,4,4] -> [4,4], left to right starting from eye
    m = torch.eye(4, dtype=F32)
    for i in range(all_T.shape[0]):
        m = m @ all_T[i]
    return m


def R_x(angle):  # angle [B,1]
    c, s, z = torch.cos(angle), torch.sin(angle), torch.zeros_like(angle)
    top = torch.tensor([[1., 0., 0.]]).expand(angle.shape[0], 1, 3)
    mid = torch.stack([z, c, -s], dim=-1)
    bot = torch.stack([z, s, c], dim=-1)
    return torch.cat([top, mid, bot], dim=-2)


def R_y(angle):
    c, s, z = torch.cos(angle), torch.sin(angle), torch.zeros_like(angle)
    top = torch.stack([c, z, s], dim=-1)
    mid = torch.tensor([[0., 1., 0.]]).expand(angle.shape[0], 1, 3)
    bot = torch.stack([-s, z, c], dim=-1)
    return torch.cat([top, mid, bot], dim=-2)


def R_z(angle):
    c, s, z = torch.cos(angle), torch.sin(angle), torch.zeros_like(angle)
    top = torch.stack([c, -s, z], dim=-1)
    mid = torch.stack([s, c, z], dim=-1)
    bot = torch.tensor([[0., 0., 1.]]).expand(angle.shape[0], 1, 3)
    return torch.cat([top, mid, bot], dim=-2)


def homogenous_transformation(R, t):
    B = R.shape[0]
    Rt = torch.cat([R, t[..., None]], dim=-1)
    bottom = torch.cat([torch.zeros(B, 1, 3), torch.ones(B, 1, 1)], dim=-1)
    return torch.cat([Rt, bottom], dim=-2)


def rotation_matrix_from_rotation_vector(vec, angle):  # vec [B,3], angle [B]
    B = vec.shape[0]
    cos = torch.cos(angle)[..., None, None]
    sin = torch.sin(angle)[..., None, None]
    vec_with_zeros = torch.cat([torch.zeros(B, 1), vec], dim=-1)
    eye = torch.eye(3).expand(B, 3, 3)
    outer = torch.einsum('...i,...j->...ij', vec, vec)
    sign = torch.tensor([[1, -1, 1], [1, 1, -1], [-1, 1, 1]], dtype=F32)
    where = torch.tensor([[0, 3, 2], [3, 0, 1], [2, 1, 0]])
    u_tilde = sign * vec_with_zeros[:, where]
    return cos * eye + sin * u_tilde + (1 - cos) * outer


def rotation_matrix_from_rpy(rpy):  # [F,3]
    rpy = rpy[..., None]
    roll, pitch, yaw = rpy[:, 0], rpy[:, 1], rpy[:, 2]
    return R_x(roll) @ R_y(pitch) @ R_z(yaw)


class UrdfForwardKinematicTorch:
    """Built from the golden kinematic-table JSON (the reference parser's own output) so that
    this oracle shares no table code with the product."""

    def __init__(self, golden: dict):
        self.frame_names = list(golden["frame_names"])
        self.order = list(golden["order"])
        paths = golden["backward_paths"]
        F = len(self.frame_names)
        max_len = max(len(p) for p in paths)
        idx = {n: i for i, n in enumerate(self.frame_names)}
        self.kinematic_chains = [[idx[p] for p in path] + [F] * (max_len - len(path)) for path in paths]
        self._q_reordering = torch.tensor(golden["q_reordering"])
        rpy = torch.tensor(golden["rpy"], dtype=F32)
        xyz = torch.tensor(golden["xyz"], dtype=F32)
        self.T_constant = homogenous_transformation(rotation_matrix_from_rpy(rpy), xyz)
        self.axis = torch.tensor(golden["axis"], dtype=F32)
        jt = golden["joint_type"]
        self.is_revolute = torch.tensor([t == "revolute" for t in jt], dtype=F32)[:, None, None]
        self.is_prismatic = torch.tensor([t == "prismatic" for t in jt], dtype=F32)[:, None, None]
        self.is_fixed = torch.tensor([t == "fixed" for t in jt], dtype=F32)[:, None, None]

    def forward(self, q, frame):  # q [1,n] -> [1,4,4]
        q = q.squeeze(0)
        F = self.T_constant.shape[0]
        q = torch.cat([q, torch.zeros(1)], dim=-1)
        q = q[self._q_reordering]
        T_fixed = torch.eye(4).expand(F, 4, 4)
        R_rev = rotation_matrix_from_rotation_vector(self.axis.reshape(-1, 3), q.reshape(-1))
        T_rev = homogenous_transformation(R_rev, torch.zeros(F, 3))
        T_pri = homogenous_transformation(torch.eye(3).expand(F, 3, 3), q[:, None] * self.axis)
        T_var = self.is_fixed * T_fixed + self.is_revolute * T_rev + self.is_prismatic * T_pri
        T = self.T_constant @ T_var
        T = torch.cat([T, torch.eye(4)[None]], dim=0)
        chain = self.kinematic_chains[self.frame_names.index(frame)]
        return reduce_matrix_prod(T[chain])[None]

    def differentiate(self, q, qd, frame):  # q, qd [1,n]
        q = q.squeeze(0).detach().clone().requires_grad_(True)
        qd = qd.squeeze(0)
        x = self.forward(q[None], frame).reshape(-1)
        xd = jacobian_vector_product(x, q, qd)
        J = torch.stack([torch.autograd.grad(x[i], q, retain_graph=True, allow_unused=True)[0]
                         if x[i].requires_grad else torch.zeros_like(q) for i in range(x.shape[0])])
        J = torch.stack([j if j is not None else torch.zeros_like(q) for j in J])
        c = jacobian_vector_product(xd, q, qd)
        return x[None].detach(), xd[None].detach(), J[None].detach(), c[None].detach()


# ---------------------------------------------------------------------------------------
# helper/rmp_helper.py
def jacobian_vector_product(v, u, w):
    """J_v(u) @ w by double backward (helper/rmp_helper.py:50-60); keeps the graph so that it
    can be differentiated again (c = JVP of xd)."""
    if not v.requires_grad:
        return torch.zeros_like(v)
    dummy = torch.ones_like(v, requires_grad=True)
    inner = torch.einsum('...i,...i->...', v, dummy).sum()
    g, = torch.autograd.grad(inner, u, create_graph=True, allow_unused=True)
    if g is None or not g.requires_grad:
        return torch.zeros_like(v)
    inner2 = torch.einsum('...i,...i->...', g, w).sum()
    jvp, = torch.autograd.grad(inner2, dummy, create_graph=True, allow_unused=True)
    return jvp if jvp is not None else torch.zeros_like(v)


def rmp_differentiate(fn):
    def differentiate_fn(q, qd):  # q, qd [B,m]
        q = q.detach().clone().requires_grad_(True)
        x = fn(q)  # [B,k]
        xd = jacobian_vector_product(x, q, qd)
        c = jacobian_vector_product(xd, q, qd)
        B, k = x.shape
        rows = []
        for i in range(k):  # batch_jacobian, one output component at a time
            g, = torch.autograd.grad(x[:, i].sum(), q, retain_graph=True, allow_unused=True)
            rows.append(g if g is not None else torch.zeros_like(q))
        J = torch.stack(rows, dim=1)  # [B,k,m]
        return x.detach(), xd.detach(), J.detach(), c.detach()
    return differentiate_fn


def soft_norm(v, c):
    n = torch.linalg.norm(v, dim=-1)
    h = n + 1 / c * torch.log(1 + torch.exp(-2 * c * n))
    return v / h[:, None]


def directionally_stretched_metric(v, beta, c):
    zeta = soft_norm(v, c)
    A = torch.einsum('...i,...j->...ij', zeta, zeta)
    I = torch.eye(A.shape[-1]).expand(A.shape[0], -1, -1)
    return beta * A + (1 - beta) * I


# ---------------------------------------------------------------------------------------
# taskmap.py
class IdentityTaskmap:
    def forward(self, q):
        return q

    def differentiate(self, q, qd):
        return rmp_differentiate(self.forward)(q, qd)


class TaskmapByForwardKinematic:
    def __init__(self, fkine, frame):
        self.fkine, self.frame = fkine, frame

    def forward(self, q):
        return self.fkine.forward(q, self.frame)

    def differentiate(self, q, qd):
        return self.fkine.differentiate(q, qd, self.frame)


class TaskmapFrom4x4ToPosition:
    def forward(self, inp):
        return inp.reshape(-1, 4, 4)[:, :3, 3]

    def differentiate(self, q, qd):
        return rmp_differentiate(self.forward)(q, qd)


def euler_from_rotation_matrix(Rm):  # kinematics.py:74-96
    r00, r10, r21, r22, r20 = Rm[:, 0, 0], Rm[:, 1, 0], Rm[:, 2, 1], Rm[:, 2, 2], Rm[:, 2, 0]
    theta_y = -torch.asin(r20)
    cos_theta_y = torch.cos(theta_y)
    safe = torch.where(torch.abs(cos_theta_y) < 1e-6, torch.ones_like(cos_theta_y), cos_theta_y)
    theta_z = torch.atan2(r10 / safe, r00 / safe)
    theta_x = torch.atan2(r21 / safe, r22 / safe)
    return torch.stack((theta_x, theta_y, theta_z), dim=-1)


class TaskmapFrom4x4ToEuler:  # taskmap.py:57-67
    def forward(self, inp):
        return euler_from_rotation_matrix(inp.reshape(-1, 4, 4)[:, :3, :3])

    def differentiate(self, q, qd):
        return rmp_differentiate(self.forward)(q, qd)


class TaskmapJointFrame4x4ToDistance:
    def __init__(self, pos_on_link, pos_on_obs):
        self.pos_on_link = torch.as_tensor(pos_on_link, dtype=F32)
        self.pos_on_obs = torch.as_tensor(pos_on_obs, dtype=F32)

    def forward(self, inp):
        T = inp.reshape(-1, 4, 4)
        T = T.expand(self.pos_on_link.shape[0], 4, 4)
        pos_joint = T[:, :3, 3]
        rel = (self.pos_on_link - pos_joint).detach()  # tf.stop_gradient
        crit = pos_joint + rel
        return torch.linalg.norm(crit - self.pos_on_obs, dim=-1)[:, None]

    def differentiate(self, q, qd):
        B = self.pos_on_link.shape[0]
        return rmp_differentiate(self.forward)(q.repeat_interleave(B, dim=0), qd.repeat_interleave(B, dim=0))


class TaskmapRelative4x4:
    """taskmap.py:79-99: T_ref @ [I | relative_pos], one relative position per pair."""

    def __init__(self, relative_pos):
        self.relative_pos = torch.as_tensor(relative_pos, dtype=F32)

    def forward(self, inp):
        B = self.relative_pos.shape[0]
        T_ref = inp.reshape(-1, 4, 4)
        if T_ref.shape[0] != B:
            T_ref = T_ref.expand(B, 4, 4)
        T_rel = homogenous_transformation(torch.eye(3).expand(B, 3, 3), self.relative_pos)
        return (T_ref @ T_rel).reshape(-1, 16)

    def differentiate(self, q, qd):
        B = self.relative_pos.shape[0]
        return rmp_differentiate(self.forward)(q.repeat_interleave(B, dim=0), qd.repeat_interleave(B, dim=0))


class _Chained:
    def __init__(self, t1, t2):
        self.t1, self.t2 = t1, t2

    def forward(self, q):
        return self.t2.forward(self.t1.forward(q))

    def differentiate(self, q, qd):
        out_1, dout1, J_1, c_1 = self.t1.differentiate(q, qd)
        out_2, _, J_2, c_2 = self.t2.differentiate(out_1, dout1)
        dout = torch.einsum('bkm,bm->bk', J_2, dout1.expand(J_2.shape[0], -1))
        J = J_2 @ J_1
        c = c_2 + torch.einsum('bkm,bm->bk', J_2, c_1.expand(J_2.shape[0], -1))
        return out_2, dout, J, c


def chain_taskmaps(lst):
    ch = lst[0]
    for t in lst[1:]:
        ch = _Chained(ch, t)
    return ch


# ---------------------------------------------------------------------------------------
# leaves (rmp2.py / rmp.py).  P = parameter list in the reference constructor's order.
def target_attractor(P, goal, x, xd):
    kp, kd, eps, ell, amin, smax, smin, sb, ellb = P
    goal = torch.as_tensor(goal, dtype=F32)
    delta = goal - x
    dn = torch.linalg.norm(delta, dim=1)[:, None]
    soft = torch.maximum(dn, eps / 10 * torch.ones_like(dn))
    dhat = delta / soft
    xdd = kp * delta / (dn + eps) - kd * xd
    B, k = x.shape
    eye = torch.eye(k).expand(B, k, k)
    S = torch.einsum('bi,bj->bij', dhat, dhat)
    sd = dn / ell
    a = ((1. - amin) * torch.exp(-.5 * sd * sd) + amin)[..., None]
    metric = a * smax * eye + (1. - a) * smin * S
    bsd = dn / ellb
    ba = torch.exp(-.5 * bsd * bsd)
    boost = (ba * sb + (1. - ba) * 1.)[..., None]
    return xdd, boost * metric


def target_policy(P, goal, x, xd):
    alpha, beta_d, c = P
    goal = torch.as_tensor(goal, dtype=F32)

    def motion(x, xd):
        v = goal - x
        n = torch.linalg.norm(v)
        h = n + c * torch.log(1 + torch.exp(-2 * c * n))
        return alpha * (1 / h * v) - beta_d * xd
    f_attract = motion(x, xd)
    n = torch.linalg.norm(x - goal)
    beta = 1 - torch.exp(-0.5 * n ** 2 / 1 ** 2)
    H = directionally_stretched_metric(v=f_attract, c=c, beta=beta)
    w = torch.exp(-n / 3)
    return f_attract, w * H


def joint_velocity_cap(P, x, xd):
    vmax, region, gain, wgt = P
    cutoff = vmax - region
    dv = torch.abs(xd) - cutoff
    xdd = -torch.abs(gain * dv) * torch.sign(xd)
    clipped = torch.minimum(dv, torch.tensor(region - 1e-6, dtype=F32))
    ratio = clipped / region
    diag = torch.diag_embed(ratio ** 2)
    metric = wgt / (1.0 - diag)
    acc = torch.where(torch.abs(xd) < cutoff, torch.zeros_like(xdd), xdd)
    return acc, metric


def joint_damping(P, x, xd):
    kd, ms, inertia = P
    B, k = x.shape
    n = torch.linalg.norm(xd, dim=1, keepdim=True)
    acc = -(kd * n) * xd
    metric = torch.eye(k).expand(B, k, k) * ((ms * n)[..., None] + inertia)
    return acc, metric


def obstacle_avoidance(P, x, xd):
    margin, dgain, dstd, deps, gate_len, rgain, rstd, radius, mscal, estd, eeps = P
    x = x - margin
    x = torch.maximum(x, torch.zeros_like(x))
    base = mscal / (x / estd + eeps)
    gate = x * x / (radius * radius) - 2. * x / radius + 1.
    gate = torch.where(x > radius, torch.zeros_like(gate), gate)
    metric = base * gate
    repel = rgain * torch.exp(-(x / rstd))
    sig = torch.sigmoid(xd / gate_len)
    damp = -(1. - sig) * dgain * xd / (x / dstd + deps)
    accel = repel + damp
    metric = torch.where(x > radius, torch.zeros_like(metric), (1 - sig) * metric)
    return accel, metric[..., None]


def cspace_biasing(P, goal, x, xd):
    ms, kp, kd, thresh, inertia = P
    x = x - torch.as_tensor(goal, dtype=F32)
    B, k = x.shape
    xn = torch.linalg.norm(x, dim=1, keepdim=True)
    xhat = x / xn
    pos = torch.where(xn < thresh, -x * kp, -thresh * xhat * kp)
    vel = -kd * xd
    metric = torch.eye(k).expand(B, k, k) * (ms + inertia)
    return pos + vel, metric


def joint_limit_avoidance(P, lo, hi, q, qd):
    gp, gd = P
    lo, hi = torch.as_tensor(lo, dtype=F32), torch.as_tensor(hi, dtype=F32)
    d_upper = (hi - q) / (hi - lo)
    d_lower = (q - lo) / (hi - lo)
    d = torch.minimum(d_upper, d_lower)
    r = 0.15
    c_0, c_1, c_2, c_3 = 1, 0, -3 / r ** 2, 2 / r ** 3
    spline = c_3 * d ** 3 + c_2 * d ** 2 + c_1 * d + c_0
    w = torch.where(d > r, torch.zeros_like(spline), spline)
    qd_max = 20 * (2 * np.pi) / 60
    v = qd / qd_max
    H = directionally_stretched_metric(v, beta=0.9, c=5)
    A = w * H  # [1,n] * [1,n,n] -> scales COLUMNS (quirk Q2)
    return -gp * q - gd * qd, A


def config_space_biasing(P, q0, q, qd):
    gp, gd, w = P
    q0 = torch.as_tensor(q0, dtype=F32)
    return gp * (q0 - q) - gd * qd, w * torch.eye(q.shape[-1])[None]


# ---------------------------------------------------------------------------------------
# rmp.py: RmpCore
LEAF_FN = {1: "target_attractor", 2: "joint_velocity_cap", 3: "joint_damping", 4: "obstacle_avoidance",
           5: "cspace_biasing", 6: "target_policy", 7: "joint_limit_avoidance", 8: "config_space_biasing"}


def collision_avoidance(P, d, vec, x, xd):
    """rmp.py:264-315 (d, vec are the Datamanager's 'distance' / 'normal_vec')."""
    eta_rep, nu_rep, eta_damp, nu_damp, r, c = [float(p) for p in P]
    d = torch.as_tensor(d, dtype=F32)
    vec = torch.as_tensor(vec, dtype=F32)

    def motion(x, xd):
        alpha_rep = eta_rep * torch.exp(-d / nu_rep)
        f_rep = alpha_rep[:, None] * vec
        eps = torch.tensor(1e-6, dtype=F32)
        alpha_damp = eta_damp / (d / nu_damp + eps)
        scaling = torch.clamp_min(torch.einsum('...i,...i->...', -xd, vec), 0.)
        P_obs = torch.einsum('...,...i,...j->...ij', scaling, vec, vec)
        f_damp = alpha_damp[:, None] * torch.einsum('bij,bj->bi', P_obs, xd)
        return f_rep - f_damp

    c_2, c_3 = -3 / r ** 2, 2 / r ** 3
    spline = c_3 * d ** 3 + c_2 * d ** 2 + 0 * d + 1
    w = torch.where(d > r, torch.zeros_like(spline), spline)
    f_obs = motion(x, xd)
    H = directionally_stretched_metric(v=f_obs, c=c, beta=0)
    return f_obs, w[:, None, None] * H


def evaluate_one(fkine, leaves, q, qd, goal, pairs=None):
    """RmpCore.evaluate for ONE robot.  leaves: list of dicts(kind, taskmap, frame(name), params,
    vec_a, vec_b, goal_offset); pairs: {leaf_index: (p_link[B,3], p_obs[B,3])}, or
    (relative_pos[B,3], normal_vec[B,3], distance[B]) for an attached-point leaf (taskmap 3).
    Returns (qdd fp64 [n], M fp64, f fp64)."""
    n = len(q)
    f_comb, M_comb = np.zeros(n), np.zeros((n, n))
    qt, qdt = torch.tensor(q, dtype=F32), torch.tensor(qd, dtype=F32)
    for li, lf in enumerate(leaves):
        kind, tm = lf["kind"], lf["taskmap"]
        if tm == 0:
            taskmap = IdentityTaskmap()
        elif tm == 1:
            taskmap = chain_taskmaps([TaskmapByForwardKinematic(fkine, lf["frame"]), TaskmapFrom4x4ToPosition()])
        elif tm == 3:
            rel, nvec, dist = pairs[li]
            if len(rel) == 0:
                continue
            taskmap = chain_taskmaps([TaskmapByForwardKinematic(fkine, lf["frame"]), TaskmapRelative4x4(rel),
                                      TaskmapFrom4x4ToPosition()])
        else:
            pl, po = pairs[li]
            if len(pl) == 0:
                continue
            taskmap = chain_taskmaps([TaskmapByForwardKinematic(fkine, lf["frame"]),
                                      TaskmapJointFrame4x4ToDistance(pl, po)])
        x, xd, J, c = taskmap.differentiate(qt[None, :], qdt[None, :])
        P = lf["params"]
        g = None if lf.get("goal_offset", -1) < 0 else goal[lf["goal_offset"]:lf["goal_offset"] + x.shape[-1]]
        if kind == 1:
            xdd, A = target_attractor(P, g, x, xd)
        elif kind == 2:
            xdd, A = joint_velocity_cap(P, x, xd)
        elif kind == 3:
            xdd, A = joint_damping(P, x, xd)
        elif kind == 4:
            xdd, A = obstacle_avoidance(P, x, xd)
        elif kind == 5:
            xdd, A = cspace_biasing(P, lf["vec_a"], x, xd)
        elif kind == 6:
            xdd, A = target_policy(P, g, x, xd)
        elif kind == 7:
            xdd, A = joint_limit_avoidance(P, lf["vec_a"], lf["vec_b"], x, xd)
        elif kind == 8:
            xdd, A = config_space_biasing(P, lf["vec_a"], x, xd)
        elif kind == 9:
            xdd, A = collision_avoidance(P, dist, nvec, x, xd)
        else:
            raise ValueError(kind)
        Jt = J.transpose(1, 2)
        f = torch.einsum('bnk,bk->bn', Jt @ A, xdd - c)
        M = Jt @ A @ J
        f_comb += f.sum(dim=0).numpy()       # fp32 reduce_sum, fp64 accumulate (rmp.py:149-150)
        M_comb += M.sum(dim=0).numpy()
    # tf.linalg.pinv default rcond = 10 * max(rows, cols) * eps  [TF-doc]
    M_pinv = np.linalg.pinv(M_comb, rcond=10 * n * np.finfo(np.float64).eps)
    return M_pinv @ f_comb, M_comb, f_comb


def leaves_from_desc(desc, frame_names):
    """Decode an rmp2_desc ctypes struct into the plain dicts evaluate_one takes."""
    n = desc.robot.n_dof
    out = []
    for i in range(desc.n_leaves):
        lf = desc.leaves[i]
        npar = {1: 9, 2: 4, 3: 3, 4: 11, 5: 5, 6: 3, 7: 2, 8: 3, 9: 6}[lf.kind]
        out.append({"kind": lf.kind, "taskmap": lf.taskmap,
                    "frame": frame_names[lf.frame] if lf.frame >= 0 else None,
                    "params": [float(np.float32(lf.params[k])) for k in range(npar)],
                    "vec_a": [float(lf.vec_a[k]) for k in range(n)], "vec_b": [float(lf.vec_b[k]) for k in range(n)],
                    "goal_offset": lf.goal_offset})
    return out
