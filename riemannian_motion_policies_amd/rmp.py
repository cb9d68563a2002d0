"""`RmpCore` and the older leaf policies, with the reference's names (rmp.py:111-382).

RmpCore keeps the reference's registry protocol (add_rmp / remove_rmp_by_name / __str__,
rmp.py:114-131) but `evaluate` no longer loops over policies in Python: the whole set is
compiled to a flat descriptor once (re-compiled only when the set changes) and each call is
one fused HIP control step for all robots in `q`.
"""
from __future__ import annotations

import numpy as np
import torch

from . import descriptor as D
from ._native import ERR_UNSUPPORTED, Rmp2Error
from .data_management import ArrayVar, as_array, as_tensor
from .rmp2 import RiemannianMotionPolicy
from .taskmap import IdentityTaskmap, TaskmapSphereDistance, classify
from .urdf import KinematicTable


class QddResult(np.ndarray):
    """ndarray with the `.numpy()` the reference's callers use (06_cluttered_environment.py:124)."""

    def numpy(self):
        return np.asarray(self)


def _stage_together(arrays, device, limit_floats: int = 1 << 22):
    """Host arrays -> fp32 device tensors of the same shapes through ONE host-to-device copy (each at a 16-byte multiple of a
    common buffer); beyond `limit_floats` in total the copies are made one by one (nothing to gain, one host pass more)."""
    arrays = [np.ascontiguousarray(a, dtype=np.float32) for a in arrays]
    if sum(a.size for a in arrays) > limit_floats:
        return [torch.from_numpy(a).to(device) for a in arrays]
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.size + 3) & ~3
    buf = np.empty(total, dtype=np.float32)
    for a, o in zip(arrays, offs):
        buf[o:o + a.size] = a.reshape(-1)
    dev = torch.from_numpy(buf).to(device)
    return [dev[o:o + a.size].view(a.shape) for a, o in zip(arrays, offs)]


def _null_table(n_dof: int) -> KinematicTable:
    """Robot with no frames: identity-task-map-only sets (e.g. experiments/two_joint_robot/03)."""
    z = np.zeros
    return KinematicTable(frame_names=[], parent=z(0, np.int32), joint_type=z(0, np.int32), q_index=z(0, np.int32),
                          axis=z((0, 3), np.float32), T_const=z((0, 4, 4), np.float32), has_collision=z(0, bool),
                          order=[f"q{i}" for i in range(n_dof)], link_names=[], limits_lower=z(0, np.float32),
                          limits_upper=z(0, np.float32))


class _StageSource:
    """One call of RmpCore.update_distances: the inputs of the closest-point stage (snapshots: the caller may advance q or move
    the obstacles in place afterwards) and, once somebody asks for them, its output arrays."""

    def __init__(self, core, eng, q, single, prim, lc, n_leaves, primitive=None):
        self.core, self.eng, self.single, self.n_leaves = core, eng, single, n_leaves
        self.primitive = primitive    # None (by record size) | "cylinder"
        # "the q the stage was given, unmodified": the tensor OBJECT (holding it pins its storage: the allocator cannot hand the
        # address to a fresh tensor that would then pass for it) and its version counter (inference-mode tensors have none:
        # the fused route is then not taken)
        self._q_ref = q
        try:
            self._q_version = q._version
        except RuntimeError:
            self._q_version = None
        self.q = (q[None] if single else q).clone()
        self.prim = prim.clone()
        self.lc = None if lc is None else lc.clone()
        self.K = int(prim.shape[0])
        self._arrays = None
        self._zeros = None
        self.fusable = True    # (cleared when the engine refuses the fused form: its limits are the library's to state)

    def same_q(self, q) -> bool:
        if q is not self._q_ref or self._q_version is None:
            return False
        try:
            return q._version == self._q_version
        except RuntimeError:
            return False

    def link_capsules_or_origins(self):
        if self.lc is not None:
            return self.lc
        if self._zeros is None:   # a capsule of zero length and radius at the frame origin IS the frame origin
            self._zeros = torch.zeros((self.n_leaves, 8), dtype=torch.float32, device=self.eng.device)
        return self._zeros

    def arrays(self):
        if self._arrays is None:
            table = self.eng.obstacles(spheres=self.prim, primitive=self.primitive)
            pl, po = self.eng.closest_points(self.q, table, link_capsules=self.lc)
            self._arrays = (pl, po)
            self.core._pairs_cache = (pl, po, [self.K] * self.n_leaves)
        return self._arrays

    def view(self, i, which):
        a = self.arrays()[which][:, i * self.K:(i + 1) * self.K]
        return a[0] if self.single else a


class _LazyPairs:
    """{frame name: (p_link, p_obs)} of one update_distances call; reading an entry runs the stage."""

    def __init__(self, source, frames):
        self.source, self.frames = source, list(frames)

    def __getitem__(self, frame):
        i = self.frames.index(frame)
        return self.source.view(i, 0), self.source.view(i, 1)

    def __iter__(self):
        return iter(self.frames)

    def __len__(self):
        return len(self.frames)

    def keys(self):
        return list(self.frames)

    def items(self):
        return [(f, self[f]) for f in self.frames]


class RmpCore:
    """Manages multiple RMPs and resolves them into one joint acceleration (rmp.py:111-155).

    Differences to the reference, all additive: `evaluate` also accepts a fleet `q[R,n]`;
    `device` / `solve` select the GPU and the resolve mode ("auto": LU with pseudo-inverse
    fall-through, "pinv": always the reference's pseudo-inverse).  The reference's shared
    mutable default `rmps={}` (rmp.py:114) is NOT reproduced: each core owns its dict.
    """

    def __init__(self, rmps=None, device: int = 0, solve: str = "auto"):
        self.rmps = {} if rmps is None else rmps
        self.device = device
        self.solve = solve
        self.spheres = None  # shared sphere table [K,4] for TaskmapSphereDistance leaves
        self._engine = None
        self._signature = None
        self._quick = None
        self._pairs = []

    def __str__(self):
        out = ''
        if len(self.rmps) > 0:
            out += '\n' + 'used RMPs:' + '\n'
            for i, rmp in enumerate(self.rmps.values()):
                out += '\t'.join([str(i), rmp.name, str(type(rmp))]) + '\n'
        else:
            out += 'no RMPs in use.' + '\n'
        return out

    def add_rmp(self, rmp):
        self.rmps[rmp.name] = rmp

    def remove_rmp_by_name(self, name):
        self.rmps.pop(name)

    # ------------------------------------------------------------------------------
    def _quick_signature(self, n_dof):
        """What decides the compiled program, read from the leaves without building their specs: identities of the policy
        and task-map objects, scalar parameters, constant vectors.  (Goals and obstacle data are per-call inputs.)"""
        sig = [n_dof, self.solve]
        for rmp in self.rmps.values():
            va, vb = rmp._vectors()
            g = rmp._goal()
            sig.append((id(rmp), type(rmp).__name__, id(rmp.taskmap), tuple(rmp._params()),
                        None if va is None else np.asarray(va, dtype=np.float64).tobytes(),
                        None if vb is None else np.asarray(vb, dtype=np.float64).tobytes(),
                        None if g is None else int(g.shape[-1] if hasattr(g, "shape") else len(g)),
                        tuple((id(st), type(st).__name__, id(getattr(st, "fkine", None)), getattr(st, "frame", None))
                              for st in rmp.taskmap.stages())))
        return tuple(sig)

    def _compile(self, n_dof):
        quick = self._quick_signature(n_dof)
        if self._engine is not None and quick == self._quick:
            return self._engine
        fks = []
        for rmp in self.rmps.values():
            _, fk, _ = classify(rmp.taskmap)
            if fk is not None and all(fk.fkine is not f for f in fks):
                fks.append(fk.fkine)
        if len(fks) > 1:
            raise NotImplementedError("all FK task maps of one RmpCore must share one UrdfForwardKinematic")
        table = fks[0].table if fks else _null_table(n_dof)
        if table.n_dof != n_dof:
            raise ValueError(f"q has {n_dof} entries, the robot has {table.n_dof} dof")
        self._table = table
        specs = [rmp.leaf_spec(lambda fk: table.frame_index(fk.frame)) for rmp in self.rmps.values()]
        sig = (id(table), self.solve, tuple(s.signature() for s in specs))
        if sig != self._signature:
            from .engine import Engine
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(D.build_desc(table, specs, self.solve), self.device)
            self._signature = sig
        self._quick = quick
        self._pairs = [(rmp, kind, last) for rmp, (kind, _, last) in ((r, classify(r.taskmap)) for r in self.rmps.values())
                       if kind in (D.TASKMAP_FK_DISTANCE, D.TASKMAP_FK_POINT)]
        return self._engine

    def engine_for(self, q):
        """The compiled engine of the current policy set for joint vectors shaped like `q` ([n] or [R, n])."""
        return self._compile(int(q.shape[-1]))

    def _pair_leaves(self):
        """(policy, task-map kind, last stage) of the leaves that consume per-pair obstacle data, in leaf order (as of the last
        _compile: every entry point compiles first)."""
        return self._pairs

    def update_distances(self, q, primitives, link_capsules=None, primitive=None):
        """The closest-point preprocessing stage on the device (simulation.py:462-484 calculate_distances followed by
        data_management.py:16-31 update): for every TaskmapJointFrame4x4ToDistance leaf and every obstacle primitive
        ([K,4] spheres or [K,8] capsules) the nearest points of the link (its capsule from `link_capsules`, rows in leaf order;
        None: the frame origin) and the obstacle, as device tensors [R,K,3] ([K,3] for one robot) in the leaves' holders.
        LAZY: the arrays are formed when somebody reads a holder.  An evaluate(q, ...) on the same, unmodified q whose distance
        leaves still hold what this call gave them never forms them: it hands the primitives and the link capsules to the step,
        which forms the in-range pairs itself (rmp2_obstacles.link_capsules; same values, explicit-pair semantics).
        Returns a mapping {frame name: (p_link, p_obs)} (reading an entry runs the stage); .frames lists the names.
        ("Unmodified" is torch's version counter of q: in-place torch operations and Engine.rollout bump it; a write through
        the raw pointer by foreign code does not -- call update_distances again after one, as the reference's loop does every
        step.)"""
        single = q.dim() == 1 if isinstance(q, torch.Tensor) else np.ndim(q) == 1
        eng = self.engine_for(q)
        qt = as_tensor(q, eng.device)
        leaves = [(rmp, last) for rmp, kind, last in self._pair_leaves() if kind == D.TASKMAP_FK_DISTANCE
                  and not isinstance(last, TaskmapSphereDistance)]
        if not leaves:
            raise ValueError("update_distances: the policy set has no TaskmapJointFrame4x4ToDistance leaf")
        dist_idx = [i for i in range(eng.desc.n_leaves) if eng.desc.leaves[i].taskmap == D.TASKMAP_FK_DISTANCE]
        if len(leaves) != len(dist_idx):
            raise NotImplementedError("mixing explicit-pair and sphere distance task maps in one core")
        if len(dist_idx) != len(eng._dist_leaves):
            raise NotImplementedError("update_distances: attached-point leaves (TaskmapRelative4x4) carry their own pair data")
        prim = as_tensor(primitives, eng.device)
        lc = None if link_capsules is None else as_tensor(link_capsules, eng.device)
        src = _StageSource(self, eng, qt, single, prim, lc, len(leaves), primitive=primitive)
        self._stage = src
        names = self._table.frame_names
        frames = []
        for i, (rmp, last) in enumerate(leaves):
            for which, attr in enumerate(("pos_on_link_in_base_frame", "pos_on_obstacle_in_base_frame")):
                h = getattr(last, attr)
                if isinstance(h, ArrayVar):
                    h.assign_lazy(lambda i=i, which=which: src.view(i, which), owner=src)
                else:   # a plain attribute cannot defer: the stage runs now
                    setattr(last, attr, src.view(i, which))
            frames.append(names[eng.desc.leaves[dist_idx[i]].frame])
        return _LazyPairs(src, frames)

    def _fused_stage_obstacles(self, eng, q, pair_rmps):
        """The obstacle input of a step whose distance leaves all still hold what the last update_distances(q, ...) gave them,
        for the same unmodified q: the stage's INPUTS (primitives + link capsules) instead of its output.  None otherwise."""
        src = getattr(self, "_stage", None)
        if src is None or not src.fusable or src.eng is not eng or not src.same_q(q):
            return None
        for rmp, kind, last in pair_rmps:
            if kind != D.TASKMAP_FK_DISTANCE:
                return None
            for attr in ("pos_on_link_in_base_frame", "pos_on_obstacle_in_base_frame"):
                h = getattr(last, attr, None)
                if not (isinstance(h, ArrayVar) and h.owner is src):
                    return None
        return eng.obstacles(spheres=src.prim, link_capsules=src.link_capsules_or_origins(), primitive=src.primitive)

    def _evaluate_device(self, q, qd, spheres, link_capsules=None):
        """evaluate() for tensors that already live on the engine's device: nothing goes through the host, the result is a
        device tensor on the caller's stream.  Same gathering rules as below."""
        single = q.dim() == 1
        eng = self.engine_for(q)
        dev = eng.device
        q2, qd2 = (q[None] if single else q), (as_tensor(qd, dev)[None] if single else as_tensor(qd, dev))
        R = q2.shape[0]
        goals = [as_tensor(g, dev) for g in (rmp._goal() for rmp in self.rmps.values()) if g is not None]
        goal = None
        if goals:
            goal = torch.cat([g if g.dim() == 2 else g.expand(R, g.shape[0]) for g in goals], dim=1) \
                if any(g.dim() == 2 for g in goals) else torch.cat(goals)
        obstacles = None
        pair_rmps = self._pair_leaves()
        if pair_rmps:
            if all(isinstance(last, TaskmapSphereDistance) for _, _, last in pair_rmps):
                sp = spheres if spheres is not None else self.spheres
                if sp is None:
                    raise ValueError("TaskmapSphereDistance leaves need evaluate(..., spheres=[K,4])")
                obstacles = eng.obstacles(spheres=as_tensor(sp, dev),
                                          link_capsules=None if link_capsules is None else as_tensor(link_capsules, dev))
            elif (fused := self._fused_stage_obstacles(eng, q, pair_rmps)) is not None:
                try:
                    out = eng.step(q2, qd2, goal=goal, obstacles=fused)
                    return out[0] if single else out
                except Rmp2Error as e:   # beyond the fused form's limits (table size, resolve mode ...): the arrays then
                    if getattr(e, "code", None) != ERR_UNSUPPORTED:   # (the library's code, not the wording of its message)
                        raise
                    self._stage.fusable = False
                    return self._evaluate_device(q, qd, spheres, link_capsules)
            elif not any(isinstance(last, TaskmapSphereDistance) for _, _, last in pair_rmps):
                def fleet(a, nd):
                    a = as_tensor(a, dev)
                    return a if a.dim() == nd else a.expand((R,) + tuple(a.shape))
                pl, po, dd = [], [], []
                for rmp, kind, last in pair_rmps:
                    if kind == D.TASKMAP_FK_DISTANCE:
                        pl.append(fleet(last.pos_on_link_in_base_frame, 3))
                        po.append(fleet(last.pos_on_obstacle_in_base_frame, 3))
                        dd.append(None)
                    else:
                        pl.append(fleet(last.relative_pos, 3))
                        po.append(fleet(rmp.vec, 3))
                        dd.append(fleet(rmp.d, 2))
                    if pl[-1].shape != po[-1].shape or (dd[-1] is not None and dd[-1].shape != pl[-1].shape[:2]):
                        raise ValueError(f"{rmp.name}: pair arrays disagree in shape")
                has_point = any(d is not None for d in dd)
                if has_point:
                    dd = [d if d is not None else torch.zeros(a.shape[:2], dtype=torch.float32, device=dev) for d, a in zip(dd, pl)]
                counts = [a.shape[1] for a in pl]
                whole = self._whole_pair_arrays(pl, po, counts) if not has_point else None
                p_link, p_obs = whole if whole is not None else (torch.cat(pl, dim=1), torch.cat(po, dim=1))
                obstacles = eng.obstacles(p_link=p_link, p_obs=p_obs, dist=torch.cat(dd, dim=1) if has_point else None,
                                          pair_counts=counts)
            else:
                raise NotImplementedError("mixing explicit-pair and sphere distance task maps in one core")
        out = eng.step(q2, qd2, goal=goal, obstacles=obstacles)
        return out[0] if single else out

    def _whole_pair_arrays(self, pl, po, counts):
        """The holders are usually the per-leaf views update_distances handed out: then the arrays it wrote ARE the kernel's
        [R,P,3] input and nothing is gathered."""
        cache = getattr(self, "_pairs_cache", None)
        if cache is None or cache[2] != counts or pl[0].shape[0] != cache[0].shape[0]:
            return None
        off = 0
        for a, b, k in zip(pl, po, counts):
            wa, wb = cache[0][:, off:off + k], cache[1][:, off:off + k]
            if a.data_ptr() != wa.data_ptr() or b.data_ptr() != wb.data_ptr() or a.stride() != wa.stride() or b.stride() != wb.stride():
                return None
            off += k
        return cache[0], cache[1]

    def evaluate(self, q, qd, spheres=None, link_capsules=None):
        """q, qd: [n] (one robot, as in the reference) or [R, n].  Returns qdd of the same shape: a QddResult (host array) for
        host inputs, a device tensor -- no host hop anywhere -- when q is a tensor on the engine's device."""
        if isinstance(q, torch.Tensor) and q.is_cuda:
            return self._evaluate_device(q, qd, spheres, link_capsules)
        single = np.ndim(q) == 1
        q2 = np.atleast_2d(np.asarray(as_array(q), dtype=np.float32))
        qd2 = np.atleast_2d(np.asarray(as_array(qd), dtype=np.float32))
        R, n = q2.shape
        eng = self._compile(n)
        # goals are mutable attributes of the leaves (01_target_rmp_only.py:61-63): re-read every call
        goals = []
        per_robot = False
        for rmp in self.rmps.values():
            g = rmp._goal()
            if g is not None:
                g = np.asarray(as_array(g), dtype=np.float32)
                per_robot |= g.ndim == 2
                goals.append(g)
        goal = None
        if goals:
            goal = np.concatenate([g if g.ndim == 2 else np.broadcast_to(g, (R, g.shape[0])) for g in goals], axis=1) \
                if per_robot else np.concatenate(goals)
        # obstacle data: gather the holders of all pair-consuming leaves (leaf order)
        obstacles = None
        pair_rmps = self._pair_leaves()
        if pair_rmps:
            if all(isinstance(last, TaskmapSphereDistance) for _, _, last in pair_rmps):
                sp = spheres if spheres is not None else self.spheres
                if sp is None:
                    raise ValueError("TaskmapSphereDistance leaves need evaluate(..., spheres=[K,4])")
                obstacles = eng.obstacles(spheres=as_array(sp) if not isinstance(sp, torch.Tensor) else sp, link_capsules=link_capsules)
            elif not any(isinstance(last, TaskmapSphereDistance) for _, _, last in pair_rmps):
                def fleet(a, nd):
                    a = np.asarray(as_array(a), np.float32)
                    return a if a.ndim == nd else np.broadcast_to(a, (R,) + a.shape)
                pl, po, dd = [], [], []
                for rmp, kind, last in pair_rmps:
                    if kind == D.TASKMAP_FK_DISTANCE:   # closest-point pairs (taskmap.py:115-138)
                        pl.append(fleet(last.pos_on_link_in_base_frame, 3))
                        po.append(fleet(last.pos_on_obstacle_in_base_frame, 3))
                        dd.append(np.zeros(pl[-1].shape[:2], np.float32))
                    else:                               # attached points (taskmap.py:79-99, rmp.py:264-315)
                        pl.append(fleet(last.relative_pos, 3))
                        po.append(fleet(rmp.vec, 3))
                        dd.append(fleet(rmp.d, 2))
                    if not (pl[-1].shape == po[-1].shape and dd[-1].shape == pl[-1].shape[:2]):
                        raise ValueError(f"{rmp.name}: pair arrays disagree in shape")
                has_point = any(kind == D.TASKMAP_FK_POINT for _, kind, _ in pair_rmps)
                # one staging copy for everything the step reads (q, qd, goal, pair arrays): at R = 1 -- the reference's own
                # calling pattern -- five separate host-to-device copies cost more than the kernel
                parts = [q2, qd2] + ([goal] if goal is not None else []) + \
                        [np.concatenate(pl, axis=1), np.concatenate(po, axis=1)] + ([np.concatenate(dd, axis=1)] if has_point else [])
                dev_parts = _stage_together(parts, eng.device)
                q2, qd2 = dev_parts[0], dev_parts[1]
                k = 2
                if goal is not None:
                    goal, k = dev_parts[2], 3
                obstacles = eng.obstacles(p_link=dev_parts[k], p_obs=dev_parts[k + 1], dist=dev_parts[k + 2] if has_point else None,
                                          pair_counts=[a.shape[1] for a in pl])
            else:
                raise NotImplementedError("mixing explicit-pair and sphere distance task maps in one core")
        out = eng.step(q2, qd2, goal=goal, obstacles=obstacles)
        res = out.cpu().numpy()
        return (res[0] if single else res).view(QddResult)


# ---- older leaf policies (rmp.py:226-382) ----------------------------------------------

class TargetPolicy(RiemannianMotionPolicy):
    """rmp.py:226-260 (quirk Q8: c*log in h, 1/c*log in soft_norm)."""
    KIND = D.LEAF_TARGET_POLICY

    def __init__(self, alpha, beta, c, goal, taskmap, name='Target_RMP'):
        super().__init__(name, taskmap)
        self.goal = goal
        self.c = c
        self.alpha = alpha
        self.beta = beta
        self.sigma_H = 1
        self.sigma_w = 3

    def _params(self):
        return [self.alpha, self.beta, self.c]

    def _goal(self):
        return self.goal

    def _allowed_taskmaps(self):
        return (D.TASKMAP_FK_POSITION, D.TASKMAP_IDENTITY)


class CollisionAvoidance(RiemannianMotionPolicy):
    """rmp.py:264-315: repulsion + directional damping away from a data-fed obstacle direction, metric
    spline(d) * I.  `d` [B] / `vec` [B,3] are array holders (the Datamanager's 'distance' / 'normal_vec'),
    re-read at every evaluate; the task map must be the chain [FK(frame), TaskmapRelative4x4, 4x4->position]
    (experiments/two_joint_robot/05_obstacle_avoidance.py:51-61)."""
    KIND = D.LEAF_COLLISION_AVOIDANCE

    def __init__(self, d, vec, eta_rep, nu_rep, eta_damp, nu_damp, r, c, taskmap, name='collision_avoidance'):
        super().__init__(name, taskmap)
        self.d, self.vec = d, vec
        self.eta_rep, self.nu_rep, self.eta_damp, self.nu_damp, self.r, self.c = eta_rep, nu_rep, eta_damp, nu_damp, r, c

    def _params(self):
        return [self.eta_rep, self.nu_rep, self.eta_damp, self.nu_damp, self.r, self.c]

    def _allowed_taskmaps(self):
        return (D.TASKMAP_FK_POINT,)


class ConfigurationSpaceBiasing(RiemannianMotionPolicy):
    """rmp.py:318-347: PD controller towards q0 with metric w*I."""
    KIND = D.LEAF_CONFIG_SPACE_BIASING

    def __init__(self, gamma_p, gamma_d, q0, name, w=0.05):
        super().__init__(name, taskmap=IdentityTaskmap())
        self.gamma_p = gamma_p
        self.gamma_d = gamma_d
        self.q_0 = q0
        self.w = w

    def _params(self):
        return [self.gamma_p, self.gamma_d, self.w]

    def _vectors(self):
        return np.asarray(self.q_0, dtype=np.float32), None


class JointLimitAvoidance(RiemannianMotionPolicy):
    """rmp.py:349-382 (non-symmetric metric H*diag(w): quirk Q2, reproduced)."""
    KIND = D.LEAF_JOINT_LIMIT_AVOIDANCE

    def __init__(self, lower_limits, upper_limits, gamma_p, gamma_d, name='joint_limit_avoidance'):
        super().__init__(name, taskmap=IdentityTaskmap())
        self.lower_limits = np.asarray(lower_limits, dtype=np.float32)
        self.upper_limits = np.asarray(upper_limits, dtype=np.float32)
        self.gamma_p = gamma_p
        self.gamma_d = gamma_d

    def _params(self):
        return [self.gamma_p, self.gamma_d]

    def _vectors(self):
        return self.lower_limits, self.upper_limits
