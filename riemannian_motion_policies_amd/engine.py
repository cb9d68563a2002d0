"""Engine: one robot type + one RMP set on one MI355X, driven through the C ABI.

PyTorch is used for device memory, streams and (in fleet.py) torch.distributed only; all
arithmetic happens inside librmp2_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _native
from . import descriptor as D


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.ascontiguousarray(t, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _is_resident(t, device) -> bool:
    """A contiguous fp32 tensor on `device`: usable by the library as it is (no staging copy)."""
    return (isinstance(t, torch.Tensor) and t.device == device and t.dtype == torch.float32 and t.is_contiguous())


def _require_resident(device, **tensors) -> None:
    for name, t in tensors.items():
        if t is not None and not _is_resident(t, device):
            raise ValueError(f"{name} must be a contiguous fp32 tensor on {device}: a staging copy would be made on "
                             "torch's current stream (not ordered against an explicit launch stream) and, for a bound "
                             "launch, later in-place writes to the original would never be seen")


class Engine:
    def __init__(self, desc: D.Desc, device: int | torch.device = 0):
        if not torch.cuda.is_available():
            raise _native.Rmp2Error("no HIP device visible; the RMP2 engine has no CPU fallback")
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.desc = desc
        self.n_dof = desc.robot.n_dof
        self.n_frames = desc.robot.n_frames
        self._h = C.c_void_p()
        self._lib = _native.lib()
        self._fence_attached = False  # a bound launch attached a completion fence to the handle (bind(done_fence=))
        _native.check(self._lib.rmp2_create(C.byref(desc), self.device.index or 0, C.byref(self._h)))
        self._dist_leaves = D.distance_leaf_indices(desc)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rmp2_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_kernel(self) -> str:
        """Kernel (mapping of robots to lanes) the last step / rollout of this engine launched."""
        return self._lib.rmp2_last_kernel(self._h).decode()

    # ------------------------------------------------------------------------------
    def obstacles(self, *, spheres=None, p_link=None, p_obs=None, pair_counts: Optional[Sequence[int]] = None,
                  csr_offset=None, csr_index=None, dist=None, link_capsules=None, primitive: Optional[str] = None):
        """Build the per-step `rmp2_obstacles` struct from device tensors (kept alive by the result).
        link_capsules [n_distance_leaves, 8] = (a, radius, b, -) per distance leaf, in its frame's coordinates
        (urdf.link_capsules), with a shared table `spheres`: the control point of a pair is the nearest point of the link's
        capsule to the obstacle, formed inside the step (the fused form of closest_points(link_capsules=) + explicit pairs).
        Attached-point leaves (TaskmapRelative4x4 + CollisionAvoidance) take the same two arguments instead of the per-pair arrays
        (p_link = relative_position, p_obs = normal_vec, dist): their Datamanager fields are then formed inside the step from the
        closest points of link capsule and primitive, per control step -- the form that can roll out."""
        o = D.Obstacles()
        keep = []
        if link_capsules is not None:
            if spheres is None or p_link is not None:
                raise ValueError("link_capsules go with a primitive table: obstacles(spheres=..., link_capsules=...[, csr_offset=, csr_index=])")
            link_capsules = _f32(link_capsules, self.device)
            # (one capsule per leaf that consumes per-pair obstacle data -- distance leaves and attached-point leaves -- in leaf order)
            n_dist = len(D.distance_leaf_indices(self.desc))
            if tuple(link_capsules.shape) != (n_dist, 8):
                raise ValueError(f"link_capsules must be [{n_dist}, 8] (one capsule per distance / attached-point leaf, in leaf order)")
            o.link_capsules = link_capsules.data_ptr()
            keep.append(link_capsules)
        if p_link is not None:
            p_link, p_obs = _f32(p_link, self.device), _f32(p_obs, self.device)
            if p_link.shape != p_obs.shape or p_link.dim() != 3 or p_link.shape[2] != 3:
                raise ValueError("p_link / p_obs must both be [R, P, 3]")
            o.mode, o.n_pairs = D.OBS_EXPLICIT_PAIRS, p_link.shape[1]
            dl = self._dist_leaves
            if pair_counts is None:
                if not dl or o.n_pairs % len(dl):
                    raise ValueError("pair_counts required: pairs do not split evenly over the distance leaves")
                pair_counts = [o.n_pairs // len(dl)] * len(dl)
            if len(pair_counts) != len(dl) or sum(pair_counts) != o.n_pairs:
                raise ValueError("pair_counts must have one entry per distance leaf and sum to P")
            acc, k = 0, 0
            for i in range(self.desc.n_leaves + 1):
                o.pair_begin[i] = acc
                if i < self.desc.n_leaves and i in dl:
                    acc += int(pair_counts[k])
                    k += 1
            o.p_link, o.p_obs = p_link.data_ptr(), p_obs.data_ptr()
            keep += [p_link, p_obs]
            if dist is not None:   # attached-point leaves: p_link = relative_position, p_obs = normal_vec
                dist = _f32(dist, self.device)
                if tuple(dist.shape) != tuple(p_link.shape[:2]):
                    raise ValueError("dist must be [R, P]")
                o.dist = dist.data_ptr()
                keep.append(dist)
        elif spheres is not None:
            spheres = _f32(spheres, self.device)
            if spheres.dim() != 2 or spheres.shape[1] not in (4, 8):
                raise ValueError("spheres must be [K, 4] = (cx, cy, cz, radius) or, for capsules, "
                                 "[K, 8] = (ax, ay, az, radius, bx, by, bz, unused)")
            # 8-float records are capsules (a, radius, b, -) unless primitive="cylinder": (centre, radius, unit axis, half height),
            # the reference's flat-capped cylinder obstacles (simulation.py:245-261)
            if primitive not in (None, "sphere", "capsule", "cylinder"):
                raise ValueError("primitive must be 'sphere', 'capsule' or 'cylinder'")
            if primitive == "cylinder":
                if spheres.shape[1] != 8:
                    raise ValueError("cylinder records are [K, 8] = (cx, cy, cz, radius, ux, uy, uz, half_height)")
                o.primitive = D.PRIM_CYLINDER
            else:
                if primitive is not None and (primitive == "capsule") != (spheres.shape[1] == 8):
                    raise ValueError(f"primitive={primitive!r} does not go with records of {spheres.shape[1]} floats")
                o.primitive = D.PRIM_CAPSULE if spheres.shape[1] == 8 else D.PRIM_SPHERE
            o.n_spheres, o.spheres = spheres.shape[0], spheres.data_ptr()
            keep.append(spheres)
            if csr_offset is not None:
                csr_offset = csr_offset.to(device=self.device, dtype=torch.int32).contiguous()
                csr_index = csr_index.to(device=self.device, dtype=torch.int32).contiguous()
                if csr_index.numel() == 0:   # every list empty: the C ABI still wants a readable pointer (it cannot see the counts)
                    csr_index = torch.zeros(1, dtype=torch.int32, device=self.device)
                o.mode, o.csr_offset, o.csr_index = D.OBS_RAGGED_SPHERES, csr_offset.data_ptr(), csr_index.data_ptr()
                keep += [csr_offset, csr_index]
            else:
                o.mode = D.OBS_SHARED_SPHERES
        else:
            o.mode = D.OBS_NONE
        o._keep = keep
        return o

    def step(self, q: torch.Tensor, qd: torch.Tensor, goal: Optional[torch.Tensor] = None, obstacles=None,
             out: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
             M: Optional[torch.Tensor] = None, f: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """One control step for the R robots in q/qd ([R, n_dof] fp32 device tensors).
        Asynchronous on `stream` (default: torch's current stream).  With an explicit raw `stream` every tensor must
        already be a contiguous fp32 tensor on the engine's device (conversion copies would run on torch's current
        stream, unordered against `stream`)."""
        if stream is not None:
            _require_resident(self.device, q=q, qd=qd, goal=goal)
        if self._fence_attached:
            self._attach_fence(None)
        q, qd = _f32(q, self.device), _f32(qd, self.device)
        if q.dim() != 2 or q.shape[1] != self.n_dof or q.shape != qd.shape:
            raise ValueError(f"q and qd must be [R, {self.n_dof}], got {tuple(q.shape)} / {tuple(qd.shape)}")
        R = q.shape[0]
        goal_ptr, goal_stride = None, 0
        if self.desc.goal_floats:
            if goal is None:
                raise ValueError("this RMP set has goal-bearing leaves: pass goal")
            goal = _f32(goal, self.device)
            if goal.dim() == 1:
                if goal.shape[0] != self.desc.goal_floats:
                    raise ValueError(f"goal must have {self.desc.goal_floats} floats")
            elif tuple(goal.shape) == (R, self.desc.goal_floats):
                goal_stride = self.desc.goal_floats
            else:
                raise ValueError(f"goal must be [{self.desc.goal_floats}] or [R, {self.desc.goal_floats}]")
            goal_ptr = goal.data_ptr()
        if self._dist_leaves and (obstacles is None or obstacles.mode == D.OBS_NONE):
            raise ValueError("this RMP set has distance leaves: pass obstacles=engine.obstacles(...)")
        if out is None:
            out = torch.empty((R, self.n_dof), dtype=torch.float32, device=self.device)
        elif out.shape != q.shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp32 [R, n_dof] tensor")
        o = D.Outputs()
        o.qdd = out.data_ptr()
        if status is not None:
            assert status.dtype in (torch.int32, torch.uint32) and status.numel() == R
            o.status = status.data_ptr()
        if M is not None:
            assert M.dtype == torch.float64 and M.numel() == R * self.n_dof * self.n_dof
            o.M = M.data_ptr()
        if f is not None:
            assert f.dtype == torch.float64 and f.numel() == R * self.n_dof
            o.f = f.data_ptr()
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        obs = obstacles if obstacles is not None else None
        rc = self._lib.rmp2_step(self._h, q.data_ptr(), qd.data_ptr(), goal_ptr, goal_stride,
                                 C.byref(obs) if obs is not None else None, C.byref(o), R, s)
        _native.check(rc, self._h)
        return out

    def bind(self, q: torch.Tensor, qd: torch.Tensor, goal: Optional[torch.Tensor] = None, obstacles=None,
             out: Optional[torch.Tensor] = None, stream=None, done_fence=None):
        """Pre-validate and pre-marshal one step on FIXED device buffers (the usual control loop:
        the simulator writes q/qd in place, the engine writes qdd in place).  Returns
        (launch, out): `launch()` is a bare C-ABI call (~2 us of host time).  The launch reads the caller's buffers
        themselves, so q, qd and goal must be contiguous fp32 tensors on the engine's device (anything else would be
        copied once and the copy, not the caller's buffer, would be read forever after).
        `done_fence` (fleet._Fence): signalled by the launch's own completion (rmp2_set_step_fence) -- what the
        obstacle exchange needs to know before it overwrites the table this launch reads."""
        _require_resident(self.device, q=q, qd=qd, goal=goal if self.desc.goal_floats else None, out=out)
        out = self.step(q, qd, goal, obstacles=obstacles, out=out, stream=stream)  # validates + warms up
        R = q.shape[0]
        goal_ptr, goal_stride = None, 0
        keep = [q, qd, out, obstacles]
        if self.desc.goal_floats:
            goal_stride = 0 if goal.dim() == 1 else self.desc.goal_floats
            goal_ptr = goal.data_ptr()
            keep.append(goal)
        o = D.Outputs()
        o.qdd = out.data_ptr()
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        obs_ref = C.byref(obstacles) if obstacles is not None else None
        out_ref = C.byref(o)
        fn, h, qp, qdp = self._lib.rmp2_step, self._h, q.data_ptr(), qd.data_ptr()
        keep.append(o)
        if done_fence is None:
            def launch(_keep=keep):
                if self._fence_attached:
                    self._attach_fence(None)
                rc = fn(h, qp, qdp, goal_ptr, goal_stride, obs_ref, out_ref, R, s)
                if rc:
                    _native.check(rc, h)
        else:
            keep.append(done_fence)
            setf, fh = self._lib.rmp2_set_step_fence, done_fence._h

            def launch(_keep=keep):
                setf(h, fh)
                self._fence_attached = True
                rc = fn(h, qp, qdp, goal_ptr, goal_stride, obs_ref, out_ref, R, s)
                if rc:
                    _native.check(rc, h)
        return launch, out

    def _attach_fence(self, fence) -> None:
        self._lib.rmp2_set_step_fence(self._h, fence._h if fence is not None else None)
        self._fence_attached = fence is not None

    def obstacle_trajectory(self, tables: torch.Tensor, csr_offset=None, csr_index=None, primitive: Optional[str] = None):
        """Obstacle tables of a rollout with MOVING obstacles: `tables` [n_control_steps, K, 4] (spheres) or
        [n_control_steps, K, 8] (capsules); control step k of rollout(..., obstacles=this) reads tables[k]."""
        tables = _f32(tables, self.device)
        if tables.dim() != 3 or tables.shape[2] not in (4, 8):
            raise ValueError("tables must be [n_control_steps, K, 4] or [n_control_steps, K, 8]")
        o = self.obstacles(spheres=tables[0], csr_offset=csr_offset, csr_index=csr_index, primitive=primitive)
        o.spheres = tables.data_ptr()
        o._keep.append(tables)
        o._table_steps = int(tables.shape[0])
        return o

    def reserve(self, robots: int) -> None:
        """Pre-size the handle's device buffers for steps of up to `robots` robots (rmp2_reserve): afterwards a step never
        allocates -- needed before capturing a stream with a handle whose step is two kernels (solve = "pinv" without an
        inertia leaf, rank-deficient sets); a no-op for every other handle."""
        _native.check(self._lib.rmp2_reserve(self._h, int(robots)), self._h)

    def rollout(self, q: torch.Tensor, qd: torch.Tensor, goal: Optional[torch.Tensor] = None, obstacles=None,
                n_control_steps: int = 1, substeps: int = 10, dt: float = 0.01, out: Optional[torch.Tensor] = None,
                status: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """Closed-loop rollout in ONE launch: `n_control_steps` x (control step, then `substeps` semi-implicit
        Euler ticks of `dt` with qdd held).  q and qd (contiguous fp32 device tensors) are advanced IN PLACE;
        returns the last qdd.  obstacles = obstacle_trajectory(...): the obstacles move between control steps."""
        for t in (q, qd):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError("rollout needs contiguous fp32 CUDA tensors (they are updated in place)")
        if q.dim() != 2 or q.shape[1] != self.n_dof or q.shape != qd.shape:
            raise ValueError(f"q and qd must be [R, {self.n_dof}]")
        if self._fence_attached:
            self._attach_fence(None)
        R = q.shape[0]
        goal_ptr, goal_stride = None, 0
        if self.desc.goal_floats:
            if goal is None:
                raise ValueError("this RMP set has goal-bearing leaves: pass goal")
            goal = _f32(goal, self.device)
            goal_stride = 0 if goal.dim() == 1 else self.desc.goal_floats
            goal_ptr = goal.data_ptr()
        if self._dist_leaves and (obstacles is None or obstacles.mode == D.OBS_NONE):
            raise ValueError("this RMP set has distance leaves: pass obstacles=engine.obstacles(...)")
        if out is None:
            out = torch.empty((R, self.n_dof), dtype=torch.float32, device=self.device)
        elif not _is_resident(out, self.device) or tuple(out.shape) != (R, self.n_dof):
            raise ValueError("out must be a contiguous fp32 [R, n_dof] tensor on the engine's device")
        o = D.Outputs()
        o.qdd = out.data_ptr()
        if status is not None:
            if not (isinstance(status, torch.Tensor) and status.device == self.device and status.is_contiguous()
                    and status.dtype in (torch.int32, torch.uint32) and status.numel() == R):
                raise ValueError("status must be a contiguous int32 / uint32 tensor of R elements on the engine's device")
            o.status = status.data_ptr()
        table_steps = int(getattr(obstacles, "_table_steps", 0)) if obstacles is not None else 0
        if table_steps > 1 and table_steps != int(n_control_steps):
            raise ValueError(f"the obstacle trajectory holds {table_steps} tables, the rollout has {n_control_steps} control steps")
        cfg = D.RolloutCfg(int(n_control_steps), int(substeps), float(dt), table_steps)
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        rc = self._lib.rmp2_rollout(self._h, q.data_ptr(), qd.data_ptr(), goal_ptr, goal_stride,
                                    C.byref(obstacles) if obstacles is not None else None, C.byref(cfg), C.byref(o), R, s)
        _native.check(rc, self._h)
        # q and qd were advanced through their raw pointers: tell torch (version counters; an in-place operation on an empty
        # slice launches nothing) -- RmpCore's fused route after update_distances relies on "same q, unmodified"
        q[:0].zero_()
        qd[:0].zero_()
        return out

    def forward_kinematics(self, q: torch.Tensor) -> torch.Tensor:
        q = _f32(q, self.device)
        R = q.shape[0]
        T = torch.empty((R, self.n_frames, 4, 4), dtype=torch.float32, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _native.check(self._lib.rmp2_forward_kinematics(self._h, q.data_ptr(), T.data_ptr(), R, s), self._h)
        return T

    def closest_points(self, q: torch.Tensor, table, link_capsules=None):
        """Closest-point preprocessing stage on its own (simulation.py:462-484 calculate_distances):
        returns (p_link, p_obs), each [R, n_distance_leaves * K, 3], for the shared primitive `table`
        built by obstacles(spheres=...).  The pair arrays can be fed back as obstacles(p_link=, p_obs=).
        link_capsules [n_distance_leaves, 8] = (a, radius, b, -) in each leaf's frame coordinates: the control point of a
        pair is then the nearest point of the LINK's capsule to the obstacle (different per pair, as PyBullet reports it),
        not the frame origin."""
        q = _f32(q, self.device)
        R = q.shape[0]
        n_dist = sum(1 for i in range(self.desc.n_leaves) if self.desc.leaves[i].taskmap == D.TASKMAP_FK_DISTANCE)
        P = n_dist * int(table.n_spheres)
        p_link = torch.empty((R, P, 3), dtype=torch.float32, device=self.device)
        p_obs = torch.empty_like(p_link)
        s = torch.cuda.current_stream(self.device).cuda_stream
        lc_ptr = None
        if link_capsules is not None:
            link_capsules = _f32(link_capsules, self.device)
            if tuple(link_capsules.shape) != (n_dist, 8):
                raise ValueError(f"link_capsules must be [{n_dist}, 8] (one capsule per distance leaf, in leaf order)")
            lc_ptr = link_capsules.data_ptr()
        _native.check(self._lib.rmp2_closest_points_links(self._h, q.data_ptr(), C.byref(table), lc_ptr, p_link.data_ptr(),
                                                          p_obs.data_ptr(), R, s), self._h)
        return p_link, p_obs

    def differentiate(self, q: torch.Tensor, qd: torch.Tensor, frame: int):
        q, qd = _f32(q, self.device), _f32(qd, self.device)
        R, n = q.shape
        x, xd, c = (torch.empty((R, 16), dtype=torch.float32, device=self.device) for _ in range(3))
        J = torch.empty((R, 16, n), dtype=torch.float32, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _native.check(self._lib.rmp2_differentiate(self._h, q.data_ptr(), qd.data_ptr(), int(frame), x.data_ptr(),
                                                   xd.data_ptr(), J.data_ptr(), c.data_ptr(), R, s), self._h)
        return x, xd, J, c

    def differentiate_euler(self, q: torch.Tensor, qd: torch.Tensor, frame: int):
        """(x, xd, J, c) of the chain [FK(frame), TaskmapFrom4x4ToEuler]: x, xd, c [R,3], J [R,3,n]."""
        q, qd = _f32(q, self.device), _f32(qd, self.device)
        R, n = q.shape
        x, xd, c = (torch.empty((R, 3), dtype=torch.float32, device=self.device) for _ in range(3))
        J = torch.empty((R, 3, n), dtype=torch.float32, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        _native.check(self._lib.rmp2_differentiate_euler(self._h, q.data_ptr(), qd.data_ptr(), int(frame), x.data_ptr(),
                                                         xd.data_ptr(), J.data_ptr(), c.data_ptr(), R, s), self._h)
        return x, xd, J, c


def bind_pair(eng_a: "Engine", q_a, qd_a, goal_a, obstacles_a, out_a, eng_b: "Engine", q_b, qd_b, goal_b, obstacles_b, out_b,
              stream=None):
    """Pre-marshal the control steps of TWO engines (two robot types of one shard) as one C-ABI call, rmp2_step_pair: one fused
    grid where the library has an instantiation for the pair (include/rmp2.h), two launches otherwise.  Buffers as for
    Engine.bind (contiguous fp32 tensors on the engines' device, read and written in place).  Returns launch()."""
    if eng_a.device != eng_b.device:
        raise ValueError("bind_pair: both engines must live on one device")
    for eng, q, qd, goal, obs, out in ((eng_a, q_a, qd_a, goal_a, obstacles_a, out_a), (eng_b, q_b, qd_b, goal_b, obstacles_b, out_b)):
        _require_resident(eng.device, q=q, qd=qd, goal=goal if eng.desc.goal_floats else None, out=out)
        eng.step(q, qd, goal, obstacles=obs, out=out, stream=stream)   # validates the arguments (and warms the kernels up)
    lib = eng_a._lib
    outs, args = [], []
    for eng, q, qd, goal, obs, out in ((eng_a, q_a, qd_a, goal_a, obstacles_a, out_a), (eng_b, q_b, qd_b, goal_b, obstacles_b, out_b)):
        o = D.Outputs()
        o.qdd = out.data_ptr()
        outs.append(o)
        gp = goal.data_ptr() if eng.desc.goal_floats else None
        gs = 0 if (not eng.desc.goal_floats or goal.dim() == 1) else eng.desc.goal_floats
        args += [eng._h, q.data_ptr(), qd.data_ptr(), gp, gs, C.byref(obs) if obs is not None else None, C.byref(o), q.shape[0]]
    s = stream if stream is not None else torch.cuda.current_stream(eng_a.device).cuda_stream
    keep = (q_a, qd_a, goal_a, obstacles_a, out_a, q_b, qd_b, goal_b, obstacles_b, out_b, outs)
    fn = lib.rmp2_step_pair

    def launch(_keep=keep):
        rc = fn(*args, s)
        if rc:
            _native.check(rc, eng_a._h)
    return launch
