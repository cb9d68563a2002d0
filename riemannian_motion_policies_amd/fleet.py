"""Multi-GPU fleet: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI),
robots sharded contiguously, outputs stay sharded.

Robots are independent units (rmp.py:133-155 touches only its own q, qd), so the data path
needs NO collective except when the shared obstacle table is produced distributed: then each
rank owns K/world spheres and the table is all-gathered once per step.  The
message is <= 1 KB, i.e. latency-bound: it is issued on a side stream and joined by an event
right before the control-step kernel, so it overlaps whatever the caller queued before.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split; the first `total % world` ranks get one extra robot."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def balanced_bounds(weights, world: int):
    """Config 5: split so that sum(weights) (~ pairs per robot) is balanced, not the robot count.
    Returns world+1 cut indices into the (type-sorted) robot list."""
    import numpy as np
    csum = np.concatenate([[0.0], np.cumsum(np.asarray(weights, dtype=np.float64))])
    cuts = [int(np.searchsorted(csum, csum[-1] * r / world, side="left")) for r in range(world)] + [len(weights)]
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts


class _Fence:
    """Device-scope stream fence (include/rmp2.h rmp2_fence_*): a HIP event without timing and without the system-scope
    release of a default event (torch.cuda.Event is one).  Between two kernels of one stream a default event's record
    costs ~7 us on MI355X; this one ~1 us.  Only for ordering work of ONE GPU."""

    def __init__(self, device: torch.device):
        import ctypes
        from . import _native
        self._lib = _native.lib()
        self._h = ctypes.c_void_p()
        rc = self._lib.rmp2_fence_create(device.index or 0, ctypes.byref(self._h))
        if rc != 0:
            raise RuntimeError("rmp2_fence_create: " + self._lib.rmp2_last_error(None).decode())

    def record(self, stream: "torch.cuda.Stream") -> None:
        if self._lib.rmp2_fence_record(self._h, stream.cuda_stream) != 0:
            raise RuntimeError("rmp2_fence_record: " + self._lib.rmp2_last_error(None).decode())

    def wait(self, stream: "torch.cuda.Stream") -> None:
        if self._lib.rmp2_fence_wait(self._h, stream.cuda_stream) != 0:
            raise RuntimeError("rmp2_fence_wait: " + self._lib.rmp2_last_error(None).decode())

    def __del__(self):
        try:
            self._lib.rmp2_fence_destroy(self._h)
        except Exception:
            pass


class _SystemFence:
    """Stream fence WITH the system-scope release / acquire of a default HIP event (torch.cuda.Event), same interface as
    _Fence.  What the consumer of an all-gathered table needs at world > 1: the table is written by PEER GPUs over xGMI,
    past this GPU's L2, and a device-scope fence would let the next kernel read lines its L2 still holds from two steps
    ago (round-2 advisor finding)."""

    def __init__(self, device: torch.device):
        self._ev = torch.cuda.Event()

    def record(self, stream: "torch.cuda.Stream") -> None:
        self._ev.record(stream)

    def wait(self, stream: "torch.cuda.Stream") -> None:
        stream.wait_event(self._ev)


class ObstacleExchange:
    """All-gather of the per-rank slices of the shared sphere table [K, 4] (RCCL, side stream).

    Two table buffers: the gather for step k + 1 can be in flight while the kernel of step k reads the other buffer
    (`start` right after launching the step; `finish` at the top of the next one).  Ordering is by events only:
    the gather waits for the producer of `local` and for the last kernel that read the buffer it overwrites; the
    consuming step waits for the gather.

    Deployment note: HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); when the side
    stream shares a queue with the compute stream the gather serialises behind the kernel it should overlap (+20 us per
    step measured).  Export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises (bench.py does)."""

    def __init__(self, spheres_per_rank: int, device, group=None, collective=None):
        # collective=False: a one-rank exchange (the slice is the table, a local copy) inside a job whose default group is larger --
        # what bench.py's world-1 leg of an N-rank run needs; no rank of the job is involved.
        self.group = group
        self.collective = (dist.is_available() and dist.is_initialized()) if collective is None else bool(collective)
        self.world = dist.get_world_size(group) if self.collective else 1
        self.rank = dist.get_rank(group) if self.collective else 0
        self.device = torch.device(device)
        cuda = self.device.type == "cuda"
        self.tables = [torch.zeros((self.world * spheres_per_rank, 4), dtype=torch.float32, device=self.device)
                       for _ in range(2)]
        # (highest priority: the gather's few workgroups are dispatched ahead of the next step kernel's, which fill the GPU)
        self.side = torch.cuda.Stream(self.device, priority=-1) if cuda else None
        # (fences, not torch events: a default event between two step kernels costs ~7 us of the step, see _Fence)
        fences = cuda and self.collective
        # ready[b]: "table b has been gathered".  At world 1 the gather is a local copy and a device-scope fence orders it;
        # at world > 1 peers wrote the table over xGMI, so the consumer needs the system-scope release / acquire of a
        # default event (the write-after-read side below, reader_done, orders work of THIS GPU only and keeps the cheap fence)
        ready_cls = _Fence if self.world == 1 else _SystemFence
        self.ready = [ready_cls(self.device) for _ in range(2)] if fences else None
        self.reader_done = [None, None]   # fence after the last kernel that read buffer b
        self._reader_events = [_Fence(self.device) for _ in range(2)] if fences else None
        self._next, self._pending, self._last = 0, [], None

    @property
    def table(self) -> torch.Tensor:
        """The most recently finished table."""
        return self.tables[self._last if self._last is not None else 0]

    def start(self, local: torch.Tensor, produced=None) -> None:
        """Issue the all-gather of `local` [K/world, 4] into the free buffer.  `produced`: event after which `local`
        is valid (default: everything issued so far on the current stream).  With a process group the collective
        is issued even for one rank (the RCCL path is then the one exercised on a one-GPU box); without one the
        slice is the table."""
        if len(self._pending) >= 2:   # (checked before anything is mutated: a caller may catch this and finish())
            raise RuntimeError("ObstacleExchange has two table buffers: finish() a gather before starting a third")
        b = self._next
        self._next ^= 1
        self._pending.append(b)
        if not self.collective:
            self.tables[b].copy_(local)
            return
        if self.side is not None:
            if isinstance(produced, _Fence):
                produced.wait(self.side)
            elif produced is not None:
                self.side.wait_event(produced)
            else:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
            if self.reader_done[b] is not None:
                self.reader_done[b].wait(self.side)
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.tables[b], local.contiguous(), group=self.group)
                self.ready[b].record(self.side)
        else:
            dist.all_gather_into_tensor(self.tables[b], local.contiguous(), group=self.group)

    def finish(self) -> torch.Tensor:
        """Make the current stream wait for the oldest outstanding gather; returns its table."""
        b = self._pending.pop(0)
        self._last = b
        if self.collective and self.side is not None:
            self.ready[b].wait(torch.cuda.current_stream(self.device))
        return self.tables[b]

    def reader_fence(self, table: torch.Tensor):
        """The fence the exchange waits on before it overwrites `table` (one of self.tables): hand it to
        Engine.bind(done_fence=) so that the reading launch signals it itself, then call consumed(attached=True)."""
        if self._reader_events is None:
            return None
        return self._reader_events[[t.data_ptr() for t in self.tables].index(table.data_ptr())]

    def consumed(self, attached: bool = False) -> None:
        """Call after launching the kernel that reads the table returned by the last finish().  attached=True: that
        launch carried reader_fence(table) as its completion fence -- nothing to record here."""
        if self.collective and self.side is not None and self._last is not None:
            ev = self._reader_events[self._last]
            if not attached:
                ev.record(torch.cuda.current_stream(self.device))
            self.reader_done[self._last] = ev


def rccl_library_path() -> str:
    """The librccl.so this process already uses: PyTorch's own copy (torch/lib), else the ROCm installation's."""
    import os
    cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"]
    for c in cand:
        if os.path.exists(c):
            return c
    raise RuntimeError("no librccl.so found next to torch or under /opt/rocm/lib")


class NativeObstacleExchange:
    """ObstacleExchange with the all-gather, the stream orderings and the step launch inside librmp2_hip.so
    (include/rmp2.h rmp2_exchange_*): ONE C-ABI call per control step.  The Python-driven exchange above costs ~50 us of
    host time per step (the c10d collective call alone ~20 us) -- more than the 44 us the step kernel takes, so a config-4
    rank driven through it is host bound.  The RCCL communicator is the library's own (ncclCommInitRank on a unique id that
    rank 0 creates and torch.distributed broadcasts); with no process group it is a communicator of one rank.

        exch = NativeObstacleExchange(spheres_per_rank, device)      # collective over the default group
        exch.start(local0)                                           # table of step 0
        for k in ...: qdd = exch.step(engine, q, qd, goal, out, next_local=local_k_plus_1)
    """

    def __init__(self, spheres_per_rank: int, device, group=None, depth: int = 1, *, rank: Optional[int] = None,
                 world: Optional[int] = None, uid: Optional[bytes] = None, rccl_library: Optional[str] = None):
        """Collective over `group` (the default process group when one exists): rank 0 creates the RCCL unique id,
        torch.distributed broadcasts it, every rank joins the communicator.
        rank / world / uid (all three): join a communicator WITHOUT torch.distributed -- the caller ships the id (from
        `NativeObstacleExchange.unique_id`) to the ranks itself, e.g. several ranks as threads of one process.
        rccl_library: path of the collective library to bind (default: the librccl.so this process already uses)."""
        import ctypes
        from . import _native
        self._lib = _native.lib()
        self._ctypes = ctypes
        self.device = torch.device(device)
        self.depth = int(depth)
        if self.device.type != "cuda":
            raise ValueError("NativeObstacleExchange needs a HIP device (the CPU test path uses ObstacleExchange)")
        self.spheres_per_rank = int(spheres_per_rank)
        explicit = rank is not None or world is not None or uid is not None
        if explicit and (rank is None or world is None or uid is None):
            raise ValueError("rank, world and uid go together")
        collective = not explicit and dist.is_available() and dist.is_initialized()
        self.world = int(world) if explicit else (dist.get_world_size(group) if collective else 1)
        self.rank = int(rank) if explicit else (dist.get_rank(group) if collective else 0)
        # -- local part (may fail on one rank only): resolve the library, rank 0 creates the id --
        local_error = None
        try:
            path = (rccl_library or rccl_library_path()).encode()
            if not explicit and self.rank == 0:
                uid = self.unique_id(path.decode())
        except Exception as exc:   # noqa: BLE001 -- agreed on below, re-raised on every rank
            local_error, path = exc, b""
        if collective and self.world > 1:
            # every rank learns whether ALL ranks got this far BEFORE anyone enters a blocking collective on the id: a rank
            # that raised here would otherwise leave its peers waiting in the broadcast / in ncclCommInitRank for ever
            ok = torch.tensor([0 if local_error is not None else 1], dtype=torch.int32, device=self.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                raise _native.Rmp2Error("NativeObstacleExchange: a rank failed before the communicator was formed"
                                        + (f" (this rank: {local_error})" if local_error is not None else " (another rank)"))
            t = torch.frombuffer(bytearray(uid if self.rank == 0 else bytes(128)), dtype=torch.uint8).to(self.device)
            dist.broadcast(t, 0, group=group)
            uid = bytes(t.cpu().numpy().tobytes())
        elif local_error is not None:
            raise local_error
        if len(uid) != 128:
            raise ValueError("uid must be the 128 bytes of rmp2_exchange_unique_id")
        uid_buf = (ctypes.c_char * 128).from_buffer_copy(bytes(uid))
        self._h = ctypes.c_void_p()
        rc = self._lib.rmp2_exchange_create(path, uid_buf, self.rank, self.world, self.device.index or 0,
                                            self.spheres_per_rank, ctypes.byref(self._h))
        if rc != 0:
            raise _native.Rmp2Error("rmp2_exchange_create: " + self._lib.rmp2_last_error(None).decode())
        self._table = ctypes.c_void_p()
        if self.depth != 1:   # (depth + 1 gathers outstanding: include/rmp2.h rmp2_exchange_set_depth)
            self._check(self._lib.rmp2_exchange_set_depth(self._h, self.depth))

    @staticmethod
    def unique_id(rccl_library: Optional[str] = None) -> bytes:
        """The 128 bytes of a fresh communicator id (rmp2_exchange_unique_id): rank 0 creates it, every rank joins on it."""
        import ctypes
        from . import _native
        lib = _native.lib()
        buf = (ctypes.c_char * 128)()
        rc = lib.rmp2_exchange_unique_id((rccl_library or rccl_library_path()).encode(), buf)
        if rc != 0:
            raise _native.Rmp2Error("rmp2_exchange_unique_id: " + lib.rmp2_last_error(None).decode())
        return bytes(buf)

    @property
    def nranks(self) -> int:
        """Ranks of the communicator the library joined (from the exchange object, not from the launcher's environment)."""
        return int(self._lib.rmp2_exchange_nranks(self._h))

    def _check(self, rc):
        if rc != 0:
            from . import _native
            raise _native.Rmp2Error(f"rmp2_exchange error {rc}: " + self._lib.rmp2_exchange_last_error(self._h).decode())

    def set_peer_wait(self, on: bool = True) -> None:
        """A one-rank exchange keeps the GPU-side wait an N-rank one needs (single-GPU emulation of an N-rank run)."""
        self._check(self._lib.rmp2_exchange_set_peer_wait(self._h, 1 if on else 0))

    @property
    def pending(self) -> int:
        """Gathers started and not yet consumed by a step."""
        return int(self._lib.rmp2_exchange_pending(self._h))

    def start(self, local: torch.Tensor, stream=None, local_is_ready: bool = False) -> None:
        """Issue the gather of `local` [spheres_per_rank, 4] (contiguous fp32 on the device; it must stay untouched until the
        gather has run).  local_is_ready: the caller guarantees `local` is complete now (no ordering event on `stream`)."""
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        self._check(self._lib.rmp2_exchange_start(self._h, local.data_ptr(), 1 if local_is_ready else 0, s))

    def bind(self, engine, q, qd, goal, out, next_local=None, stream=None, next_local_is_ready: bool = False):
        """Pre-marshal the per-step call on FIXED buffers: returns launch() = one C-ABI call (wait for the oldest gathered
        table, issue the gather of `next_local`, launch the step with the reader fence as its completion signal)."""
        from . import descriptor as D
        C = self._ctypes
        o = D.Outputs()
        o.qdd = out.data_ptr()
        s = stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream
        goal_ptr = goal.data_ptr() if goal is not None else None
        goal_stride = 0 if (goal is None or goal.dim() == 1) else engine.desc.goal_floats
        fn, xh, eh = self._lib.rmp2_exchange_step, self._h, engine._h
        qp, qdp, R = q.data_ptr(), qd.data_ptr(), q.shape[0]
        nl = next_local.data_ptr() if next_local is not None else None
        nlr = 1 if next_local_is_ready else 0
        oref, tref = C.byref(o), C.byref(self._table)
        keep = (q, qd, goal, out, next_local, o)

        def launch(_keep=keep):
            rc = fn(xh, eh, qp, qdp, goal_ptr, goal_stride, nl, nlr, oref, R, s, tref)
            if rc:
                self._check(rc)
        return launch

    def step(self, engine, q, qd, goal, out, next_local=None, stream=None, next_local_is_ready: bool = False):
        self.bind(engine, q, qd, goal, out, next_local, stream, next_local_is_ready)()
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rmp2_exchange_destroy(self._h)
            self._h = self._ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def agree_on_exchange(make, world: int, device=None, group=None):
    """Every rank calls `make()` (e.g. the NativeObstacleExchange constructor); the ranks then agree (all-reduce MIN over
    `group`) on whether ALL of them succeeded.  Returns (exchange, None) on every rank if so; otherwise (None, error) on
    every rank -- a rank whose own `make()` succeeded closes what it built and reports the peers' failure --, so that no rank
    falls back to another exchange alone and leaves its peers waiting in a collective."""
    exch, err = None, None
    try:
        exch = make()
    except Exception as e:   # noqa: BLE001 -- whatever kept this rank from building its exchange is agreed on below
        err = e
    ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=device)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok.item()) == 0:
        if exch is not None:
            try:
                exch.close()
            finally:
                exch = None
        return None, err if err is not None else RuntimeError("another rank could not build its exchange")
    return exch, None


class Fleet:
    """Engine + obstacle exchange for this rank's shard."""

    def __init__(self, desc, device: int, spheres_per_rank: int = 0, group=None):
        from .engine import Engine
        self.engine = Engine(desc, device)
        self.exchange = ObstacleExchange(spheres_per_rank, torch.device("cuda", device), group) if spheres_per_rank else None

    def step(self, q, qd, goal=None, local_spheres: Optional[torch.Tensor] = None, out=None):
        obstacles = None
        if self.exchange is not None:
            self.exchange.start(local_spheres)
            obstacles = self.engine.obstacles(spheres=self.exchange.finish())
        res = self.engine.step(q, qd, goal, obstacles=obstacles, out=out)
        if self.exchange is not None:
            self.exchange.consumed()
        return res


class MixedFleet:
    """BASELINE config 5: a fleet of several robot TYPES on one GPU.  Robots are stably
    partitioned by type so that every wavefront is type-homogeneous; each type has its own engine
    (its own compiled program); results are scattered back into the caller's robot order.

        fleet = MixedFleet({"two_joint": desc_a, "panda": desc_b}, device=0)
        qdd = fleet.step(type_of_robot, {"two_joint": (q, qd, goal, obstacle_kwargs), "panda": (...)})
    """

    def __init__(self, descs: dict, device: int = 0):
        from .engine import Engine
        self.engines = {k: Engine(d, device) for k, d in descs.items()}

    def step(self, robot_types, per_type_inputs):
        """robot_types: sequence of type keys, one per robot of the fleet (caller order).
        per_type_inputs[key] = (q[Rk,n], qd[Rk,n], goal, obstacle_kwargs) for the robots of that type
        in caller order.  Returns {key: qdd[Rk,n]} plus, under "index", the caller positions."""
        import numpy as np
        types = np.asarray(robot_types)
        out = {"index": {}}
        for key, eng in self.engines.items():
            idx = np.nonzero(types == key)[0]
            out["index"][key] = idx
            if idx.size == 0:
                continue
            q, qd, goal, okw = per_type_inputs[key]
            obstacles = eng.obstacles(**okw) if okw else None
            out[key] = eng.step(q, qd, goal, obstacles=obstacles)
        return out


class MixedFleetShard:
    """One rank's shard of BASELINE config 5 across `world` GPUs: the fleet is stably partitioned by robot
    TYPE (wavefronts stay type-homogeneous), then cut into `world` contiguous shards so that the estimated COST is
    balanced, not the robot count.  The cost of a robot is `per_robot[type] + per_pair[type] x pairs` in NANOSECONDS OF
    KERNEL TIME, calibrated from short timed launches of each type's engine (calibrate_costs below; round 2 weighed by
    the flop model of SURVEY 8(d), 1 : 2.9 TwoJoint : Panda, where the measured per-robot times are 1 : 1.7 -- the
    TwoJoint ranks would have been the stragglers).  A shard holds robots of one type or, where a cut falls inside a
    type's range, of two; it drives one engine per type present.  Robots are independent: no data-path collective."""

    BASE_FLOPS = {"two_joint": 0.5e3, "panda": 4.0e3}      # BASELINE.md section 3, config 5 (roofline accounting only)
    CONTROL_POINTS = {"two_joint": 3, "panda": 8}
    BYTES = {"two_joint": 36, "panda": 120}
    # ns of kernel time per robot and per (control point, obstacle) pair at throughput fleet sizes on one MI355X
    # (profiles/r03_cost_calibration.json, tools/calibrate_costs.py).  Every rank must cut with the SAME numbers: a
    # caller that re-calibrates at run time measures on rank 0 and broadcasts (bench.py does).
    DEFAULT_COST = {"two_joint": (0.55, 0.0160), "panda": (0.95, 0.0098)}

    @staticmethod
    def calibrate_costs(device: int, robots: int = 16384, seed: int = 5, steps: int = 200):
        """Measure {type: (ns per robot, ns per pair)} on `device`: each type's engine at `robots` robots with every robot
        seeing 0 obstacles and all K of them (two timed launches per type; the kernel's time is linear in the pair count
        between those, and near-linear in the robot count at throughput sizes)."""
        import numpy as np
        from . import configs as Cf
        from .engine import Engine
        dev = torch.device("cuda", device)
        K = Cf.N_SPHERES
        out = {}
        for key, builder, sampler in (("two_joint", Cf.config5_two_joint, Cf.sample_two_joint_states),
                                      ("panda", Cf.config3, Cf.sample_panda_states)):
            _, desc = builder("auto")
            eng = Engine(desc, device)
            st = sampler(np.random.default_rng([seed, 99]), robots)
            q, qd, goal = (torch.from_numpy(st[x]).to(dev) for x in ("q", "qd", "goal"))
            sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng([seed, 0 if key == "two_joint" else robots]))).to(dev)
            times = []
            for k in (0, K):
                off = torch.arange(robots + 1, dtype=torch.int32) * k
                idx = torch.arange(K, dtype=torch.int32).repeat(robots)[: robots * k] if k else torch.zeros(1, dtype=torch.int32)
                obs = eng.obstacles(spheres=sph, csr_offset=off, csr_index=idx)
                launch, _ = eng.bind(q, qd, goal, obstacles=obs)
                for _ in range(20):
                    launch()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(steps):
                    launch()
                b.record()
                torch.cuda.synchronize(dev)
                times.append(a.elapsed_time(b) * 1e6 / steps)   # ns per launch
            per_robot = times[0] / robots
            per_pair = max(times[1] - times[0], 0.0) / (robots * K * MixedFleetShard.CONTROL_POINTS[key])
            out[key] = (per_robot, per_pair)
        return out

    # Kernel time is NOT linear in the robot count: a mapping fills the GPU in rounds (16 384 robots put one wave on every
    # SIMD), so 21 700 Pandas cost as much as 32 768.  The cut below therefore works on measured TIME CURVES per type:
    # us per step of the type's engine at a ladder of fleet sizes, with the fleet's own list-length distribution
    # (calibrate_curves; tracked copy: profiles/r03_cost_calibration.json).
    CURVE_SIZES = (2048, 4096, 8192, 8208, 12288, 16384, 16400, 20480, 24576, 32768, 32784, 40960, 49152, 65536, 65552,
                   81920, 98304, 131072)
    DEFAULT_CURVES = None   # filled below the class (measured on MI355X)

    @staticmethod
    def calibrate_curves(device: int, sizes=None, seed: int = 5, steps: int = 120, solve: str = "pinv"):
        """Measure {type: (sizes, us per step)} on `device` with ragged lists k ~ U{0..K} (the fleet's distribution); solve: the
        resolve the fleet will run (bench.py's default, the reference's: pinv)."""
        import numpy as np
        from . import configs as Cf
        from .engine import Engine
        dev = torch.device("cuda", device)
        sizes = list(sizes or MixedFleetShard.CURVE_SIZES)
        out = {}
        for key, builder, sampler, first in (("two_joint", Cf.config5_two_joint, Cf.sample_two_joint_states, 0),
                                             ("panda", Cf.config3, Cf.sample_panda_states, 1)):
            _, desc = builder(solve)
            eng = Engine(desc, device)
            sph = Cf.sample_spheres(np.random.default_rng([seed, first]))
            if key == "two_joint":
                sph[:, :2] *= 2.0
                sph[:, 2] = 0.1
            spt = torch.from_numpy(sph).to(dev)
            us = []
            for R in sizes:
                rng = np.random.default_rng([seed, 77, R])
                st = sampler(rng, R)
                off, idx = Cf.sample_ragged(rng, R)
                q, qd, goal = (torch.from_numpy(st[x]).to(dev) for x in ("q", "qd", "goal"))
                obs = eng.obstacles(spheres=spt, csr_offset=torch.from_numpy(off), csr_index=torch.from_numpy(idx))
                launch, _ = eng.bind(q, qd, goal, obstacles=obs)
                for _ in range(20):
                    launch()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(steps):
                    launch()
                b.record()
                torch.cuda.synchronize(dev)
                us.append(a.elapsed_time(b) * 1e3 / steps)
            out[key] = (sizes, us)
        return out

    @staticmethod
    def plan_by_curves(total: int, world: int, counts, curves, pair_share=0.2):
        """Cut the type-sorted fleet so that the slowest rank's ESTIMATED TIME is minimal: bisection on the step time T; for
        a given T the ranks are filled in order with as many robots as the type's time curve allows (a rank that holds both
        types runs its two kernels within T in sum).  Robots count by their list length: effective robots =
        sum(1 - pair_share + pair_share * k_r / mean k).  Returns (cuts, ranges, est_us per rank)."""
        import numpy as np
        counts = np.asarray(counts, dtype=np.float64)
        n_tj = total // 2
        types = ("two_joint", "panda")
        seg = {"two_joint": (0, n_tj), "panda": (n_tj, total)}
        eff, curve = {}, {}
        for t in types:
            lo, hi = seg[t]
            k = counts[lo:hi]
            w = (1.0 - pair_share) + pair_share * k / max(k.mean(), 1e-9) if hi > lo else k
            eff[t] = np.concatenate([[0.0], np.cumsum(w)])           # effective robots up to position i
            sz, us = (np.asarray(x, dtype=np.float64) for x in curves[t])
            us = np.maximum.accumulate(us)                            # monotone: invertible
            # (a non-empty part costs at least the smallest measured launch: a 200-robot sliver of a second type is not free)
            curve[t] = (np.concatenate([[0.0, 1.0], sz]), np.concatenate([[0.0, us[0]], us]))

        def time_of(t, n_eff):
            sz, us = curve[t]
            if n_eff <= sz[-1]:
                return float(np.interp(n_eff, sz, us))
            return float(us[-1] * n_eff / sz[-1])                    # beyond the ladder: proportional

        def cap_of(t, T):
            """largest effective robot count of type t with time <= T"""
            sz, us = curve[t]
            if T <= 0:
                return 0.0
            if T >= us[-1]:
                return float(sz[-1] * T / us[-1])
            j = int(np.searchsorted(us, T, side="right")) - 1       # us[j] <= T < us[j+1]
            if us[j + 1] == us[j]:
                return float(sz[j + 1])
            return float(sz[j] + (sz[j + 1] - sz[j]) * (T - us[j]) / (us[j + 1] - us[j]))

        def fill(T):
            pos = {"two_joint": 0, "panda": 0}       # robots of each type assigned so far (positions in the type's range)
            cuts, est = [0], []
            for r in range(world):
                budget = T
                took = {}
                for t in types:
                    n_t = seg[t][1] - seg[t][0]
                    if pos[t] >= n_t or budget <= 0:
                        continue
                    if t == "panda" and pos["two_joint"] < n_tj:
                        continue                      # contiguous cut: Pandas only after the last TwoJoint robot
                    cap = cap_of(t, budget)
                    target = eff[t][pos[t]] + cap
                    new = int(np.searchsorted(eff[t], target, side="right")) - 1
                    new = max(min(new, n_t), pos[t])
                    took[t] = (pos[t], new)
                    budget -= time_of(t, eff[t][new] - eff[t][pos[t]])
                    pos[t] = new
                cuts.append(pos["two_joint"] + pos["panda"])
                est.append(T - budget)
            return cuts, est, pos["two_joint"] + pos["panda"] >= total

        lo_T, hi_T = 0.0, sum(time_of(t, eff[t][-1]) for t in types) + 1.0
        for _ in range(40):
            mid = 0.5 * (lo_T + hi_T)
            if fill(mid)[2]:
                hi_T = mid
            else:
                lo_T = mid
        cuts, est, ok = fill(hi_T)
        if any(cuts[r + 1] == cuts[r] for r in range(world)) and world > 1 and 0 < n_tj < total:
            # a fleet too small to fill the ranks at the minimal step time (a launch costs its latency floor however few
            # robots it carries): no rank is left empty -- the ranks are dealt to the types in proportion to their times and
            # every type is split evenly (by effective robots) over its ranks
            t_all = {t: time_of(t, eff[t][-1]) for t in types}
            n_a = int(round(world * t_all["two_joint"] / max(t_all["two_joint"] + t_all["panda"], 1e-12)))
            n_a = min(max(n_a, 1), world - 1)
            cuts = [0]
            for t, n_r, base in (("two_joint", n_a, 0), ("panda", world - n_a, n_tj)):
                for i in range(1, n_r + 1):
                    target = eff[t][-1] * i / n_r
                    cuts.append(base + min(int(np.searchsorted(eff[t], target, side="left")), len(eff[t]) - 1))
            est = []
            for r in range(world):
                t = "two_joint" if r < n_a else "panda"
                lo_, hi_ = cuts[r] - seg[t][0], cuts[r + 1] - seg[t][0]
                est.append(time_of(t, eff[t][hi_] - eff[t][lo_]))
        cuts[-1] = total
        for r in range(1, world + 1):
            cuts[r] = max(cuts[r], cuts[r - 1])
        # slivers: a rank that holds both types with fewer than 256 robots of one of them hands those to the neighbour that
        # runs that type anyway (a second kernel launch for a handful of robots costs its latency floor)
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            if lo < n_tj < hi:
                if hi - n_tj < 256 and r + 1 < world:
                    cuts[r + 1] = n_tj
                elif n_tj - lo < 256 and r > 0:
                    cuts[r] = n_tj
        ranges = []
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            ranges.append({"two_joint": (min(lo, n_tj), min(hi, n_tj)),
                           "panda": (max(lo, n_tj) - n_tj, max(hi, n_tj) - n_tj)})
        return cuts, ranges, est

    @staticmethod
    def plan(total: int, world: int, counts, cost=None):
        """counts[r] = obstacles robot r sees, in TYPE-SORTED order (first total // 2 robots: TwoJoint, rest: Panda).
        cost = {type: (ns per robot, ns per pair)} (default: DEFAULT_COST).
        Returns (cuts, ranges, work): cuts = world + 1 indices into the sorted fleet; ranges[rank] = {type: (lo, hi)} in
        per-type robot numbering; work[r] = estimated ns of robot r."""
        import numpy as np
        if cost is None and MixedFleetShard.DEFAULT_CURVES is not None:
            cost = {"curves": MixedFleetShard.DEFAULT_CURVES}
        if cost is not None and "curves" in cost:
            cuts, ranges, est = MixedFleetShard.plan_by_curves(total, world, counts, cost["curves"])
            work = np.zeros(total)                     # per-robot share of its rank's estimated time (ns)
            for r in range(world):
                n = cuts[r + 1] - cuts[r]
                if n > 0:
                    work[cuts[r]:cuts[r + 1]] = est[r] * 1e3 / n
            return cuts, ranges, work
        cost = cost or MixedFleetShard.DEFAULT_COST
        counts = np.asarray(counts)
        n_tj = total // 2
        is_tj = np.arange(total) < n_tj
        (a_tj, b_tj), (a_pd, b_pd) = cost["two_joint"], cost["panda"]
        work = np.where(is_tj, a_tj, a_pd) + np.where(is_tj, 3 * b_tj, 8 * b_pd) * counts
        cuts = balanced_bounds(work, world)
        ranges = []
        for r in range(world):
            lo, hi = cuts[r], cuts[r + 1]
            ranges.append({"two_joint": (min(lo, n_tj), min(hi, n_tj)),
                           "panda": (max(lo, n_tj) - n_tj, max(hi, n_tj) - n_tj)})
        return cuts, ranges, work

    @classmethod
    def synthetic_host(cls, total: int, world: int, rank: int, seed: int = 5, cost=None):
        """The HOST side of synthetic(): this rank's shard of the synthetic config-5 fleet as numpy inputs, no device touched --
        {type: dict(n, st = {q, qd, goal}, spheres, csr_offset, csr_index, k)} for the types the rank holds, plus "work" (the
        plan's per-robot cost estimates of the shard).  The draws are exactly synthetic()'s (which calls this), so that a
        CPU-side tool (tests/golden/make_perf_envelope.py) sees the very robots a GPU run steps."""
        import numpy as np
        from . import configs as Cf
        rng = np.random.default_rng(seed)
        counts = rng.integers(0, Cf.N_SPHERES + 1, size=total)
        _, ranges, work = cls.plan(total, world, counts, cost)
        n_tj = total // 2
        host = {"work": {}}
        for key, sampler, first in (("two_joint", Cf.sample_two_joint_states, 0), ("panda", Cf.sample_panda_states, n_tj)):
            lo, hi = ranges[rank][key]
            n = hi - lo
            if n <= 0:
                continue
            trng = np.random.default_rng([seed, first, rank])
            st = sampler(trng, n)
            sph = Cf.sample_spheres(np.random.default_rng([seed, first]))   # one table per type, same on every rank
            if key == "two_joint":   # the planar arm lives at z ~ 0.1, reach 2: spread the spheres there
                sph[:, :2] *= 2.0
                sph[:, 2] = 0.1 + 0.3 * np.random.default_rng([seed, 1]).uniform(-1, 1, len(sph)).astype(np.float32)
            k = counts[first + lo:first + hi]
            order = np.argsort(trng.random((n, Cf.N_SPHERES)), axis=1).astype(np.int32)   # a permutation per robot
            take = np.arange(Cf.N_SPHERES)[None, :] < k[:, None]
            csr_index = order[take]
            csr_offset = np.concatenate([[0], np.cumsum(k)]).astype(np.int32)
            host[key] = dict(n=n, st=st, spheres=sph, csr_offset=csr_offset, csr_index=csr_index, k=k)
            host["work"][key] = float(work[first + lo:first + hi].sum())
        return host

    @classmethod
    def synthetic(cls, total: int, world: int, rank: int, device: int, seed: int = 5, solve: str = "auto", cost=None,
                  fused: bool = True, link_geometry: bool = False):
        """Synthetic config-5 fleet of `total` robots (k_r ~ U{0..32} as CSR lists into the type's
        shared sphere table); builds only this rank's shard.  `cost`: see plan().
        link_geometry: the control point of a pair is the nearest point of the LINK's capsule to the obstacle (formed inside
        the step, rmp2_obstacles.link_capsules over the ragged lists: round 4) instead of the frame origin; the two robot
        types then step as two launches (the one-grid build carries no link geometry)."""
        import numpy as np
        from . import configs as Cf
        from .engine import Engine
        host = cls.synthetic_host(total, world, rank, seed, cost)
        self = cls()
        self.parts = {}
        self.work = 0.0
        dev = torch.device("cuda", device)
        for key, builder in (("two_joint", Cf.config5_two_joint), ("panda", Cf.config3)):
            if key not in host:
                continue
            hp = host[key]
            n, st, sph, csr_offset, csr_index, k = (hp[x] for x in ("n", "st", "spheres", "csr_offset", "csr_index", "k"))
            _, desc = builder(solve)
            eng = Engine(desc, device)
            q, qd, goal = (torch.from_numpy(st[x]).to(dev) for x in ("q", "qd", "goal"))
            out = torch.empty_like(q)
            lc = None
            if link_geometry:
                from . import urdf as U
                if key == "two_joint":
                    lc = U.link_capsules(U.TWO_JOINT_URDF, U.two_joint_table(), Cf.TWO_JOINT_CONTROL_POINT_FRAMES)
                else:
                    lc = U.link_capsules(U.PANDA_URDF, U.panda_table(), Cf.CONTROL_POINT_FRAMES)
                lc_host = lc
                lc = torch.from_numpy(lc).to(dev)
            obs = eng.obstacles(spheres=torch.from_numpy(sph).to(dev), csr_offset=torch.from_numpy(csr_offset),
                                csr_index=torch.from_numpy(csr_index), link_capsules=lc)
            # the two types' kernels are independent: the TwoJoint part runs on a side stream, beside the Pandas'
            side = torch.cuda.Stream(dev) if key == "two_joint" and "panda" in host else None
            launch, _ = eng.bind(q, qd, goal, obstacles=obs, out=out, stream=side.cuda_stream if side is not None else None)
            if side is not None:
                self._side, self._side_launch = side, launch
                self._fork, self._join = _Fence(dev), _Fence(dev)
            pairs = float(cls.CONTROL_POINTS[key] * k.sum())
            self.parts[key] = dict(engine=eng, launch=launch, out=out, n=n, keep=(q, qd, goal, obs), desc=desc,
                                   host=dict(spheres=sph, csr_offset=csr_offset, csr_index=csr_index,
                                             link_capsules=lc_host if link_geometry else None),
                                   bytes=float(n * (cls.BYTES[key] + 4) + 4 * k.sum()),
                                   flops=float(n * cls.BASE_FLOPS[key] + 240.0 * pairs))
            self.work += host["work"][key]
        self.n_two_joint = self.parts.get("two_joint", {}).get("n", 0)
        self.n_panda = self.parts.get("panda", {}).get("n", 0)
        # both types in the shard: their two steps as ONE call -- one fused grid where the library has the instantiation
        # (rmp2_step_pair: both fleets beyond 8 192 robots), else two launches
        self._pair_launch = None
        if fused and len(self.parts) == 2:
            from .engine import bind_pair
            a, b = self.parts["two_joint"], self.parts["panda"]
            self._pair_launch = bind_pair(a["engine"], a["keep"][0], a["keep"][1], a["keep"][2], a["keep"][3], a["out"],
                                          b["engine"], b["keep"][0], b["keep"][1], b["keep"][2], b["keep"][3], b["out"])
            self._pair_launch()
            self._fused = "pair" in a["engine"].last_kernel()
        self._dom = "panda" if "panda" in self.parts else "two_joint"
        d = self.parts[self._dom]
        self.dominant_bytes, self.dominant_flops, self.dominant_robots = d["bytes"], d["flops"], d["n"]
        self._launches = [p["launch"] for p in self.parts.values()]
        self._device = dev
        return self

    _side = None
    _graph = None

    def capture(self) -> bool:
        """Record the shard's step (fork, two launches, join) ONCE as a HIP graph; step() then replays it with a single
        host call.  MEASURED SLOWER on ROCm 7.2 / MI355X than the eager sequence (config 5 at world 1: 62.1 us per step against
        40.3 eager; host issue time 19.7 against 28.1 us -- the graph's cross-stream edges are executed as full barriers), so
        nothing calls this by default; kept for A/B runs (bench.py --graph).  Returns False (and keeps the eager step) if the
        capture fails."""
        if self._graph is not None:
            return True
        try:
            torch.cuda.synchronize(self._device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                # (the pre-bound launches target the streams they were bound on; the recording issues the same steps on the
                # capture stream -- and on the side stream forked from it -- through Engine.step)
                cur = torch.cuda.current_stream(self._device)

                def issue(key, stream):
                    p = self.parts[key]
                    q, qd, goal, obs = p["keep"]
                    p["engine"].step(q, qd, goal, obstacles=obs, out=p["out"], stream=stream.cuda_stream)

                if self._side is None:
                    for key in self.parts:
                        issue(key, cur)
                else:
                    other = [k for k in self.parts if k != self._dom][0]
                    self._fork.record(cur)
                    self._fork.wait(self._side)
                    issue(other, self._side)
                    self._join.record(self._side)
                    issue(self._dom, cur)
                    self._join.wait(cur)
            torch.cuda.synchronize(self._device)
            self._graph = g
            return True
        except Exception:
            self._graph = None
            return False

    def step(self):
        """One control step of the shard (graph replay after capture(), else the eager sequence)."""
        if self._graph is not None:
            self._graph.replay()
        else:
            self._step_eager()

    _pair_launch = None
    _fused = False

    def _step_eager(self):
        """With both types present: ONE call (rmp2_step_pair) -- a fused grid when both fleets are beyond 8 192 robots --;
        where that is not the fused grid the two kernels overlap on two streams: the side stream forks off the current
        stream (it sees everything enqueued so far, e.g. the simulator's state update) and joins it again."""
        if self._pair_launch is not None and self._fused:
            self._pair_launch()
            return
        if self._side is None:
            for launch in self._launches:
                launch()
            return
        cur = torch.cuda.current_stream(self._device)
        self._fork.record(cur)
        self._fork.wait(self._side)
        self._side_launch()
        self._join.record(self._side)
        self.parts[self._dom]["launch"]()
        self._join.wait(cur)

    def step_dominant(self):
        self.parts[self._dom]["launch"]()

    def dominant_kernel(self) -> str:
        return self.parts[self._dom]["engine"].last_kernel()


# measured on one MI355X (profiles/r03_cost_calibration.json: tools/calibrate_costs.py, kernels of round 3)
# (round 5: measured with solve = pinv -- bench.py's default, the reference's resolve -- on the round's kernels, tools/calibrate_costs.py,
#  profiles/r05_cost_calibration.json; round 3's curves, solve = auto: profiles/r03_cost_calibration.json)
MixedFleetShard.DEFAULT_CURVES = {
    "two_joint": (list(MixedFleetShard.CURVE_SIZES), [12.06, 12.1, 12.7, 12.71, 12.94, 13.16, 11.95, 12.38, 12.45, 12.61, 13.39, 14.17, 14.37, 15.8, 17.2, 19.08, 25.53, 29.84]),
    "panda": (list(MixedFleetShard.CURVE_SIZES), [22.43, 22.97, 26.89, 35.22, 35.47, 35.66, 32.73, 34.65, 34.75, 35.39, 38.81, 40.99, 41.75, 48.14, 63.71, 69.31, 76.88, 89.16]),
}
