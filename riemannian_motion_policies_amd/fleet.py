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


class ObstacleExchange:
    """All-gather of the per-rank slices of the shared sphere table [K, 4] (RCCL, side stream).

    Two table buffers: the gather for step k + 1 can be in flight while the kernel of step k reads the other buffer
    (`start` right after launching the step; `finish` at the top of the next one).  Ordering is by events only:
    the gather waits for the producer of `local` and for the last kernel that read the buffer it overwrites; the
    consuming step waits for the gather."""

    def __init__(self, spheres_per_rank: int, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device)
        cuda = self.device.type == "cuda"
        self.tables = [torch.zeros((self.world * spheres_per_rank, 4), dtype=torch.float32, device=self.device)
                       for _ in range(2)]
        self.side = torch.cuda.Stream(self.device) if cuda else None
        self.ready = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self.reader_done = [None, None]   # event after the last kernel that read buffer b
        self._reader_events = [torch.cuda.Event() for _ in range(2)] if cuda else None
        self._next, self._pending, self._last = 0, [], None

    @property
    def table(self) -> torch.Tensor:
        """The most recently finished table."""
        return self.tables[self._last if self._last is not None else 0]

    def start(self, local: torch.Tensor, produced=None) -> None:
        """Issue the all-gather of `local` [K/world, 4] into the free buffer.  `produced`: event after which `local`
        is valid (default: everything issued so far on the current stream)."""
        b = self._next
        self._next ^= 1
        self._pending.append(b)
        if self.world == 1:
            self.tables[b].copy_(local)
            return
        if self.side is not None:
            if produced is not None:
                self.side.wait_event(produced)
            else:
                self.side.wait_stream(torch.cuda.current_stream(self.device))
            if self.reader_done[b] is not None:
                self.side.wait_event(self.reader_done[b])
            with torch.cuda.stream(self.side):
                dist.all_gather_into_tensor(self.tables[b], local.contiguous(), group=self.group)
                self.ready[b].record(self.side)
        else:
            dist.all_gather_into_tensor(self.tables[b], local.contiguous(), group=self.group)

    def finish(self) -> torch.Tensor:
        """Make the current stream wait for the oldest outstanding gather; returns its table."""
        b = self._pending.pop(0)
        self._last = b
        if self.world > 1 and self.side is not None:
            torch.cuda.current_stream(self.device).wait_event(self.ready[b])
        return self.tables[b]

    def consumed(self) -> None:
        """Call after launching the kernel that reads the table returned by the last finish()."""
        if self.world > 1 and self.side is not None and self._last is not None:
            ev = self._reader_events[self._last]
            ev.record(torch.cuda.current_stream(self.device))
            self.reader_done[self._last] = ev


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
