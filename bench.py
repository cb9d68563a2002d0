#!/usr/bin/env python3
"""bench.py -- RMP2 control steps/s (batched robots) on N MI355X of one node.

A "step" is one pass of the hot path (rmp2_step: FK -> J, Jdot qd -> leaves -> pull-back ->
sum -> resolve) over this rank's robot batch, inputs and outputs resident in HBM.  Default
workload = BASELINE.json configs[1]: Franka Panda, TargetAttractor + JointLimitAvoidance +
JointDamping, 4096 robots per GPU, fp32 I/O (weak scaling: every rank steps its own 4096).
`--workload config3` runs the cluttered set (8 control points x 32 spheres, 65536 robots per
GPU) whose sphere table is produced distributed and all-gathered over RCCL every step.

    python bench.py [--gpus N --steps K --warmup W --workload config2|config3 --robots R]

Rank 0 prints ONE JSON line (contract in the task description): whole-job steps/s, the HBM
and VALU roofline fractions of the control-step kernel computed from the ALGORITHMIC bytes /
flops of BASELINE.md section 3, and a CPU baseline (the oracle, OpenMP over robots, timed on
this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd import descriptor as D  # noqa: E402

HBM_PEAK = 8.0e12        # B/s   MI355X_MICROARCH.md "HBM3E peak BW"
VALU_PEAK = 157.3e12     # flop/s fp32 vector (non-MFMA)
# BASELINE.md section 3 / SURVEY 8(d): algorithmic bytes and flops per robot-step
WORKLOADS = {
    "config2": dict(builder=Cf.config2, robots=4096, bytes=120, flops=3.0e3, spheres=0,
                    name="Franka Panda, target + joint-limit + damping, 4096 robots/GPU (BASELINE configs[1])"),
    "config3": dict(builder=Cf.config3, robots=65536, bytes=120, flops=66.0e3, spheres=Cf.N_SPHERES,
                    name="Franka Panda cluttered: 8 control points x 32 shared spheres, 65536 robots/GPU (BASELINE configs[2])"),
}


def host_cores() -> int:
    """Cores this process may really use: min(affinity, cgroup CPU quota)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(desc, s, spheres, budget_s=10.0):
    """Oracle (plain-C restatement, `port`) on the host cores, OpenMP over robots, bounded sample."""
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # must be set before libgomp is loaded
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    R = min(len(s["q"]), 4096)
    q, qd, goal = s["q"][:R], s["qd"][:R], s["goal"][:R]
    kw = dict(spheres=spheres) if spheres is not None else {}
    O.step(desc, q[:64], qd[:64], goal[:64], **kw)  # warm-up / page-in
    iters, t0 = 0, time.perf_counter()
    while True:
        O.step(desc, q, qd, goal, **kw)
        iters += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or iters >= 5000:
            break
    return {"value": R * iters / dt, "unit": "robot control steps/s", "cores": cores, "kind": "port",
            "sample": f"{iters} steps of {R} robots, oracle/rmp2_oracle.c (gcc -O3 -march=native, OpenMP {cores} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # ~17 ms of GPU time at the headline size: above timer noise
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS))
    ap.add_argument("--robots", type=int, default=0, help="robots per GPU (default: the workload's)")
    ap.add_argument("--solve", default="auto", choices=["auto", "pinv"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the RMP2 engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun (RANK/WORLD_SIZE set) the process group is created even for one rank, so that the
    # N > 1 code path (RCCL init, barriers, MAX-reduce of the time, obstacle all-gather) is the one
    # exercised on a single-GPU box as well
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        # keep stdout to the ONE JSON line: RCCL prints its version banner (and warnings) on fd 1 while the
        # communicator is created, so fd 1 points at stderr until the first collective has completed
        sys.stdout.flush()
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd1, 1)
            os.close(saved_fd1)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build_hip()
    if use_dist:
        dist.barrier()
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import ObstacleExchange

    wl = WORKLOADS[args.workload]
    R = args.robots or wl["robots"]
    _, desc = wl["builder"](args.solve)
    eng = Engine(desc, local_rank)

    # synthetic inputs, SURVEY 8(d): seed 1 -> performance inputs (rank-offset so shards differ)
    rng = np.random.default_rng(1 + 1000 * rank)
    s = Cf.sample_panda_states(rng, R)
    q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
    out = torch.empty_like(q)
    spheres_np = None
    exch = None
    obstacles = None
    if wl["spheres"]:
        K = wl["spheres"]
        spheres_np = Cf.sample_spheres(np.random.default_rng(7), K)   # same table on every rank
        if use_dist:
            if K % world:
                raise SystemExit("sphere count must divide by the world size")
            exch = ObstacleExchange(K // world, dev)
            local = torch.from_numpy(spheres_np[rank * (K // world):(rank + 1) * (K // world)]).to(dev)
        else:
            obstacles = eng.obstacles(spheres=torch.from_numpy(spheres_np).to(dev))

    if exch is None:
        launch, _ = eng.bind(q, qd, goal, obstacles=obstacles, out=out)   # bare C-ABI call on fixed buffers

        def one_step():
            launch()
    else:
        # obstacle all-gather on a side stream, pipelined one step ahead: the table of step k + 1 is gathered
        # (into the second buffer) while the kernel of step k runs; every step consumes a freshly gathered table
        local_ready = torch.cuda.Event()
        local_ready.record(torch.cuda.current_stream(dev))
        exch.start(local, produced=local_ready)

        def one_step():
            table = exch.finish()
            eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=table), out=out)
            exch.consumed()
            exch.start(local, produced=local_ready)

    for _ in range(args.warmup):
        one_step()
    # HIP events (on the stream the kernel is launched on) bracket GROUPS of `grp` consecutive launches of the
    # timed region: an event pair costs ~4 us of GPU time by itself, as much as a third of one launch of the
    # latency-bound kernel, so it is amortised over the group and its empty-pair reading is calibrated out.
    # kernel_ms = elapsed / grp is the steady-state time per launch INCLUDING the idle gap between two dependent
    # launches (~0.7 us) and 1/grp of an event pair; rocprofv3's kernel-trace average (profiles/) is the kernel alone.
    grp = 8 if args.steps >= 16 else 1
    n_groups = min(8, args.steps // grp)
    starts = {int(round(k * (args.steps - grp) / max(n_groups - 1, 1))) for k in range(n_groups)}
    ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in sorted(starts)}
    ends = {i + grp - 1: i for i in ev}
    stream = torch.cuda.current_stream(dev)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in ev:
            ev[i][0].record(stream)
        one_step()
        if i in ends:
            ev[ends[i]][1].record(stream)
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    raw_ms = float(np.median([a.elapsed_time(b) for a, b in ev.values()]))  # one group of `grp` launches
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(32)]
    for a, b in empty:
        a.record(stream)
        b.record(stream)
    torch.cuda.synchronize(dev)
    floor_ms = float(np.median([a.elapsed_time(b) for a, b in empty]))
    # conservative: the event pair's own cost (floor_ms, reported) is NOT subtracted when it is amortised over a
    # group -- the per-launch figure then sits between the steady-state step time and rocprofv3's kernel average
    kern_ms = raw_ms / grp if grp > 1 else max(raw_ms - floor_ms, 1e-6)

    total_steps = R * world * args.steps
    value = total_steps / dt
    if rank == 0:
        per_launch_bytes = wl["bytes"] * R
        per_launch_flops = wl["flops"] * R
        ach_bw = per_launch_bytes / (kern_ms * 1e-3)
        ach_fl = per_launch_flops / (kern_ms * 1e-3)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"traffic_{args.workload}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("robots") == R:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "RMP2 control steps/sec (batched robots)",
            "value": value,
            "unit": "robot control steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (leaves, Jacobians, pull-back) + f64 (sum over leaves, resolve)",
            "data": "synthetic",
            "config": {"workload": wl["name"], "robots_per_gpu": R, "solve": args.solve,
                       "parallelism": f"robot-batch split x{world}" + (", RCCL all-gather of the sphere table per step" if exch else "")},
            "roofline": {"bound": "hbm", "achieved": ach_bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": ach_bw / HBM_PEAK, "traffic": traffic,
                         "kernel": ("rmp2_step_hex_kernel" if R <= 20480 else ("rmp2_step_quad_kernel" if wl["spheres"] else "rmp2_step_kernel")) + " (chosen by fleet size, rmp2_hip.hip dispatch_solve)", "kernel_ms": kern_ms,
                         "event_group_launches": grp, "event_group_ms_raw": raw_ms, "event_pair_ms_empty": floor_ms,
                         "algorithmic_bytes_per_robot_step": wl["bytes"],
                         "valu": {"achieved": ach_fl / 1e12, "peak": VALU_PEAK / 1e12, "unit": "TFLOP/s",
                                  "frac": ach_fl / VALU_PEAK, "algorithmic_flops_per_robot_step": wl["flops"]},
                         "binding": "fp32 VALU / launch latency (see DESIGN.md: 120 B per robot-step cannot load HBM)"},
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(desc, s, spheres_np)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
