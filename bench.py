#!/usr/bin/env python3
"""bench.py -- RMP2 control steps/s (batched robots) on N MI355X of one node.

A "step" is one pass of the hot path (rmp2_step: FK -> J, Jdot qd -> leaves -> pull-back ->
sum -> resolve) over this rank's robot batch, inputs and outputs resident in HBM.

    python bench.py [--gpus N --steps K --warmup W --workload auto|config2|config3|config4|config5 --robots R]

Workloads (BASELINE.json configs[1..4]; weak scaling: the per-GPU fleet is fixed):
  config3  (default at N = 1) Franka Panda cluttered set, 8 control points x 32 shared spheres, 65536 robots per
           GPU -- the largest single-GPU configuration.  The latency figure of configs[1] (config 2, 4096 robots)
           rides along as the nested "secondary" object of the same JSON line (same process, second engine).
  config4  (default at N > 1) config 3 per GPU, the sphere table produced distributed (K / N spheres per rank) and
           all-gathered over RCCL every step on a side stream, one step ahead of the kernel that consumes it.
  config5  mixed fleet: 50/50 TwoJoint + Panda, ragged per-robot obstacle lists (CSR), 32768 robots per GPU;
           the type-sorted fleet is cut across ranks so that the estimated work (sum of pairs) is balanced.
  config2  Franka Panda, target + joint-limit + damping, 4096 robots per GPU.

`--gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (fresh child processes, before
anything in this process touches a GPU) and fails loudly when the node has fewer devices; under
`torch.distributed.run` the ranks are taken from the environment.

`--solve pinv` (the default: the reference's only resolve, rmp.py:148-152, as one certifying launch) gives the line's `value`;
the same workload under `--solve auto` (elimination without the certificate) rides along as the nested `solve_auto` object.

Rank 0 prints ONE JSON line (contract in the task description): whole-job steps/s, the roofline fractions of the
control-step kernel from the ALGORITHMIC bytes / flops of BASELINE.md section 3 (and, beside them, from the flops this run
EXECUTED: `roofline.executed_frac`), and (N = 1) a CPU baseline timed on this box's host cores on a bounded sample.  N > 1
lines carry `world1_same_workload_ms`: the same workload on every rank's own shard with a one-rank communicator, MAX over ranks
-- the like-for-like base of a weak-scaling efficiency.  No child process is started once HIP has been touched (host facts come
from /proc, the libraries are built in `preflight()` before the first device call).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # B/s   MI355X_MICROARCH.md "HBM3E peak BW"
VALU_PEAK = 157.3e12     # flop/s fp32 vector (non-MFMA)
PAIR_FLOPS = 240.0       # per distance pair
# BASELINE.md section 3 / SURVEY 8(d): algorithmic bytes and flops per robot-step
WORKLOADS = {
    "config2": dict(robots=4096, bytes=120, flops=3.0e3,
                    name="Franka Panda, target + joint-limit + damping, 4096 robots/GPU (BASELINE configs[1])"),
    "config3": dict(robots=65536, bytes=120, flops=66.0e3,
                    name="Franka Panda cluttered: 8 control points x 32 shared spheres, 65536 robots/GPU (BASELINE configs[2])"),
    "config3c": dict(robots=65536, bytes=120, flops=66.0e3,
                     name="Franka Panda cluttered with CAPSULE obstacles (the reference's cylinders, simulation.py:495-500, as "
                          "capsules): 8 control points x 32 shared capsules, 65536 robots/GPU"),
    "config3l": dict(robots=65536, bytes=120, flops=66.0e3,
                     name="Franka Panda cluttered with LINK geometry: the control point of each of the 8 x 32 pairs is the nearest "
                          "point of the link's capsule to the sphere (PyBullet's closest points on the link shape, "
                          "simulation.py:462-484), formed inside the step (rmp2_obstacles.link_capsules), 65536 robots/GPU"),
    "config3b": dict(robots=65536, bytes=6264, flops=66.0e3, bound="hbm",
                     name="Franka Panda cluttered, interface B: 256 explicit closest-point pairs per robot (p_link, p_obs "
                          "[R, 256, 3], the reference's Datamanager layout), 65536 robots/GPU (BASELINE.md section 3 row 3-B)"),
    "config4": dict(robots=65536, bytes=120, flops=66.0e3,
                    name="Franka Panda cluttered, 65536 robots/GPU, sphere table sharded over the ranks and "
                         "all-gathered over RCCL every step (BASELINE configs[3])"),
    "config5": dict(robots=32768, bytes=None, flops=None,
                    name="mixed fleet 50/50 TwoJoint + Franka Panda, ragged per-robot obstacle lists k_r ~ U{0..32}, "
                         "32768 robots/GPU, work-balanced cut (BASELINE configs[4])"),
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


def lscpu_summary() -> str:
    """What `lscpu` prints about the host, read from /proc/cpuinfo -- NO child process: this runs in a process that may hold a
    HIP context, and such a process must not fork + exec on this pool (count_gpus_without_hip).  main() also calls it once
    before anything can touch a GPU and keeps the string (HOST_SUMMARY)."""
    try:
        model, sockets, cores_per_socket, siblings, cpus = "", set(), 0, 0, 0
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                cpus += 1
            elif k == "model name" and not model:
                model = v
            elif k == "physical id":
                sockets.add(v)
            elif k == "cpu cores":
                cores_per_socket = int(v)
            elif k == "siblings":
                siblings = int(v)
        tpc = max(1, siblings // cores_per_socket) if cores_per_socket else 1
        return (f"Model name={model}; Socket(s)={max(len(sockets), 1)}; Core(s) per socket={cores_per_socket}; "
                f"Thread(s) per core={tpc}; CPU(s)={cpus}")
    except Exception:
        return "/proc/cpuinfo unavailable"


HOST_SUMMARY = None   # filled by main() before any HIP initialisation


def cpu_baseline(workload: str, desc, table, s, spheres):
    """The oracle on this box's host cores, on bounded samples of the same workload:
       value / cores   plain-C restatement (`port`), OpenMP over robots, every core this process may use
       threads_1       the same with one thread
       reference_style_proxy   the torch-autograd restatement run the way the reference runs (one robot per
                               call, FK re-differentiated per RMP) -- proxy, not TensorFlow"""
    import ctypes
    import numpy as np
    cores = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)   # before libgomp is loaded
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    R = min(len(s["q"]), 4096)
    q, qd, goal = s["q"][:R], s["qd"][:R], s["goal"][:R]
    kw = dict(spheres=spheres) if spheres is not None else {}
    O.step(desc, q[:64], qd[:64], goal[:64], **kw)  # warm-up / page-in (loads libgomp)
    gomp = ctypes.CDLL("libgomp.so.1")

    def timed(threads, robots, budget_s):
        gomp.omp_set_num_threads(int(threads))
        iters, t0 = 0, time.perf_counter()
        while True:
            O.step(desc, q[:robots], qd[:robots], goal[:robots], **kw)
            iters += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or iters >= 5000:
                return robots * iters / dt, iters, dt

    v1, it1, dt1 = timed(1, min(R, 512), 4.0)
    vN, itN, dtN = timed(cores, R, 8.0)
    out = {"value": vN, "unit": "robot control steps/s", "cores": cores, "kind": "port",
           "sample": f"{itN} steps of {R} robots, oracle/rmp2_oracle.c (gcc -O3 -march=native, OpenMP {cores} threads), {dtN:.1f} s",
           "threads_1": {"value": v1, "cores": 1,
                         "sample": f"{it1} steps of {min(R, 512)} robots, same C restatement, 1 thread, {dt1:.1f} s"},
           "host": (HOST_SUMMARY or lscpu_summary()) + f"; usable by this process: {cores}"}
    try:
        import torch_autodiff_oracle as TA
        from riemannian_motion_policies_amd import configs as Cf, descriptor as D
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "kinematic_tables.json")))
        fk = TA.UrdfForwardKinematicTorch(gold["panda"])
        leaves = TA.leaves_from_desc(desc, table.frame_names)
        pairs_of = None
        if spheres is not None:
            T = O.forward_kinematics(desc, q[:16], precision="f64")
            frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
            origins = T[:, frames][:, :, :3, 3].astype(np.float32)
            pl, po = Cf.pairs_from_spheres(origins, spheres)
            dl, K = D.distance_leaf_indices(desc), spheres.shape[0]

            def pairs_of(r):
                return {li: (pl[r, k * K:(k + 1) * K], po[r, k * K:(k + 1) * K]) for k, li in enumerate(dl)}
        n, t0 = 0, time.perf_counter()
        while True:
            r = n % 16
            TA.evaluate_one(fk, leaves, q[r], qd[r], goal[r], pairs_of(r) if pairs_of else None)
            n += 1
            dt = time.perf_counter() - t0
            if (dt >= 10.0 and n >= 10) or n >= 100:
                break
        out["reference_style_proxy"] = {
            "value": n / dt, "cores": 1, "kind": "proxy, not TensorFlow",
            "sample": f"{n} control steps at R = 1 (one robot per call, FK re-differentiated per RMP, nested torch.autograd "
                      f"as the reference nests GradientTapes), oracle/torch_autodiff_oracle.py, 1 thread, {dt:.1f} s"}
    except Exception as e:  # the proxy is a reported extra, never a reason to lose the bench line
        out["reference_style_proxy"] = {"value": None, "error": repr(e)}
    return out


# ---------------------------------------------------------------------------------------------------------
def _visible_filter(n: int) -> int:
    """Apply HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (comma lists; an empty value hides all)."""
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(ids))
    return n


def count_gpus_without_hip() -> int:
    """GPUs of this node WITHOUT initialising HIP in this process: the KFD topology in sysfs (nodes with simd_count > 0 are
    GPUs; CPUs have 0), filtered by the *_VISIBLE_DEVICES variables.  torch.cuda.device_count() is NOT used here: it only
    stays clear of the runtime while amdsmi initialises; otherwise it falls back to hipGetDeviceCount, and a process that
    holds a HIP context must not fork + exec children on this pool.  If sysfs is unreadable a throw-away child counts."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        for path in nodes:
            try:
                props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except Exception:
                pass
        return _visible_filter(n)
    try:   # a fresh child may initialise whatever it likes; this process stays clean
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=300).stdout.strip().splitlines()
        return int(out[-1]) if out else 0
    except Exception:
        return 0


def spawn_ranks(args) -> int:
    """--gpus N without a launcher: start N fresh ranks.  This process never initialises HIP (see count_gpus_without_hip):
    it only counts devices through sysfs, builds the library with hipcc and waits for its children."""
    have = count_gpus_without_hip()
    if have < (1 if args.rehearse_one_gpu else args.gpus):
        print(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} HIP device(s); "
              f"not silently running on fewer", file=sys.stderr)
        return 2
    import __graft_entry__ as ge
    ge.build_hip()   # once, here (hipcc, no GPU), instead of N racing builds
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if args.rehearse_one_gpu else str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # a session (= process group) of its own per rank: the supervisor can end exactly what it started
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True))
    return supervise(procs, args.rank_timeout)


def supervise(procs, timeout_s: float, poll_s: float = 0.2, grace_s: float = 5.0) -> int:
    """Wait for the rank processes.  The job is all-or-nothing: when one rank exits non-zero, or any rank is still running
    `timeout_s` seconds after the start (a hung RCCL initialisation, a collective a dead peer never joins), the surviving
    ranks' process groups are ended (SIGTERM, SIGKILL after `grace_s`) and the supervisor returns non-zero -- the failing
    rank's code, or 124 for the time-out -- instead of waiting for the driver's own limit with no line printed."""
    import signal
    t0 = time.monotonic()
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = (abs(bad[0][1]) or 1), f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() - t0 > timeout_s:
            hung = [i for i, c in enumerate(codes) if c is None]
            rc, why = 124, f"rank(s) {hung} still running after {timeout_s:.0f} s"
            break
        time.sleep(poll_s)
    print(f"bench.py: {why}; ending the other ranks", file=sys.stderr)
    for sig, wait in ((signal.SIGTERM, grace_s), (signal.SIGKILL, grace_s)):
        alive = [p for p in procs if p.poll() is None]
        if not alive:
            break
        for p in alive:
            try:
                os.killpg(p.pid, sig)   # p.pid is the id of the group created by start_new_session -- nothing else is in it
            except (ProcessLookupError, PermissionError):
                try:
                    p.send_signal(sig)
                except ProcessLookupError:
                    pass
        t1 = time.monotonic()
        while time.monotonic() - t1 < wait and any(p.poll() is None for p in procs):
            time.sleep(0.05)
    return rc


class Timed:
    """Warm-up, then EXACTLY `steps` steps between two fences (barrier + synchronize), MAX over ranks; HIP events on the
    launch stream bracket groups of consecutive launches inside the timed region (an event pair costs about as much
    GPU time as a third of a latency-bound launch, so it is amortised over a group and never subtracted)."""

    def __init__(self, dev, use_dist):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.dev, self.use_dist = torch, dist, dev, use_dist

    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    SETTLE_S = 0.05   # untimed set-up launches before the W warm-up steps: code objects resident, clocks ramped

    def run(self, one_step, steps, warmup):
        import numpy as np
        torch = self.torch
        t_settle, self.settle_launches = time.perf_counter(), 0
        while time.perf_counter() - t_settle < self.SETTLE_S:
            one_step()
            self.settle_launches += 1
            if self.settle_launches % 64 == 0:
                torch.cuda.synchronize(self.dev)
        for _ in range(warmup):
            one_step()
        grp = 8 if steps >= 16 else 1
        # (K = 20, the driver's run: ONE group of eight, the region's last eight steps -- an event pair costs GPU time inside the region)
        n_groups = min(8, steps) if grp == 1 else min(8, max(1, steps // (4 * grp)))
        starts = {int(round(k * (steps - grp) / max(n_groups - 1, 1))) for k in range(n_groups)} if n_groups > 1 else {steps - grp}
        ev = {i: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for i in sorted(starts)}
        ends = {i + grp - 1: i for i in ev}
        stream = torch.cuda.current_stream(self.dev)
        self.fence()
        t0 = time.perf_counter()
        for i in range(steps):
            if i in ev:
                ev[i][0].record(stream)
            one_step()
            if i in ends:
                ev[ends[i]][1].record(stream)
        t_host = time.perf_counter() - t0   # the host's share: time to ISSUE the K steps (the fence below waits for the GPU)
        self.fence()
        dt = time.perf_counter() - t0
        if self.use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        raw_ms = float(np.median([a.elapsed_time(b) for a, b in ev.values()]))
        empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(32)]
        for a, b in empty:
            a.record(stream)
            b.record(stream)
        torch.cuda.synchronize(self.dev)
        floor_ms = float(np.median([a.elapsed_time(b) for a, b in empty]))
        kern_ms = raw_ms / grp if grp > 1 else max(raw_ms - floor_ms, 1e-6)
        return dict(dt=dt, kernel_ms=kern_ms, grp=grp, raw_ms=raw_ms, floor_ms=floor_ms, settle=self.settle_launches,
                    host_issue_ms_per_step=t_host / steps * 1e3)


def roofline_obj(kernel, kern, per_launch_bytes, per_launch_flops, bytes_rs, flops_rs, workload, R, bound="valu", exe=None):
    """Top level = the BINDING roof of the dominant kernel (`bound`); the other roof is nested.  `achieved` is ALGORITHMIC
    work per second of kernel time: for the flop count that is the UN-CULLED count of SURVEY 8(d)
    (all 256 pairs of a robot), which the culling kernel does not execute -- `executed` says what the silicon did."""
    ach_bw = per_launch_bytes / (kern["kernel_ms"] * 1e-3)
    ach_fl = per_launch_flops / (kern["kernel_ms"] * 1e-3)
    hbm = {"achieved": ach_bw / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": ach_bw / HBM_PEAK,
           "algorithmic_bytes_per_robot_step": bytes_rs}
    valu = {"achieved": ach_fl / 1e12, "peak": VALU_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach_fl / VALU_PEAK,
            "algorithmic_flops_per_robot_step": flops_rs}
    traffic, traffic_source = traffic_of(workload, R)
    top = dict(valu if bound == "valu" else hbm)
    out = {"bound": bound, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"], "frac": top["frac"],
           "traffic": traffic, "traffic_source": traffic_source,
           "kernel": kernel, "kernel_ms": kern["kernel_ms"],
           "event_group_launches": kern["grp"], "event_group_ms_raw": kern["raw_ms"], "event_pair_ms_empty": kern["floor_ms"],
           "setup_launches_before_warmup": kern.get("settle", 0),
           "algorithmic_bytes_per_robot_step": bytes_rs, "algorithmic_flops_per_robot_step": flops_rs,
           "flops_note": "un-culled algorithmic count (SURVEY 8(d): every (control point, obstacle) pair at 240 flops); the "
                         "kernel culls out-of-range pairs, whose metric the reference computes as exactly 0 -- see `executed`",
           "hbm" if bound == "valu" else "valu": hbm if bound == "valu" else valu,
           "binding": ("fp32 VALU issue: 120 B per robot-step cannot load HBM (DESIGN.md section 5)" if bound == "valu" else
                       "HBM: the explicit closest-point pairs are 24 B each, read once per step")}
    if exe is not None:
        # beside the algorithmic fraction: the work the culling kernel cannot avoid on this fleet, against the same peak
        out["executed_flops_per_robot_step"] = exe["flops_per_robot_step"]
        out["executed_frac"] = exe["flops_per_robot_step"] * R / (kern["kernel_ms"] * 1e-3) / VALU_PEAK
        out["executed_flops"] = exe
    ex = executed_of(workload, R)
    if ex is not None:
        out["executed"] = ex
    return out


CULL_TEST_FLOPS = 7.0     # per (control point, primitive) range test: 3 FMAs + a compare (rmp2_quad.h pair_loop_culled, pass 1)


def executed_flops(eng, desc, q, spheres_np, workload):
    """What the culling kernel has to EXECUTE per robot-step on THIS fleet, counted from the inputs (outside the timed region, on
    the device through the library's own forward kinematics -- nothing stored, nothing to go stale): the non-pair part of SURVEY
    8(d)'s count (everything but 256 x 240), 240 flops for every pair that is IN RANGE of its leaf's metric
    (x - margin <= metric_modulation_radius: beyond it rmp2.py:191-195 makes the metric exactly 0 and the kernel skips the
    pair) and a range test for every pair.  Sphere tables only (config 3 / 4): returns None elsewhere."""
    import numpy as np
    import torch
    from riemannian_motion_policies_amd import descriptor as D
    if spheres_np is None or spheres_np.shape[1] != 4 or workload not in ("config3", "config4"):
        return None
    dl = D.distance_leaf_indices(desc)
    frames = [desc.leaves[i].frame for i in dl]
    margin, radius = float(desc.leaves[dl[0]].params[0]), float(desc.leaves[dl[0]].params[7])
    sph = torch.from_numpy(spheres_np).to(q.device)
    n_in = 0
    for lo in range(0, q.shape[0], 8192):
        T = eng.forward_kinematics(q[lo:lo + 8192])
        p = T[:, frames, :3, 3]                                           # [r, C, 3]
        x = (p[:, :, None, :] - sph[None, None, :, :3]).norm(dim=-1) - sph[None, None, :, 3] - margin
        n_in += int((x <= radius).sum().item())
    pairs = len(frames) * spheres_np.shape[0]
    in_range = n_in / q.shape[0]
    non_pair = WORKLOADS["config3"]["flops"] - pairs * PAIR_FLOPS
    return {"pairs_per_robot": pairs, "in_range_pairs_per_robot_step": in_range, "in_range_pair_fraction": in_range / pairs,
            "non_pair_flops": non_pair, "cull_test_flops_per_pair": CULL_TEST_FLOPS,
            "flops_per_robot_step": non_pair + in_range * PAIR_FLOPS + pairs * CULL_TEST_FLOPS,
            "how": "counted from this run's inputs (all robots of the fleet, library FK + torch on the device, outside the timed region)"}


def _stored(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _current_kernel_hash():
    try:
        import __graft_entry__ as ge
        return ge.kernel_src_hash()
    except Exception:
        return None


def traffic_of(workload, R):
    """HBM bytes per launch from the PMC passes (tools/pmc_traffic.py -> profiles/traffic_<workload>.json).  A stored figure
    is quoted only for the kernels it was measured on: the file records the hash of the kernel sources, and a mismatch (or a
    different fleet size) prints null together with the reason."""
    name = f"traffic_{workload}.json"
    tj = _stored(name)
    if tj is None:
        return None, {"file": None, "note": "no PMC measurement stored for this workload"}
    src = {"file": "profiles/" + name, "measured_at_commit": tj.get("commit"), "kernel_src_hash": tj.get("kernel_src_hash"),
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes around this command, gfx950 correction applied; "
                  "read from the stored file, not measured in this run"}
    if tj.get("robots") != R:
        src["note"] = f"stored measurement is for {tj.get('robots')} robots, this run has {R}: not quoted"
        return None, src
    cur = _current_kernel_hash()
    if tj.get("kernel_src_hash") is None or cur is None or tj["kernel_src_hash"] != cur:
        src["note"] = "kernel sources changed since the counters were collected (hash mismatch): not quoted"
        src["current_kernel_src_hash"] = cur
        return None, src
    return tj.get("hbm_bytes_per_launch"), src


def executed_of(workload, R):
    """What the kernel executed, from tracked counter / stamp runs (profiles/executed_<workload>.json): the share of the
    pairs that are in range, VALU instructions per wave, issue-slot occupancy."""
    ej = _stored(f"executed_{workload}.json")
    if ej is None or ej.get("robots") != R:
        return None
    cur = _current_kernel_hash()
    ej = dict(ej)
    ej["stale"] = not (ej.get("kernel_src_hash") is not None and ej.get("kernel_src_hash") == cur)
    return ej


def build_config34(workload, args, dev, local_rank, rank, world, R, seed_rank=None, exch=None):
    """Engine + bound launch of configs 2 / 3 / 3b / 4 on this rank's shard.  Returns (one_step, eng, desc, table, s, spheres_np,
    keep): `one_step` issues one control step on the current stream."""
    import numpy as np
    import torch
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import ObstacleExchange
    table, desc = (Cf.config2 if workload == "config2" else Cf.config3)(args.solve)
    eng = Engine(desc, local_rank)
    # synthetic inputs, SURVEY 8(d): seed 1 -> performance inputs (rank-offset so shards differ)
    sr = rank if seed_rank is None else seed_rank
    s = Cf.sample_panda_states(np.random.default_rng(1 + 1000 * sr), R)
    q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
    out = torch.empty_like(q)
    keep = [q, qd, goal, out]
    spheres_np = None
    if workload == "config2":
        launch, _ = eng.bind(q, qd, goal, out=out)   # bare C-ABI call on fixed buffers
        return launch, eng, desc, table, s, None, keep
    K = Cf.N_SPHERES
    spheres_np = Cf.sample_spheres(np.random.default_rng(7), K)   # same table on every rank
    if workload == "config3c":
        spheres_np = Cf.sample_capsules(np.random.default_rng(7), K)
    if workload == "config3l":
        from riemannian_motion_policies_amd import urdf as U
        lc = U.link_capsules(U.PANDA_URDF, table, Cf.CONTROL_POINT_FRAMES)
        tbl_t, lc_t = torch.from_numpy(spheres_np).to(dev), torch.from_numpy(lc).to(dev)
        launch, _ = eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=tbl_t, link_capsules=lc_t), out=out)
        # for the result check: the same pairs as explicit arrays (closest-point stage), which the oracle reads
        p_link, p_obs = eng.closest_points(q[:256], eng.obstacles(spheres=tbl_t), link_capsules=lc_t)
        keep += [p_link, p_obs]
        return launch, eng, desc, table, s, spheres_np, keep
    if workload in ("config3", "config3c"):
        obstacles = eng.obstacles(spheres=torch.from_numpy(spheres_np).to(dev))
        launch, _ = eng.bind(q, qd, goal, obstacles=obstacles, out=out)
        return launch, eng, desc, table, s, spheres_np, keep
    if workload == "config3b":
        # interface B (reference-faithful, data_management.py:8-37): explicit closest-point pairs per robot, produced on
        # the GPU by the closest-point stage from the same sphere table, then read back by the step as [R, 256, 3] arrays
        tbl = eng.obstacles(spheres=torch.from_numpy(spheres_np).to(dev))
        p_link, p_obs = eng.closest_points(q, tbl)
        obstacles = eng.obstacles(p_link=p_link, p_obs=p_obs)
        launch, _ = eng.bind(q, qd, goal, obstacles=obstacles, out=out)
        keep += [p_link, p_obs]
        return launch, eng, desc, table, s, spheres_np, keep
    # config4
    if K % world:
        raise SystemExit("sphere count must divide by the world size")
    local = torch.from_numpy(spheres_np[rank * (K // world):(rank + 1) * (K // world)]).to(dev)
    if args.exchange == "native":
        # gather + stream orderings + launch inside librmp2_hip.so: one C-ABI call per control step (the torch.distributed
        # loop below costs ~50 us of host time per step -- more than the step kernel takes)
        from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
        if not isinstance(exch, NativeObstacleExchange):
            from riemannian_motion_policies_amd.fleet import agree_on_exchange
            # every rank takes the same exchange: one rank falling back alone would leave the others in a collective
            exch, err = agree_on_exchange(lambda: NativeObstacleExchange(K // world, dev, depth=args.exchange_depth), world, dev)
            if exch is None:
                print(f"bench.py: native exchange unavailable (rank {rank}: {err!r}); every rank falls back to --exchange torch", file=sys.stderr)
                args.exchange = "torch"
                args.exchange_fell_back = f"native exchange unavailable on at least one rank ({err!r}); every rank took --exchange torch"
                return build_config34(workload, args, dev, local_rank, rank, world, R, seed_rank=seed_rank, exch=None)
        while exch.pending < exch.depth:   # (a reused exchange -- the emulation -- still holds the previous user's gathers)
            exch.start(local, local_is_ready=True)
        # (the rank's slice is static in this benchmark -- as the fixed `local_ready` event of the torch-driven loop below --
        # so no producer event is put between two step kernels; a moving slice orders itself with next_local_is_ready=False)
        one_step = exch.bind(eng, q, qd, goal, out, next_local=local, next_local_is_ready=True)
        keep += [exch, local]
        return one_step, eng, desc, table, s, spheres_np, keep
    exch = exch or ObstacleExchange(K // world, dev)
    # all-gather on a side stream, pipelined one step ahead: the table of step k + 1 is gathered (into
    # the second buffer) while the kernel of step k runs; every step consumes a freshly gathered table
    local_ready = torch.cuda.Event()
    local_ready.record(torch.cuda.current_stream(dev))
    # one pre-marshalled launch per table buffer (the exchange alternates between two fixed buffers); each
    # launch signals "this table has been read" through its own completion (no event packet between steps)
    bound = {t.data_ptr(): eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out,
                                    done_fence=exch.reader_fence(t))[0]
             for t in exch.tables}
    exch.start(local, produced=local_ready)

    def one_step():
        # the gather of the NEXT step's table is issued before this step's kernel: it gets its few workgroups
        # while the GPU is between two steps, instead of queueing behind a kernel that fills every SIMD and
        # all of LDS (the orderings are unchanged: it waits for the reader fence of the buffer it overwrites)
        launch = bound[exch.finish().data_ptr()]
        exch.start(local, produced=local_ready)
        launch()
        exch.consumed(attached=True)
    keep += [exch, local, bound]
    return one_step, eng, desc, table, s, spheres_np, keep


def world1_leg(args, dev, local_rank, rank, world, R, timer):
    """The like-for-like base of a weak-scaling figure: the SAME workload (config 4: this rank's 65 536 robots, the whole sphere
    table gathered through an exchange every step) with a communicator of ONE rank -- every rank times its own shard with a
    world-1 exchange of its own, at the same moment, after the N-rank timed region; the line carries the MAX over ranks
    (`ms_per_step` of the N-rank job divided by this is the weak-scaling efficiency, whatever the driver compares N = 1 with).
    Never a reason to lose the line: any failure is reported as null with the reason."""
    import torch
    import torch.distributed as dist
    info = {"what": "config 4 at world 1 on every rank's own shard (own one-rank RCCL communicator), MAX over ranks",
            "exchange": args.exchange, "steps": min(args.steps, 500)}
    ms, err = float("nan"), None
    try:
        exch1 = None
        if args.exchange == "native":
            from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
            from riemannian_motion_policies_amd import configs as Cf
            exch1 = NativeObstacleExchange(Cf.N_SPHERES, dev, depth=args.exchange_depth, rank=0, world=1,
                                           uid=NativeObstacleExchange.unique_id())
        else:
            # (the torch-driven exchange would gather over the job's DEFAULT group: N ranks, not one -- a one-rank exchange is its
            #  local-copy form; no collective of the job is issued inside this leg)
            from riemannian_motion_policies_amd.fleet import ObstacleExchange
            from riemannian_motion_policies_amd import configs as Cf
            exch1 = ObstacleExchange(Cf.N_SPHERES, dev, collective=False)
            info["what"] = "config 4 at world 1 on every rank's own shard (torch-driven exchange of one rank: a local copy), MAX over ranks"
        import copy
        a1 = copy.copy(args)
        one_step, eng1, *_rest = build_config34("config4", a1, dev, local_rank, 0, 1, R, seed_rank=rank, exch=exch1)
        k1 = Timed(dev, False).run(one_step, min(args.steps, 500), min(args.warmup, 50))
        ms = k1["dt"] / min(args.steps, 500) * 1e3
        info["kernel"] = eng1.last_kernel()
        del one_step, eng1, _rest
    except Exception as e:   # noqa: BLE001
        err = repr(e)
    t = torch.tensor([ms if err is None else float("inf")], dtype=torch.float64, device=dev)
    if timer.use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    v = float(t.item())
    if err is not None or v == float("inf"):
        info["error"] = err or "another rank failed"
        return None, info
    info["this_rank_ms"] = ms
    return v, info


TOLERANCE_RULE = ("every checked robot must pass one of, all against the CPU oracle's fp64 evaluation of the reference's formulae (`exact`): "
                  "(A) |qdd - exact|_inf <= 1e-5 * max(1, |exact|_inf) [north star]; "
                  "(B) normwise backward error against the exact system (M, f) <= 2e-5, with the forward bound it implies and the "
                  "minimum-norm check for rank-dropping resolves; (E) |qdd - exact|_inf <= 2 x the robot's fp32 envelope (the "
                  "largest distance from the fp64 evaluation among 17 fp32 evaluations of the reference's formulae on inputs moved by "
                  "an fp32 rounding: what ANY fp32 evaluation leaves on a near-contact robot) -- oracle/oracle.py accuracy_gate; no "
                  "robot is exempted, none may fail")


def check_against_oracle(desc, s, spheres_np, out, n=256, what="", pairs=None, extra_kw=None):
    """Result check of a bench workload (outside every timed region): the first n robots against the CPU oracle -- a smoke alarm
    on the buffers the timed steps wrote, not the parity statement (tests/ is).  pairs = (p_link, p_obs) device tensors: the
    oracle reads the same explicit pairs (interface B).  extra_kw: further oracle arguments (ragged lists)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    kw = dict(spheres=spheres_np) if spheres_np is not None else {}
    if pairs is not None:
        kw = dict(p_link=pairs[0][:n].cpu().numpy(), p_obs=pairs[1][:n].cpu().numpy())
    kw.update(extra_kw or {})
    got = out[:n].cpu().numpy()
    # perf inputs are unrestricted: near-contact robots carry |qdd| of 1e2..1e3 -- they are bounded by (B) / (E)
    # every clause against the EXACT (fp64) evaluation of the reference's formulae -- its q-double-dot and its system (M, f); fp32
    # evaluations of them enter through the envelope of clause E
    exact = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], precision="f64", **kw)
    verdict = O.accuracy_gate(got, exact, truth=exact["qdd64"], envelope=O.fp32_envelope(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], **kw))
    summary = O.gate_summary(verdict)
    if not verdict["ok"].all():
        raise SystemExit(f"bench.py: {what} result check FAILED against the oracle: {summary}")
    each = verdict["each"]
    return {"robots_checked": int(n), "max_abs_err": summary["worst_abs_err"], "admitted_by": {
        "A_north_star_1e-5": summary["north_star_1e-5"], "B_backward_error_2e-5": summary["backward_error"],
        "E_fp32_envelope_x2": summary["fp32_envelope"], "nan_in_oracle_and_engine": summary["both_nan"]},
        "passing_each_clause_on_its_own": {"A": int(each["a"].sum()), "B": int(each["b"].sum()), "E": int(each["e"].sum())},
        "rejected": summary["rejected"], "tolerance": TOLERANCE_RULE}


def link_pairs_for_lists(desc, lc, table, q, off, idx):
    """Explicit pairs (what the oracle reads) of a ragged fleet with link geometry, for the result check: per robot and
    distance leaf one pair per LIST ENTRY -- the nearest points of the link's capsule and the listed primitive, fp64 closed form
    on the oracle's fp64 forward kinematics -- and far-away fillers (metric exactly 0) up to the longest list."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf, descriptor as D
    frames = [desc.leaves[i].frame for i in D.distance_leaf_indices(desc)]
    T = O.forward_kinematics(desc, q, precision="f64")
    pl_all, po_all = Cf.pairs_from_link_capsules(T[:, frames], lc, table)
    n, L, K = len(q), len(frames), table.shape[0]
    kmax = max(int(np.diff(off[: n + 1]).max()), 1)
    pl = np.zeros((n, L * kmax, 3), np.float32)
    po = np.full((n, L * kmax, 3), 1.0e3, np.float32)
    for r in range(n):
        lst = idx[off[r]:off[r + 1]]
        for l in range(L):
            pl[r, l * kmax:l * kmax + len(lst)] = pl_all[r, l * K + lst]
            po[r, l * kmax:l * kmax + len(lst)] = po_all[r, l * K + lst]
    return pl, po


def emulate_world(args, workload, dev, local_rank, use_dist):
    """Single-GPU EMULATION of an N-rank run: build each of the N rank shards one after the other on THIS GPU, time each
    on its own, and report per-rank time, the maximum, and the throughput N GPUs would reach if each ran its shard at the
    speed measured here.  Config 5 has no data-path collective, so the slowest shard IS the N-GPU step; config 4's
    per-rank step carries the RCCL all-gather at world 1 (a lower bound on its latency at world N)."""
    import gc
    import numpy as np
    import torch
    from riemannian_motion_policies_amd.fleet import MixedFleetShard
    W = args.emulate_world
    wl = WORKLOADS[workload]
    R = args.robots or wl["robots"]
    steps, warmup = min(args.steps, 500), min(args.warmup, 50)
    rows, extra = [], {}
    if workload == "config5":
        curves = MixedFleetShard.calibrate_curves(local_rank) if args.calibrate else MixedFleetShard.DEFAULT_CURVES
        extra["cost_model"] = {"kind": "kernel time curves per robot type (us per step at a ladder of fleet sizes)",
                               "curves": {k: {"robots": list(v[0]), "us": [round(float(x), 2) for x in v[1]]} for k, v in curves.items()}}
        extra["cost_model_source"] = "measured in this run (--calibrate)" if args.calibrate else "fleet.MixedFleetShard.DEFAULT_CURVES"
        plans = {"calibrated": {"curves": curves}}
        if args.compare_flop_model:   # the round-2 weights (SURVEY 8(d) flops), for the before / after of the imbalance
            plans = {"flop_model_round2": {"two_joint": (0.5e3, 240.0), "panda": (4.0e3, 240.0)}, **plans}
        for pname, c in plans.items():
            prow = []
            for r in range(W):
                shard = MixedFleetShard.synthetic(R * W, W, r, local_rank, seed=5, solve=args.solve, cost=c)
                if args.graph:
                    shard.capture()
                k = Timed(dev, False).run(shard.step, steps, warmup)
                prow.append({"rank": r, "two_joint": shard.n_two_joint, "panda": shard.n_panda,
                             "us_per_step": k["dt"] / steps * 1e6, "est_us": shard.work / 1e3 if pname == "calibrated" else None,
                             "kernels": {key: p["engine"].last_kernel() for key, p in shard.parts.items()}})
                del shard
                gc.collect()
                torch.cuda.synchronize(dev)
            extra.setdefault("plans", {})[pname] = prow
        rows = extra["plans"]["calibrated"]
    else:
        # ONE exchange (one side stream) for all emulated ranks: HIP maps streams onto hardware queues in creation order,
        # and a fresh side stream per rank lands on the compute stream's queue every few ranks (measured: 156 us steps)
        from riemannian_motion_policies_amd import configs as Cf
        from riemannian_motion_policies_amd.fleet import NativeObstacleExchange, ObstacleExchange
        native = args.exchange == "native"
        exch = NativeObstacleExchange(Cf.N_SPHERES, dev, depth=args.exchange_depth) if native else ObstacleExchange(Cf.N_SPHERES, dev)
        if native:
            exch.set_peer_wait(True)   # (time the orderings an N-rank exchange needs, not the one-rank shortcut)
        for r in range(W):
            while not native and exch._pending:   # drain the previous rank's outstanding gather
                exch.finish()
            one_step, eng, desc, table, s, spheres_np, keep = build_config34(
                workload, args, dev, local_rank, 0, 1, R, seed_rank=r, exch=exch)
            k = Timed(dev, use_dist).run(one_step, steps, warmup)
            chk = check_against_oracle(desc, s, spheres_np, keep[3], what=f"{workload} rank {r}")
            rows.append({"rank": r, "robots": R, "us_per_step": k["dt"] / steps * 1e6, "kernel_us": k["kernel_ms"] * 1e3,
                         "kernel": eng.last_kernel(), "result_check": chk})
            del one_step, eng, keep
            gc.collect()
            torch.cuda.synchronize(dev)
    t = np.array([x["us_per_step"] for x in rows])
    total = R * W
    line = {
        "metric": "RMP2 control steps/sec (batched robots)",
        "value": total / (t.sum() * 1e-6),
        "unit": "robot control steps/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": float(t.sum()) * 1e-3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (leaves, Jacobians, pull-back) + f64 (sum over leaves, resolve)", "data": "synthetic",
        "config": {"workload": wl["name"], "workload_key": workload, "robots_per_gpu": R, "solve": args.solve,
                   "parallelism": f"EMULATION on one GPU of a {W}-rank job: the {W} rank shards run one after the other"},
        "value_note": f"measured: all {W} shards of the {total}-robot fleet on ONE GPU, sequentially",
        "emulated_scaling": {
            "label": "PREDICTED from a single-GPU emulation -- not a multi-GPU measurement",
            "world": W, "total_robots": total, "per_rank": rows,
            "max_us": float(t.max()), "min_us": float(t.min()), "mean_us": float(t.mean()),
            "imbalance_max_over_mean": float(t.max() / t.mean()),
            f"predicted_{W}gpu_steps_per_s": total / (t.max() * 1e-6),
            "assumes": ("no data-path collective (config 5): the slowest shard is the step" if workload == "config5" else
                        "per-rank step = kernel + RCCL all-gather measured at world 1: a LOWER bound on the exchange latency "
                        "at world N (the gather then crosses xGMI)"),
        },
    }
    line["emulated_scaling"].update({k: v for k, v in extra.items() if k != "plans"})
    if "plans" in extra and len(extra["plans"]) > 1:
        line["emulated_scaling"]["plans"] = {
            k: {"per_rank_us": [x["us_per_step"] for x in v], "shards": [[x["two_joint"], x["panda"]] for x in v],
                "max_us": max(x["us_per_step"] for x in v),
                "imbalance_max_over_mean": max(x["us_per_step"] for x in v) / (sum(x["us_per_step"] for x in v) / len(v))}
            for k, v in extra["plans"].items()}
    emit(line)
    return 0


def worker(args) -> int:
    # HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With 4, the stream that
    # carries the obstacle gather can land on the compute stream's queue (it depends on the order in which torch, c10d
    # and the exchange create their streams): the gather then serialises behind the control-step kernel it is meant to
    # overlap -- measured 100.9 us per config-4 step against 80.3 us with 2 or 8 queues.  Read by
    # the HIP runtime when it initialises, i.e. below.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np
    import torch
    import torch.distributed as dist

    from riemannian_motion_policies_amd import configs as Cf

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_one_gpu:
        local_rank = 0      # every rank on device 0 (a launcher's LOCAL_RANK is ignored): see --rehearse-one-gpu
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the RMP2 engine has no CPU path")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: no HIP device {local_rank} on this node")
    workload = args.workload
    if workload == "auto":
        workload = "config3" if world == 1 else "config4"
    if args.emulate_world and (world != 1 or workload not in ("config4", "config5")):
        raise SystemExit("--emulate-world N: one GPU (--gpus 1), --workload config4 or config5")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # the process group exists whenever a launcher set RANK (even for one rank: RCCL init, barriers and the MAX-reduce
    # are then exercised on a one-GPU box too) and for config4, whose defining element is the RCCL exchange
    use_dist = "RANK" in os.environ or workload == "config4"
    if use_dist:
        if "RANK" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
        # keep stdout to the ONE JSON line: RCCL prints its banner on fd 1 while the communicator is created
        sys.stdout.flush()
        saved_fd1 = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.rehearse_one_gpu:
                # RCCL refuses two ranks on one device; gloo carries the barriers, the MAX-reduce, the rank agreement and
                # (after the native exchange's communicator has failed on every rank) the torch-driven obstacle exchange
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
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
    from riemannian_motion_policies_amd.fleet import MixedFleetShard

    if args.emulate_world:
        rc = emulate_world(args, workload, dev, local_rank, use_dist)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return rc

    timer = Timed(dev, use_dist)
    wl = WORKLOADS[workload]
    R = args.robots or wl["robots"]
    line_extra = {}
    spheres_np = None
    desc = table = s = None
    bound = "valu"

    if workload in ("config2", "config3", "config3b", "config3c", "config3l", "config4"):
        one_step, eng, desc, table, s, spheres_np, keep = build_config34(workload, args, dev, local_rank, rank, world, R)
        kern = timer.run(one_step, args.steps, args.warmup)
        # result check, outside the timed region: the buffers the timed steps wrote, against the oracle
        line_extra["result_check"] = check_against_oracle(desc, s, spheres_np, keep[3], what=workload,
                                                          pairs=(keep[4], keep[5]) if workload in ("config3b", "config3l") else None)
        if workload == "config4":
            ex = next(k for k in keep if hasattr(k, "start") and hasattr(k, "world"))
            line_extra["exchange"] = args.exchange
            if getattr(args, "exchange_fell_back", None):
                line_extra["exchange_fell_back"] = args.exchange_fell_back
            # from the communicator the exchange joined (the library's own for the native exchange, c10d's otherwise), not from
            # WORLD_SIZE: the driver can hold it against --gpus
            line_extra["rccl_nranks"] = int(ex.nranks) if args.exchange == "native" else (
                None if args.rehearse_one_gpu else int(dist.get_world_size()))     # (a rehearsal's process group is gloo's)
            line_extra["exchange_depth"] = int(args.exchange_depth) if args.exchange == "native" else 1
        if workload == "config4" and world > 1:
            line_extra["world1_same_workload_ms"], line_extra["world1_same_workload"] = world1_leg(
                args, dev, local_rank, rank, world, R, timer)
        bytes_rs, flops_rs = wl["bytes"], wl["flops"]
        per_launch_bytes, per_launch_flops = bytes_rs * R, flops_rs * R
        exe = executed_flops(eng, desc, keep[0], spheres_np, workload)
        kernel_name = eng.last_kernel() + " (chosen by fleet size, rmp2_hip.hip dispatch_solve)"
        total_robots = R * world
        bound = wl.get("bound", "valu")
        parallelism = f"robot-batch split x{world}" + (
            f", RCCL all-gather of the sphere table per step (side stream, exchange = {args.exchange}, tables gathered {args.exchange_depth if args.exchange == 'native' else 1} step(s) ahead)" if workload == "config4" else "")
    else:
        # ---- config 5: type-sorted mixed fleet, ragged obstacle lists, cost-balanced cut across the ranks ----
        cost = None
        if args.calibrate:   # rank 0 measures the time curves, every rank cuts with the same numbers
            n = len(MixedFleetShard.CURVE_SIZES)
            c = torch.zeros(2 * n, dtype=torch.float64, device=dev)
            if rank == 0:
                m = MixedFleetShard.calibrate_curves(local_rank)
                c = torch.tensor(list(m["two_joint"][1]) + list(m["panda"][1]), dtype=torch.float64, device=dev)
            if use_dist:
                dist.broadcast(c, 0)
            c = c.tolist()
            sz = list(MixedFleetShard.CURVE_SIZES)
            cost = {"curves": {"two_joint": (sz, c[:n]), "panda": (sz, c[n:])}}
            line_extra["cost_model"] = {"robots": sz, "two_joint_us": c[:n], "panda_us": c[n:]}
        shard = MixedFleetShard.synthetic(R * world, world, rank, local_rank, seed=5, solve=args.solve, cost=cost,
                                          link_geometry=args.link_geometry)
        if args.link_geometry:
            line_extra["link_geometry"] = "closest points on the links' capsules, formed inside the step over the ragged lists"
        # (a HIP graph of the two-stream step replays SLOWER than the eager sequence on this runtime -- 62.1 against 40.3 us per
        # step, profiles/r03_config5_graph_ab.txt: its cross-stream edges become full barriers -- so eager is the default)
        line_extra["step_issue"] = "hip graph replay" if (args.graph and shard.capture()) else "eager (<= 6 host calls)"
        one_step = shard.step
        exe = None
        kern = timer.run(one_step, args.steps, args.warmup)
        # the dominant kernel (the Panda engine's) timed on its own right after the timed region, same buffers
        kk = Timed(dev, False).run(shard.step_dominant, min(args.steps, 200), 10)
        kern.update(kernel_ms=kk["kernel_ms"], grp=kk["grp"], raw_ms=kk["raw_ms"], floor_ms=kk["floor_ms"])
        # result check of BOTH robot types of this rank's shard (outside the timed region), ragged lists and all
        chk = {}
        for key, part in shard.parts.items():
            pq, pqd, pgoal, _ = part["keep"]
            m = min(256, part["n"])
            off = part["host"]["csr_offset"][: m + 1]
            host = {"q": pq[:m].cpu().numpy(), "qd": pqd[:m].cpu().numpy(), "goal": pgoal[:m].cpu().numpy()}
            if args.link_geometry:   # the oracle reads explicit pairs: one per list entry (fp64 closed form), far fillers beyond
                pl_, po_ = link_pairs_for_lists(part["desc"], part["host"]["link_capsules"], part["host"]["spheres"], host["q"],
                                                off, part["host"]["csr_index"])
                chk[key] = check_against_oracle(part["desc"], host, None, part["out"], n=m, what=f"config5 {key} (link geometry)",
                                                extra_kw=dict(p_link=pl_, p_obs=po_))
            else:
                chk[key] = check_against_oracle(part["desc"], host, part["host"]["spheres"], part["out"], n=m, what=f"config5 {key}",
                                                extra_kw=dict(csr_offset=off, csr_index=part["host"]["csr_index"][: off[-1]]))
            chk[key].pop("tolerance")
        chk["tolerance"] = TOLERANCE_RULE
        line_extra["result_check"] = chk
        per_launch_bytes, per_launch_flops = shard.dominant_bytes, shard.dominant_flops
        bytes_rs = per_launch_bytes / max(shard.dominant_robots, 1)
        flops_rs = per_launch_flops / max(shard.dominant_robots, 1)
        kernel_name = shard.dominant_kernel() + " (Panda engine of this rank's shard; timed on its own after the timed region)"
        total_robots = R * world
        parallelism = (f"type-sorted fleet cut x{world} by calibrated per-robot cost (fleet.MixedFleetShard.plan); "
                       "one engine per robot type per rank; no data-path collective")
        counts = torch.tensor([float(shard.n_two_joint), float(shard.n_panda), float(shard.work)], dtype=torch.float64, device=dev)
        if use_dist:
            allc = [torch.zeros_like(counts) for _ in range(world)]
            dist.all_gather(allc, counts)
        else:
            allc = [counts]
        line_extra["shards"] = [{"two_joint": int(c[0]), "panda": int(c[1]), "est_cost_us": float(c[2]) / 1e3} for c in allc]

    value = total_robots * args.steps / kern["dt"]
    rc = 0
    if rank == 0:
        line = {
            "metric": "RMP2 control steps/sec (batched robots)",
            "value": value,
            "unit": "robot control steps/s",
            "n_gpus": 1 if args.rehearse_one_gpu else world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": kern["dt"] / args.steps * 1e3,
            "host_issue_ms_per_step": kern["host_issue_ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (leaves, Jacobians, pull-back) + f64 (sum over leaves, resolve)",
            "data": "synthetic",
            "config": {"workload": wl["name"], "workload_key": workload, "robots_per_gpu": R, "solve": args.solve,
                       "parallelism": parallelism},
            "roofline": roofline_obj(kernel_name, kern, per_launch_bytes, per_launch_flops, bytes_rs, flops_rs,
                                     "config3" if workload == "config4" else workload, R, bound, exe),
        }
        line.update(line_extra)
        if args.rehearse_one_gpu:
            line["rehearsal"] = {
                "ranks": world, "gpus": 1, "backend": "gloo",
                "note": f"{world} rank processes time-sharing ONE GPU: executes the launcher, the rank agreement, the exchange's "
                        "all-rank fall-back and the barrier / MAX-over-ranks timing of the N > 1 path on a one-GPU box.  NOT a "
                        "scaling measurement: `value` is the ranks' robots over the time they took sharing the device"}
        if workload in ("config3", "config4") and not args.no_secondary:
            # BASELINE configs[1] in the same process: the latency-bound fleet (4096 robots: one wave per SIMD)
            t2, d2 = Cf.config2(args.solve)
            eng2 = Engine(d2, local_rank)
            s2 = Cf.sample_panda_states(np.random.default_rng(1), 4096)
            q2, qd2, g2 = (torch.from_numpy(s2[k]).to(dev) for k in ("q", "qd", "goal"))
            o2 = torch.empty_like(q2)
            launch2, _ = eng2.bind(q2, qd2, g2, out=o2)
            k2 = Timed(dev, False).run(launch2, 2000, 200)
            w2 = WORKLOADS["config2"]
            line["secondary"] = {
                "workload": w2["name"], "robots": 4096, "value": 4096 * 2000 / k2["dt"], "unit": "robot control steps/s",
                "steps": 2000, "warmup": 200, "ms_per_step": k2["dt"] / 2000 * 1e3, "n_gpus": 1,
                "roofline": roofline_obj(eng2.last_kernel(), k2, w2["bytes"] * 4096, w2["flops"] * 4096, w2["bytes"], w2["flops"],
                                         "config2", 4096, "valu"),
                "note": "latency regime: 1024 waves on 1024 SIMDs, ~3 us of the launch is dispatch floor (DESIGN.md section 5)"}
        if workload == "config3" and not args.no_secondary:
            # the same workload under the OTHER resolve, same process, same inputs.  The line's `value` is on solve = "pinv", the
            # reference's only resolve (rmp.py:153-154: one certifying launch); "auto" (plain elimination,
            # pseudo-inverse for flagged robots only) rides along so that what reference semantics cost stays visible
            other = "auto" if args.solve == "pinv" else "pinv"
            _, dp = Cf.config3(other)
            engp = Engine(dp, local_rank)
            qp, qdp, gp = keep[0], keep[1], keep[2]
            op = torch.empty_like(qp)
            launchp, _ = engp.bind(qp, qdp, gp, obstacles=engp.obstacles(spheres=torch.from_numpy(spheres_np).to(dev)), out=op)
            kp = Timed(dev, False).run(launchp, min(args.steps, 1000), min(args.warmup, 100))
            torch.cuda.synchronize(dev)
            line["solve_" + other] = {
                "workload": wl["name"] + f', solve = "{other}"', "robots": R,
                "ms_per_step": kp["dt"] / min(args.steps, 1000) * 1e3, "value": R * min(args.steps, 1000) / kp["dt"],
                "unit": "robot control steps/s", "kernel": engp.last_kernel(), "kernel_ms": kp["kernel_ms"],
                "max_abs_diff_to_" + args.solve: float((op - keep[3]).abs().nan_to_num(0.0).max().item()),
                "note": ("plain elimination; the pseudo-inverse only for robots the elimination flags" if other == "auto" else
                         "full rank certified per robot inside the elimination (pinv = inv there), Jacobi pseudo-inverse for the rest")}
        if not args.no_cpu_baseline and world == 1 and desc is not None:
            line["cpu_baseline"] = cpu_baseline(workload, desc, table, s, spheres_np)
        emit(line)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc


REAL_STDOUT = None   # fd of the process's real stdout once main() has pointed fd 1 at stderr


def emit(line) -> None:
    """The ONE JSON line, on the real stdout.  main() points fd 1 at stderr for the rest of the process: RCCL prints a version banner
    on fd 1 when a process creates its first communicator (torch's, or the library's own in a rehearsal), rocprofv3 and hipcc talk
    there too, and under torch.distributed.run every rank shares the launcher's stdout -- none of it may reach the line's reader."""
    text = json.dumps(line) + "\n"
    if REAL_STDOUT is None:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        os.write(REAL_STDOUT, text.encode())


def preflight():
    """Everything that may start a child process or read the host, done BEFORE this process can hold a HIP context (a process
    with one must not fork + exec on this pool): the host summary, the library build (hipcc, only when the sources changed;
    rank 0 of a launcher, or the lone process) and the checker's library (oracle/: `make`, only when stale -- loading the
    checker here is not using it; it is called after the timed region, in check_against_oracle / cpu_baseline)."""
    global HOST_SUMMARY
    HOST_SUMMARY = lscpu_summary()
    if os.environ.get("RANK", "0") == "0":
        import __graft_entry__ as ge
        ge.build_hip()
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        O.lib()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--robots", type=int, default=0, help="robots per GPU (default: the workload's)")
    ap.add_argument("--solve", default="pinv", choices=["auto", "pinv"],
                    help='pinv (default): the reference\'s only resolve, tf.linalg.pinv(M) f (rmp.py:153-154); auto: elimination with '
                         'the pseudo-inverse kept for flagged robots -- rides along as the nested "solve_auto" leg of the default line')
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="config4 / config5 on ONE GPU: build and time each of the N rank shards in turn, report per-rank "
                         "times and the predicted N-GPU throughput (clearly labelled as an emulation)")
    ap.add_argument("--calibrate", action="store_true",
                    help="config5: measure the per-type cost model (ns per robot, ns per pair) on rank 0 before cutting the fleet")
    ap.add_argument("--calibrate-robots", type=int, default=16384)
    ap.add_argument("--compare-flop-model", action="store_true",
                    help="--emulate-world with config5: also time the shards the round-2 flop-model weights would cut")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="config4: obstacle exchange inside librmp2_hip.so (one C-ABI call per step) or driven from Python "
                         "through torch.distributed (A/B)")
    ap.add_argument("--exchange-depth", type=int, default=1, choices=[1, 2],
                    help="config4, native exchange: tables gathered this many control steps ahead (1: every step reads obstacles "
                         "one control step old, as the reference's loop does; 2: one more step of staleness)")
    ap.add_argument("--link-geometry", action="store_true",
                    help="config5: control points on the links' capsules (fitted to the Panda's meshes / the TwoJoint URDF's "
                         "primitives) instead of the frame origins, formed inside the step over the ragged lists")
    ap.add_argument("--graph", action="store_true", help="config5: replay the shard's step as a HIP graph (A/B: measured slower than eager)")
    ap.add_argument("--rank-timeout", type=float, default=540.0,
                    help="--gpus N without a launcher: seconds after which ranks that are still running are ended and the job fails")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="--gpus N on a node with ONE GPU: every rank on device 0, torch's gloo backend instead of RCCL (which "
                         "refuses two ranks on one device).  Executes the N > 1 control flow; the line is labelled a rehearsal")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    preflight()
    sys.exit(worker(args))


if __name__ == "__main__":
    main()
