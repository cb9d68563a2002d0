"""BASELINE config 4's defining element on real hardware: the sphere table all-gathered over RCCL (torch.distributed
backend "nccl") on a side stream, double-buffered one step ahead, feeding the HIP control step.

World size 1 -- the driver's GPU box has one device -- but the collective is the real one (ObstacleExchange issues
all_gather_into_tensor whenever a process group exists), and so are the stream / event orderings between the producer of
the local slice, the gather, the consuming kernel and the gather that overwrites the buffer two steps later.  The
process group lives in THIS process: a GPU-initialised process must not fork + exec children on this pool.  Ranks > 1
are covered on CPU by tests/test_fleet_gloo.py (gloo, world size 2)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL = 1e-5


@pytest.fixture(scope="module")
def nccl_group(hip_lib):
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    created = False
    if not dist.is_initialized():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


def test_rccl_obstacle_exchange_feeds_the_hip_step(nccl_group, golden_dir):
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import ObstacleExchange
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = Engine(desc, 0)
    q, qd, goal = (torch.from_numpy(g[k]).to(dev) for k in ("q", "qd", "goal"))
    K = len(g["spheres"])
    exch = ObstacleExchange(K, dev)
    assert exch.collective and exch.world == 1
    # the obstacles move between control steps (the reference's "dynamic" obstacles are static bodies whose positions
    # change from step to step, quirk Q6): step k must see table k, through whichever of the two buffers it lands in
    tables = [g["spheres"].copy() for _ in range(5)]
    for k, t in enumerate(tables):
        t[:, :2] *= np.float32(1.0 + 0.03 * k)   # pushed outwards, away from the arms: clearance only grows (>= 0.05 m kept)
    local = torch.from_numpy(tables[0]).to(dev)
    produced = torch.cuda.Event()
    produced.record(torch.cuda.current_stream(dev))
    exch.start(local, produced=produced)
    outs, seen = [], []
    for k in range(4):
        tbl = exch.finish()                       # current stream waits for the gather of table k
        seen.append(tbl.data_ptr())
        outs.append(eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=tbl)).clone())
        exch.consumed()                           # the gather that reuses this buffer must wait for this kernel
        local.copy_(torch.from_numpy(tables[k + 1]).to(dev))   # producer of the next local slice, on the current stream
        produced = torch.cuda.Event()
        produced.record(torch.cuda.current_stream(dev))
        exch.start(local, produced=produced)      # gathered into the OTHER buffer while the kernel of step k runs
    exch.finish()
    torch.cuda.synchronize(dev)
    assert len(set(seen)) == 2 and seen[0] == seen[2] and seen[1] == seen[3], "the two table buffers must alternate"
    assert "quad" in eng.last_kernel() or "hex" in eng.last_kernel()
    for k in range(4):
        ref = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[k])["qdd64"]
        err = np.abs(outs[k].cpu().numpy() - ref).max(axis=1)
        tol = ATOL * np.maximum(1.0, np.abs(ref).max(axis=1))
        assert (err <= tol).all(), f"step {k}: worst {err.max():.3e}"
    # the tables differ enough for a stale buffer to be caught
    ref0 = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[0])["qdd64"]
    ref3 = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[3])["qdd64"]
    assert np.abs(ref0 - ref3).max() > 1e-3


def test_bound_launches_signal_the_reader_fence_themselves(nccl_group, golden_dir):
    """bench.py's config-4 loop: one pre-marshalled launch per table buffer, each carrying the exchange's reader fence as
    its completion fence (rmp2_set_step_fence) instead of an event recorded behind it; the producer of the local slice
    orders itself with a device-scope fence too.  Moving tables, many steps: a gather that overwrote a table too early,
    or a step that read one too early, shows as a wrong q-double-dot."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import ObstacleExchange, _Fence
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = Engine(desc, 0)
    q, qd, goal = (torch.from_numpy(g[k]).to(dev) for k in ("q", "qd", "goal"))
    K, steps = len(g["spheres"]), 12
    exch = ObstacleExchange(K, dev)
    tables = [g["spheres"].copy() for _ in range(steps + 1)]
    for k, t in enumerate(tables):
        t[:, :2] *= np.float32(1.0 + 0.02 * k)
    staged = [torch.from_numpy(t).to(dev) for t in tables]
    local = staged[0].clone()
    out = torch.empty_like(q)
    bound = {t.data_ptr(): eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out,
                                    done_fence=exch.reader_fence(t))[0] for t in exch.tables}
    produced = _Fence(dev)
    cur = torch.cuda.current_stream(dev)
    produced.record(cur)
    exch.start(local, produced=produced)
    outs = []
    for k in range(steps):
        bound[exch.finish().data_ptr()]()
        exch.consumed(attached=True)
        outs.append(out.clone())
        local.copy_(staged[k + 1])               # next slice, on the current stream, behind the launch
        produced.record(cur)
        exch.start(local, produced=produced)
    exch.finish()
    # an un-fenced step afterwards detaches the fence again (the handle's fence is sticky by design)
    plain = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=staged[3]))
    torch.cuda.synchronize(dev)
    # this test is about WHICH table a step read, not about near-contact rounding (the outward drift brings some spheres
    # within 0.05 m of an arm, where fp32 FK rounding is amplified to a few 1e-5, tests/test_gpu_kernel_variants.py):
    # 1e-4 relative, against tables that differ by > 1e-3 from one step to the next
    refs = [O.step(desc, g["q"], g["qd"], g["goal"], spheres=t)["qdd64"] for t in tables[:steps]]
    for k in list(range(steps)) + [None]:
        ref = refs[3 if k is None else k]
        got = (plain if k is None else outs[k]).cpu().numpy()
        err = np.abs(got - ref).max(axis=1)
        tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=1))
        assert (err <= tol).all(), f"step {k}: worst {err.max():.3e}"
    assert min(np.abs(refs[k] - refs[k + 1]).max() for k in range(steps - 1)) > 1e-3
    with pytest.raises(RuntimeError):
        exch.start(local); exch.start(local); exch.start(local)


@pytest.mark.parametrize("depth", [1, 2])
def test_native_exchange_one_call_per_step_against_the_oracle(hip_lib, golden_dir, depth):
    """The exchange inside librmp2_hip.so (rmp2_exchange_*: RCCL bound at run time, communicator of its own): gather,
    stream orderings and the step launch are one C-ABI call per control step.  The table moves every step; step k must
    see table k through whichever buffer it lands in (depth + 1 buffers, tables gathered `depth` steps ahead), the slices
    are rewritten on the launch stream between steps (a gather must wait for that producer, and must not overwrite a buffer
    an earlier step is still reading)."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = Engine(desc, 0)
    reps = 64   # 4 096 robots: the kernels take long enough for a missing ordering to show
    q, qd, goal = (torch.from_numpy(np.tile(g[k], (reps, 1))).to(dev) for k in ("q", "qd", "goal"))
    out = torch.empty_like(q)
    K, steps = len(g["spheres"]), 6
    tables = [g["spheres"].copy() for _ in range(steps + depth)]
    for k, t in enumerate(tables):
        t[:, :2] *= np.float32(1.0 + 0.03 * k)
    dev_tables = [torch.from_numpy(t).to(dev) for t in tables]
    # the slice of a gather must stay untouched until the gather has run: one slice buffer per outstanding gather (+ 1)
    slices = [torch.empty_like(dev_tables[0]) for _ in range(depth + 1)]
    exch = NativeObstacleExchange(K, dev, depth=depth)
    assert exch.world == 1
    for k in range(depth):                      # prime: tables 0 .. depth - 1
        slices[k % (depth + 1)].copy_(dev_tables[k])
        exch.start(slices[k % (depth + 1)])
    assert exch.pending == depth
    outs, seen = [], []
    for k in range(steps):
        sl = slices[(k + depth) % (depth + 1)]
        sl.copy_(dev_tables[k + depth])         # producer of table k + depth on the launch stream, right before the call that gathers it
        exch.step(eng, q, qd, goal, out, next_local=sl)   # waits for table k, gathers table k + depth, launches step k
        seen.append(exch._table.value)
        outs.append(out[: g["q"].shape[0]].clone())
    torch.cuda.synchronize(dev)
    assert len(set(seen)) == 3 and all(seen[i] == seen[i + 3] for i in range(steps - 3)), \
        "the three table buffers must rotate (at either depth)"
    for k in range(steps):
        # (about WHICH table a step read, not about near-contact rounding: 1e-4 relative against tables that differ by
        # more than 1e-3 from one step to the next -- as test_bound_launches_signal_the_reader_fence_themselves)
        ref = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[k])["qdd64"]
        err = np.abs(outs[k].cpu().numpy() - ref).max(axis=1)
        tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=1))
        assert (err <= tol).all(), f"step {k}: worst {err.max():.3e}"
        if k:
            prev = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[k - 1])["qdd64"]
            assert np.abs(ref - prev).max() > 1e-3
    with pytest.raises(Exception, match="outstanding"):
        for _ in range(depth + 2):
            exch.start(slices[0])
    exch.close()
