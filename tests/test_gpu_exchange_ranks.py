"""The nranks > 1 branch of the native obstacle exchange (include/rmp2.h rmp2_exchange_*, csrc/rmp2_hip.hip) on the
one-GPU test box: TWO ranks in ONE process on ONE GPU, one host thread and one HIP stream per rank, against a stand-in
collective library (tests/stub_rccl.hip: the five RCCL entry points, the all-gather as stream-ordered device copies with a
cross-stream event rendez-vous).  What runs here and nowhere else on one GPU: `ncclCommInitRank` with nranks = 2 on an id
created by rank 0, slices that differ per rank landing at their rank offset of the table, the system-scope `ready` events,
and the GPU-side wait of the step on its table that a one-rank exchange drops.  No process is forked or exec'ed.
The robots of a rank are independent of every other rank's (reference rmp.py:133-155); the ranks only share the sphere
table, which each of them must see COMPLETE -- q-double-dot of both ranks against the oracle on the FULL table."""
import ctypes
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stub_rccl(hip_lib):
    import __graft_entry__ as g
    return g.build_stub_rccl()


def _tables(g, n):
    tables = [g["spheres"].copy() for _ in range(n)]
    for k, t in enumerate(tables):
        t[:, :2] *= np.float32(1.0 + 0.03 * k)   # pushed outwards: clearance only grows, every table differs from the last
    return tables


@pytest.mark.parametrize("depth", [1, 2])
def test_two_ranks_in_one_process_see_the_full_table(hip_lib, stub_rccl, golden_dir, depth):
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    world, steps, reps = 2, 6, 64
    K = len(g["spheres"])
    per = K // world
    half = g["q"].shape[0] // world
    tables = _tables(g, steps + depth)
    stub = ctypes.CDLL(stub_rccl)
    stub.stub_rccl_allgathers.restype = ctypes.c_uint64
    calls_before = stub.stub_rccl_allgathers()
    uid = NativeObstacleExchange.unique_id(stub_rccl)
    results, errors = {}, {}
    gate = threading.Barrier(world, timeout=60)

    def rank_main(rank):
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                eng = Engine(desc, 0)
                mine = slice(rank * half, (rank + 1) * half)     # this rank's robots: its half of the golden fleet, tiled
                q, qd, goal = (torch.from_numpy(np.tile(g[k][mine], (reps, 1))).to(dev) for k in ("q", "qd", "goal"))
                out = torch.empty_like(q)
                rows = slice(rank * per, (rank + 1) * per)       # this rank's rows of the sphere table
                dev_slices = [torch.from_numpy(np.ascontiguousarray(t[rows])).to(dev) for t in tables]
                bufs = [torch.empty_like(dev_slices[0]) for _ in range(depth + 1)]
                exch = NativeObstacleExchange(per, dev, depth=depth, rank=rank, world=world, uid=uid, rccl_library=stub_rccl)
                assert exch.nranks == world and exch.rank == rank
                for k in range(depth):
                    bufs[k % (depth + 1)].copy_(dev_slices[k])
                    exch.start(bufs[k % (depth + 1)])
                outs, seen = [], []
                for k in range(steps):
                    sl = bufs[(k + depth) % (depth + 1)]
                    sl.copy_(dev_slices[k + depth])      # producer of this rank's slice of table k + depth, on the launch stream
                    exch.step(eng, q, qd, goal, out, next_local=sl)
                    seen.append(exch._table.value)
                    outs.append(out[:half].clone())
                stream.synchronize()
                results[rank] = dict(outs=[o.cpu().numpy() for o in outs], seen=seen, kernel=eng.last_kernel())
                gate.wait()                              # both ranks leave their communicator together
                exch.close()
        except BaseException as exc:  # noqa: BLE001 -- reported by the main thread
            errors[rank] = exc
            try:
                gate.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank thread hangs"
    assert not errors, f"rank errors: {errors}"
    # one stub all-gather per rank and table
    assert stub.stub_rccl_allgathers() - calls_before == world * (steps + depth)
    for rank in range(world):
        res = results[rank]
        mine = slice(rank * half, (rank + 1) * half)
        assert len(set(res["seen"])) == 3, "the three table buffers must rotate (at either depth)"
        for k in range(steps):
            ref = O.step(desc, g["q"][mine], g["qd"][mine], g["goal"][mine], spheres=tables[k])["qdd64"]
            err = np.abs(res["outs"][k] - ref).max(axis=1)
            tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=1))   # WHICH table was read (tables differ by > 1e-3 per step)
            assert (err <= tol).all(), f"rank {rank} step {k}: worst {err.max():.3e}"
            # a table holding only this rank's own rows (the peer's slice missing or stale) must NOT pass
            partial = tables[k].copy()
            other = slice((1 - rank) * per, (2 - rank) * per)
            partial[other] = tables[max(k - 1, 0)][other] if k else partial[other] + np.float32(0.5)
            ref_partial = O.step(desc, g["q"][mine], g["qd"][mine], g["goal"][mine], spheres=partial)["qdd64"]
            assert np.abs(ref_partial - ref).max() > 1e-3, "the peer's rows must matter to this rank's result"


def test_step_with_all_buffers_outstanding_launches_before_it_gathers(hip_lib, golden_dir):
    """Round-3 advisor finding: with depth + 1 gathers outstanding (allowed by rmp2_exchange_start), the gather that
    rmp2_exchange_step issues for `next_local` lands in the buffer THIS step reads; it must be ordered behind this step's
    read, not in front of the launch.  Primes depth + 1 gathers, then steps with next_local every call: step k must still
    see table k (here the real RCCL at world 1)."""
    import torch
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.engine import Engine
    from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
    dev = torch.device("cuda", 0)
    g = np.load(os.path.join(golden_dir, "config3.npz"))
    _, desc = Cf.config3()
    eng = Engine(desc, 0)
    reps = 256   # 16 384 robots: a step long enough for an early gather to tear the table under it
    q, qd, goal = (torch.from_numpy(np.tile(g[k], (reps, 1))).to(dev) for k in ("q", "qd", "goal"))
    out = torch.empty_like(q)
    n = g["q"].shape[0]
    # (buffers in rotation, depth): with three buffers only depth 2 has every buffer outstanding (x->next == b: the gather goes out
    # AFTER the launch); with RMP2_EXCHANGE_BUFFERS=2 -- the documented A/B knob: depth + 1 buffers -- depth 1 takes that branch too
    # (round-4 advisor: the knob's modulo was never run).  The variable is read in rmp2_exchange_create.
    for buffers, depth in (("3", 1), ("3", 2), ("2", 1)):
        steps = 8
        ahead = depth + 1
        tables = _tables(g, steps + ahead)
        dev_tables = [torch.from_numpy(t).to(dev) for t in tables]
        bufs = [torch.empty_like(dev_tables[0]) for _ in range(ahead + 1)]
        saved_env = os.environ.get("RMP2_EXCHANGE_BUFFERS")
        os.environ["RMP2_EXCHANGE_BUFFERS"] = buffers
        try:
            exch = NativeObstacleExchange(len(g["spheres"]), dev, depth=depth)
        finally:
            if saved_env is None:
                os.environ.pop("RMP2_EXCHANGE_BUFFERS", None)
            else:
                os.environ["RMP2_EXCHANGE_BUFFERS"] = saved_env
        seen = []
        for k in range(ahead):
            bufs[k % (ahead + 1)].copy_(dev_tables[k])
            exch.start(bufs[k % (ahead + 1)])
        assert exch.pending == depth + 1
        outs = []
        for k in range(steps):
            sl = bufs[(k + ahead) % (ahead + 1)]
            sl.copy_(dev_tables[k + ahead])
            exch.step(eng, q, qd, goal, out, next_local=sl)
            assert exch.pending == depth + 1
            seen.append(exch._table.value)
            outs.append(torch.stack((out[:n], out[-n:])).clone())   # first and last wave of the grid
        torch.cuda.synchronize(dev)
        # which rotation ran: the tables the steps read cycle through 3 buffers, or through depth + 1 = 2 under the knob
        nbuf = 3 if buffers == "3" else depth + 1
        assert len(set(seen)) == nbuf and all(seen[k] == seen[k % nbuf] for k in range(steps)), (buffers, depth, seen)
        for k in range(steps):
            ref = O.step(desc, g["q"], g["qd"], g["goal"], spheres=tables[k])["qdd64"]
            for part in outs[k].cpu().numpy():
                err = np.abs(part - ref).max(axis=1)
                tol = 1e-4 * np.maximum(1.0, np.abs(ref).max(axis=1))
                assert (err <= tol).all(), f"buffers {buffers} depth {depth} step {k}: worst {err.max():.3e}"
        exch.close()


def test_a_communicator_of_another_size_is_refused(hip_lib, stub_rccl):
    """rmp2_exchange_nranks is ncclCommCount of the communicator that FORMED, and rmp2_exchange_create refuses one whose size or
    rank differs from what it was asked to join with (round-4 advisor: the function used to echo its own argument)."""
    import torch
    from riemannian_motion_policies_amd import _native
    from riemannian_motion_policies_amd.fleet import NativeObstacleExchange
    dev = torch.device("cuda", 0)
    uid = NativeObstacleExchange.unique_id(stub_rccl)
    exch = NativeObstacleExchange(4, dev, rank=0, world=1, uid=uid, rccl_library=stub_rccl)
    assert exch.nranks == 1
    exch.close()
    os.environ["STUB_RCCL_LIE_ABOUT_COUNT"] = "1"       # the stand-in's communicator now reports one rank more than it has
    try:
        with pytest.raises(_native.Rmp2Error, match="formed with 2 rank"):
            NativeObstacleExchange(4, dev, rank=0, world=1, uid=NativeObstacleExchange.unique_id(stub_rccl), rccl_library=stub_rccl)
    finally:
        os.environ.pop("STUB_RCCL_LIE_ABOUT_COUNT", None)
