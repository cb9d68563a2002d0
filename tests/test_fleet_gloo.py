"""N > 1 path on CPU: two gloo ranks shard the fleet, all-gather the distributed sphere table
and step their shard; the concatenated result must equal the single-process result.  The
control step itself is played by the CPU oracle here (test infrastructure): what is under
test is the sharding + obstacle exchange of riemannian_motion_policies_amd/fleet.py."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    from riemannian_motion_policies_amd.fleet import ObstacleExchange, shard_bounds
    R, K = 50, 32
    s = Cf.sample_panda_states(np.random.default_rng(11), R)
    spheres = Cf.sample_spheres(np.random.default_rng(12), K)
    _, desc = Cf.config3()
    start, count = shard_bounds(R, world, rank)
    ex = ObstacleExchange(K // world, "cpu")
    ex.start(torch.from_numpy(spheres[rank * (K // world):(rank + 1) * (K // world)]))
    table = ex.finish().numpy().copy()           # (a view of buffer 0, which the pipelined gathers below reuse)
    assert np.array_equal(table, spheres), "all-gathered sphere table differs from the global table"
    # pipelined use: two gathers in flight (double buffering), consumed in issue order
    per = K // world
    for shift in (1.0, 2.0):
        ex.start(torch.from_numpy(spheres[rank * per:(rank + 1) * per] + np.float32(shift)))
    for shift in (1.0, 2.0):
        assert np.array_equal(ex.finish().numpy(), spheres + np.float32(shift))
        ex.consumed()
    # a one-rank exchange INSIDE the two-rank job (bench.py's world-1 leg): the slice is the table, no collective of the job is
    # issued -- rank 1 makes an extra exchange step that rank 0 does not, which a gather over the default group would hang on
    solo = ObstacleExchange(K, "cpu", collective=False)
    assert solo.world == 1 and solo.rank == 0 and solo.tables[0].shape == (K, 4)
    for _ in range(1 + rank):
        solo.start(torch.from_numpy(spheres + np.float32(rank)))
        assert np.array_equal(solo.finish().numpy(), spheres + np.float32(rank))
        solo.consumed()
    sl = slice(start, start + count)
    r = O.step(desc, s["q"][sl], s["qd"][sl], s["goal"][sl], spheres=table)
    np.save(os.path.join(tmp, f"qdd_{rank}.npy"), r["qdd"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_step_equals_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    from riemannian_motion_policies_amd import configs as Cf
    O.build()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"qdd_{r}.npy") for r in range(2)])
    s = Cf.sample_panda_states(np.random.default_rng(11), 50)
    _, desc = Cf.config3()
    want = O.step(desc, s["q"], s["qd"], s["goal"], spheres=Cf.sample_spheres(np.random.default_rng(12), 32))["qdd"]
    assert np.array_equal(got, want)
