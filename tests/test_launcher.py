"""Host logic of the N-rank launcher, on CPU: bench.py's rank supervisor (a failing or hanging rank ends the job, non-zero),
the ranks' agreement on which obstacle exchange they all take (fleet.agree_on_exchange under 2-rank gloo with the native
constructor failing on ONE rank), and the stand-in collective library of tests/test_gpu_exchange_ranks.py (loads, exports the
five RCCL entry points the exchange binds -- no compute calls without a GPU)."""
import ctypes
import os
import socket
import subprocess
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _child(code: str):
    return subprocess.Popen([sys.executable, "-c", code], start_new_session=True)


def test_supervisor_all_ranks_ok():
    import bench
    procs = [_child("import time; time.sleep(0.2)") for _ in range(3)]
    assert bench.supervise(procs, timeout_s=30) == 0


def test_supervisor_ends_the_siblings_of_a_failed_rank():
    import bench
    procs = [_child("import time; time.sleep(60)"), _child("import sys; sys.exit(3)"), _child("import time; time.sleep(60)")]
    t0 = time.monotonic()
    rc = bench.supervise(procs, timeout_s=50, grace_s=2)
    assert rc == 3 and time.monotonic() - t0 < 15, "a dead rank must end the job at once, with its exit code"
    assert all(p.poll() is not None for p in procs), "no rank may outlive the job"


def test_supervisor_times_out_on_a_hung_rank():
    import bench
    # one rank ignores SIGTERM (a process stuck in a collective often does): it is killed after the grace period
    procs = [_child("import time; time.sleep(0.1)"),
             _child("import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(60)")]
    t0 = time.monotonic()
    rc = bench.supervise(procs, timeout_s=1.0, grace_s=1.0)
    assert rc == 124 and time.monotonic() - t0 < 15
    assert all(p.poll() is not None for p in procs)


class _FakeExchange:
    closed = False

    def close(self):
        self.closed = True


def _agree_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from riemannian_motion_policies_amd.fleet import agree_on_exchange
    built = _FakeExchange()

    def make_fails_on_rank_1():
        if rank == 1:
            raise RuntimeError("dlopen(librccl.so): not found")   # what a rank without the library would raise
        return built

    exch, err = agree_on_exchange(make_fails_on_rank_1, world)
    # every rank falls back TOGETHER; the rank that did build its exchange has closed it
    assert exch is None and err is not None
    assert built.closed == (rank == 0)
    assert ("dlopen" in str(err)) == (rank == 1)
    good, err2 = agree_on_exchange(lambda: built, world)
    assert good is built and err2 is None
    dist.barrier()      # (the collectives after the agreement still match up on both ranks)
    open(os.path.join(tmp, f"ok_{rank}"), "w").close()
    dist.destroy_process_group()


def test_ranks_agree_on_the_exchange_when_one_cannot_build_it(tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_agree_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok_{r}").exists() for r in range(2))


def test_stub_collective_library_exports_what_the_exchange_binds():
    import __graft_entry__ as g
    lib = ctypes.CDLL(g.build_stub_rccl())
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString",
                "stub_rccl_allgathers"):
        assert hasattr(lib, sym), sym
    uid = (ctypes.c_char * 128)()
    assert lib.ncclGetUniqueId(uid) == 0 and bytes(uid).startswith(b"stub-rccl")
    uid2 = (ctypes.c_char * 128)()
    lib.ncclGetUniqueId(uid2)
    assert bytes(uid) != bytes(uid2)


def test_committed_bench_line_keeps_the_contract():
    """The line `python bench.py` printed on the GPU box this round (profiles/r05_bench_default.json): every field the driver's
    contract names, the roofline and cpu_baseline objects, the tolerance rule + robots admitted per clause in result_check, and the
    round-5 additions: `value` on solve = "pinv" with the auto leg nested, the executed-flop fraction beside the algorithmic one."""
    import json
    line = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(line["value"] - line["config"]["robots_per_gpu"] * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    cpu = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference")
    chk = line["result_check"]
    assert chk["rejected"] == 0 and sum(chk["admitted_by"].values()) == chk["robots_checked"] and "north star" in chk["tolerance"]
    assert set(chk["passing_each_clause_on_its_own"]) == {"A", "B", "E"}
    assert line["config"]["solve"] == "pinv" and "solve_auto" in line and line["solve_auto"]["ms_per_step"] > 0
    assert line["value"] >= 1.43e9                                            # (the round-4 review's bar for the pinv headline)
    assert 0 < roof["executed_frac"] < roof["frac"] and roof["executed_flops_per_robot_step"] < 66.0e3
    assert len(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()) == 1   # ONE line on stdout


def test_spawned_ranks_share_device_0_in_a_rehearsal(monkeypatch):
    """bench.py --gpus N --rehearse-one-gpu: spawn_ranks needs ONE device, gives every child LOCAL_RANK 0 and its own RANK, and
    still refuses N ranks on real devices the node does not have."""
    import types
    import bench
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            started.append(env)

        def poll(self):
            return 0

    import __graft_entry__ as ge
    monkeypatch.setattr(ge, "build_hip", lambda: None)
    monkeypatch.setattr(bench, "count_gpus_without_hip", lambda: 1)
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    args = types.SimpleNamespace(gpus=2, rehearse_one_gpu=False, rank_timeout=5.0)
    assert bench.spawn_ranks(args) == 2 and not started          # two real devices asked for, one present: refused, nothing started
    args.rehearse_one_gpu = True
    assert bench.spawn_ranks(args) == 0
    assert [e["RANK"] for e in started] == ["0", "1"] and {e["LOCAL_RANK"] for e in started} == {"0"}
    assert {e["WORLD_SIZE"] for e in started} == {"2"} and len({e["MASTER_PORT"] for e in started}) == 1


def test_committed_two_rank_rehearsal_lines():
    """profiles/r05_rehearsal_2ranks_*.json: what `tools/refresh_profiles_r05.sh` printed on the one-GPU box -- two rank
    PROCESSES through bench.py's own launcher and through torch.distributed.run (the driver's command shape).  Config 4: the
    native exchange's communicator failed on both ranks (RCCL refuses two ranks on one device), the ranks agreed and every rank
    took the torch-driven exchange; both configs: every checked robot passed the gate; each line says it is a rehearsal."""
    import json
    for name in ("config4", "config4_torchrun", "config5"):
        line = json.loads(open(os.path.join(ROOT, "profiles", f"r05_rehearsal_2ranks_{name}.json")).read().strip().splitlines()[-1])
        assert line["rehearsal"]["ranks"] == 2 and line["rehearsal"]["gpus"] == 1 and line["n_gpus"] == 1
        assert "NOT a scaling measurement" in line["rehearsal"]["note"]
        total = line["config"]["robots_per_gpu"] * 2
        assert abs(line["value"] - total * 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]      # all ranks' robots over the MAX time
        chk = line["result_check"]
        if name == "config5":
            assert chk["two_joint"]["rejected"] == 0 and chk["panda"]["rejected"] == 0 and len(line["shards"]) == 2
        else:
            assert chk["rejected"] == 0
            assert line["exchange"] == "torch" and "every rank took --exchange torch" in line["exchange_fell_back"]
            assert line["rccl_nranks"] is None
            # the world-1 leg of the same workload: every rank's own shard, no collective of the job inside it
            assert 0 < line["world1_same_workload_ms"] < line["ms_per_step"] and "MAX over ranks" in line["world1_same_workload"]["what"]
