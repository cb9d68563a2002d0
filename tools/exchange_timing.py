"""Host-side cost of one config-4 step, piece by piece (python tools/exchange_timing.py): the RCCL all-gather of the
sphere table on the side stream, the event waits, and the launch -- enqueue times (no GPU sync inside the loops), then
the steady-state step time of three ways to drive the loop."""
import os, socket, sys, time
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.fleet import ObstacleExchange

dev = torch.device("cuda", 0)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
_, desc = Cf.config3()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
out = torch.empty_like(q)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).to(dev)
exch = ObstacleExchange(32, dev)
ev = torch.cuda.Event(); ev.record()
N = 500

def timeit(fn, n=N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6

side = torch.cuda.Stream(dev)
tbl = torch.empty_like(sph)
def gather_only():
    with torch.cuda.stream(side):
        dist.all_gather_into_tensor(tbl, sph)
print("all_gather_into_tensor on a side stream: host %.1f us / call, wall %.1f us / call" % timeit(gather_only))
obs = eng.obstacles(spheres=sph)
launch, _ = eng.bind(q, qd, goal, obstacles=obs, out=out)
print("bound launch:                            host %.1f us / call, wall %.1f us / call" % timeit(launch))
print("eng.obstacles + eng.step:                host %.1f us / call, wall %.1f us / call" % timeit(lambda: eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=sph), out=out)))
exch.start(sph, produced=ev)
def bench_style():
    t = exch.finish()
    eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out)
    exch.consumed()
    exch.start(sph, produced=ev)
print("round-1 loop (finish/obstacles+step/consumed/start): host %.1f us, wall %.1f us per step" % timeit(bench_style))
tbl_last = exch.finish()
bound = {t.data_ptr(): eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out)[0] for t in exch.tables}
exch.start(sph, produced=ev)
def bound_style():
    bound[exch.finish().data_ptr()]()
    exch.consumed()
    exch.start(sph, produced=ev)
print("bench.py config4 loop (finish / bound launch / consumed / start): host %.1f us, wall %.1f us per step" % timeit(bound_style, 2000))
exch.finish()
# the same without the reader-done event (diagnostic only: the buffer a gather overwrites was last read two kernels ago)
def no_consumed():
    bound[exch.finish().data_ptr()]()
    exch.start(sph, produced=ev)
exch.start(sph, produced=ev)
print("  ... without consumed():                                         host %.1f us, wall %.1f us per step" % timeit(no_consumed, 2000))
exch.finish()
# the reader-done fence carried by the launch itself (Engine.bind(done_fence=), rmp2_set_step_fence)
boundf = {t.data_ptr(): eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out, done_fence=exch.reader_fence(t))[0]
          for t in exch.tables}
def fenced_style():
    boundf[exch.finish().data_ptr()]()
    exch.consumed(attached=True)
    exch.start(sph, produced=ev)
exch.start(sph, produced=ev)
print("  ... reader fence attached to the launch:                        host %.1f us, wall %.1f us per step" % timeit(fenced_style, 2000))
exch.finish()
dist.destroy_process_group()
