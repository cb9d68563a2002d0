"""What the stream-ordering pieces of the obstacle exchange cost the step period (65 536 robots, config 3):
  A  plain bound launches back to back
  B  every launch carries a completion fence (hipExtLaunchKernelGGL stop event) nobody waits on
  C  B + a side stream that waits on the fence and copies a 512-byte table (the world-1 gather)
  D  C + the launch stream waits on the copy's fence (the full exchange pattern)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.fleet import _Fence
dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
_, desc = Cf.config3()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
out = torch.empty_like(q)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).to(dev)
tabs = [sph.clone(), sph.clone()]
def timed(fn, n=1000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, (t1 - t0) / n * 1e6
plain = [eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out)[0] for t in tabs]
k = [0]
def A():
    plain[k[0] & 1](); k[0] += 1
print("A plain launches:                         %.1f us per step (host %.1f)" % timed(A))
fences = [_Fence(dev), _Fence(dev)]
fenced = [eng.bind(q, qd, goal, obstacles=eng.obstacles(spheres=t), out=out, done_fence=f)[0] for t, f in zip(tabs, fences)]
def B():
    fenced[k[0] & 1](); k[0] += 1
print("B completion fence on every launch:       %.1f us per step (host %.1f)" % timed(B))
side = torch.cuda.Stream(dev, priority=-1)
ready = [_Fence(dev), _Fence(dev)]
cur = torch.cuda.current_stream(dev)
def C():
    b = k[0] & 1
    fences[b ^ 1].wait(side)
    with torch.cuda.stream(side):
        tabs[b ^ 1].copy_(sph, non_blocking=True)
    ready[b ^ 1].record(side)
    fenced[b](); k[0] += 1
for f in fences: f.record(cur)
print("C + side stream copies the other table:   %.1f us per step (host %.1f)" % timed(C))
def D():
    b = k[0] & 1
    ready[b].wait(cur)
    fences[b ^ 1].wait(side)
    with torch.cuda.stream(side):
        tabs[b ^ 1].copy_(sph, non_blocking=True)
    ready[b ^ 1].record(side)
    fenced[b](); k[0] += 1
for f in ready: f.record(side)
print("D + launch stream waits for its table:    %.1f us per step (host %.1f)" % timed(D))
