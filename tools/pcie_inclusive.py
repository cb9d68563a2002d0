"""Secondary figure of SURVEY 8(d): the control step INCLUDING the host round trip of q, qd (H2D) and qdd (D2H) through
pinned buffers, one fleet, one stream.  python tools/pcie_inclusive.py [config2|config3] [R] [pinv|auto]  (pinv, the reference's only resolve and the bench line's, by default)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
solve = sys.argv[3] if len(sys.argv) > 3 else "pinv"
_, desc = getattr(Cf, wl)(solve)
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
hq, hqd = (torch.from_numpy(s[k]).pin_memory() for k in ("q", "qd"))
hout = torch.empty((R, 9), dtype=torch.float32).pin_memory()
q, qd, out = torch.empty_like(hq, device="cuda"), torch.empty_like(hqd, device="cuda"), torch.empty((R, 9), device="cuda")
goal = torch.from_numpy(s["goal"]).cuda()
obs = eng.obstacles(spheres=torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()) if wl == "config3" else None
launch, _ = eng.bind(q, qd, goal, obstacles=obs, out=out)
def step():
    q.copy_(hq, non_blocking=True)
    qd.copy_(hqd, non_blocking=True)
    launch()
    hout.copy_(out, non_blocking=True)
for _ in range(50):
    step()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(500):
        step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 500)
t = float(np.median(ts))
print(f"{wl} R={R} solve={solve}: {t * 1e6:.1f} us per step with H2D(q, qd) + D2H(qdd) on one stream -> {R / t / 1e6:.0f} M steps/s "
      f"({120 * R / t / 1e9:.2f} GB/s over the host link incl. goal-less 108 B/robot)")
