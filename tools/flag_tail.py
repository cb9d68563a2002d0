"""Diagnostic: how many robots of a bench shard leave the fast resolve (status bits), and what the step then costs.
The control-step kernel's time is the time of its SLOWEST wave: one robot on the careful path sets it.
usage: flag_tail.py [robots]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
from riemannian_motion_policies_amd.fleet import MixedFleetShard
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)

def timed(launch, n=200):
    for _ in range(20): launch()
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()   # (wall clock: a bound launch may sit on a side stream)
    for _ in range(n): launch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

_, desc = Cf.config3()
eng = Engine(desc, 0)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).to(dev)
obs = eng.obstacles(spheres=sph)
for r in range(8):
    s = Cf.sample_panda_states(np.random.default_rng(1 + 1000 * r), R)
    q, qd, goal = (torch.from_numpy(s[k]).to(dev) for k in ("q", "qd", "goal"))
    st = torch.zeros(R, dtype=torch.int32, device=dev)
    out = eng.step(q, qd, goal, obstacles=obs, status=st)
    torch.cuda.synchronize()
    stc = st.cpu().numpy()
    launch, _ = eng.bind(q, qd, goal, obstacles=obs)
    t = timed(launch)
    bad = np.nonzero(stc)[0]
    print(f"config3 seed-rank {r}: {t:7.1f} us  careful-path robots {int((stc & 2).astype(bool).sum() + 0)} pinv {int((stc & 2 > 0).sum())} "
          f"status histogram {dict(zip(*np.unique(stc, return_counts=True)))}  |qdd|max of flagged "
          f"{[float(out[i].abs().max()) for i in bad[:4]]}")
for r in range(4):
    shard = MixedFleetShard.synthetic(32768 * 8, 8, r, 0)
    for key, p in shard.parts.items():
        q, qd, goal, ob = p["keep"]
        st = torch.zeros(p["n"], dtype=torch.int32, device=dev)
        p["engine"].step(q, qd, goal, obstacles=ob, status=st)
        torch.cuda.synchronize()
        stc = st.cpu().numpy()
        t = timed(p["launch"])
        print(f"config5 rank {r} {key} n={p['n']}: {t:7.1f} us  status histogram {dict(zip(*np.unique(stc, return_counts=True)))} {p['engine'].last_kernel()}")
