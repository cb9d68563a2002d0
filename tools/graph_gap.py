"""Back-to-back rmp2_step launches: plain stream launches vs one hipGraph holding K kernel nodes.
usage: python tools/graph_gap.py [config2|config3] [R] [K]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
_, desc = getattr(Cf, wl)()
eng = Engine(desc, 0)
s = Cf.sample_panda_states(np.random.default_rng(1), R)
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
out = torch.empty_like(q)
obs = eng.obstacles(spheres=torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()) if wl == "config3" else None
side = torch.cuda.Stream()
launch_main, _ = eng.bind(q, qd, goal, obstacles=obs, out=out)
launch_side, _ = eng.bind(q, qd, goal, obstacles=obs, out=out, stream=side.cuda_stream)
for _ in range(20):
    launch_main()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(K):
        launch_main()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / K * 1e6)
print(f"{wl} R={R}: stream launches  {np.median(ts):7.2f} us/step")
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    launch_side()
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(K):
            launch_side()
g.replay()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / K * 1e6)
print(f"{wl} R={R}: graph of {K} nodes {np.median(ts):7.2f} us/step")
