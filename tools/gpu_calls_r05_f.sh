#!/bin/bash
# interface B, streamed form, what bounds it: the stream without the pair arithmetic, the arithmetic without the stream (results are
# then wrong by construction: the result check is off for these two)
O=gpurun_out/r05; mkdir -p $O
python - > $O/interface_b_floor.txt 2>&1 <<'PY'
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
R = 65536
s = Cf.sample_panda_states(np.random.default_rng(1), R)
sph = Cf.sample_spheres(np.random.default_rng(7), Cf.N_SPHERES)
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
print("# config3b, 65 536 robots, solve = pinv: us per step of the streamed explicit-pair step and of its two halves")
for label, env in (("single-loop two-wave form", dict(RMP2_EXPLICIT_STREAM="0")), ("streamed", dict(RMP2_EXPLICIT_STREAM="1", RMP2_STREAM_STAGGER="0")),
                   ("streamed, no pair in range (stream + everything but the pair arithmetic)", dict(RMP2_EXPLICIT_STREAM="1", RMP2_STREAM_STAGGER="256")),
                   ("streamed, no DMA issued (everything but the stream)", dict(RMP2_EXPLICIT_STREAM="1", RMP2_STREAM_STAGGER="512")),
                   ("streamed, neither", dict(RMP2_EXPLICIT_STREAM="1", RMP2_STREAM_STAGGER="768"))):
    os.environ.update(env)
    _, desc = Cf.config3("pinv")
    eng = Engine(desc, 0)
    for k in env: os.environ.pop(k)
    pl, po = eng.closest_points(q, eng.obstacles(spheres=torch.from_numpy(sph)))
    out = torch.empty_like(q)
    launch, _ = eng.bind(q, qd, goal, obstacles=eng.obstacles(p_link=pl, p_obs=po), out=out)
    for _ in range(50): launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): launch()
    torch.cuda.synchronize()
    print(f"{label:80s} {(time.perf_counter() - t0) / 400 * 1e6:7.2f} us   [{eng.last_kernel()[:70]}]")
PY
cat $O/interface_b_floor.txt | cut -c1-200
