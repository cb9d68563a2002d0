"""Diagnostic: when do the waves of one config-3 launch start and end?  Uses the -DRMP2_STAMPS build
(tools/diag/librmp2_stamps.so): every wave writes its shader-clock stamps; prints the launch's span, the
distribution of wave lifetimes and of start / end times, and the tick rate against a HIP-event timing."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RMP2_LIB", os.path.join(ROOT, "tools", "diag", "librmp2_stamps.so"))
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
desc = Cf.config3()[1]
s = Cf.sample_panda_states(np.random.default_rng(1), R)
if os.environ.get("UNIFORM") == "1":  # every robot the same state: what is left of the spread is placement, not data
    s = {k: np.ascontiguousarray(np.broadcast_to(v[:1], v.shape)) for k, v in s.items()}
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7))).cuda()
eng = Engine(desc, 0)
obs = eng.obstacles(spheres=sph)
buf = torch.zeros(R * 9 + 16 * ((R + 15) // 16), dtype=torch.float64, device="cuda")  # f rows, then 16 stamps per wave
f = buf[: R * 9].view(R, 9)
for _ in range(200):
    eng.step(q, qd, goal, obstacles=obs, f=f)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    eng.step(q, qd, goal, obstacles=obs, f=f)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
st = buf.cpu().numpy().view(np.uint64)[R * 9 : R * 9 + ((R + 15) // 16) * 16].reshape(-1, 16).astype(np.int64)
t0, t1 = st[:, 0], st[:, 6]
span = t1.max() - t0.min()
life = t1 - t0
pc = lambda a, p: np.percentile(a, p)
print(f"{eng.last_kernel()}  RMP2_QUAD_MINW={os.environ.get('RMP2_QUAD_MINW', '2')}  R={R}")
print(f"step {us:.2f} us (events, stamped build); span {span} ticks -> {span / us / 1e3:.3f} ticks/ns")
print(f"wave lifetime ticks: p5 {pc(life, 5):.0f} median {pc(life, 50):.0f} p95 {pc(life, 95):.0f} max {life.max()}")
rs, re_ = (t0 - t0.min()) / span, (t1 - t0.min()) / span
print("start time (fraction of span): " + " ".join(f"p{p} {pc(rs, p):.3f}" for p in (5, 25, 50, 75, 95, 100)))
print("end   time (fraction of span): " + " ".join(f"p{p} {pc(re_, p):.3f}" for p in (0, 5, 25, 50, 75, 95, 100)))
h, _ = np.histogram(rs, bins=10, range=(0, 1))
print("starts per tenth of the span:", h.tolist())
h, _ = np.histogram(re_, bins=10, range=(0, 1))
print("ends   per tenth of the span:", h.tolist())

# s_memtime is per XCD (the counters are not aligned across XCDs): look inside clusters of nearby start stamps
order = np.argsort(t0)
ts, te = t0[order], t1[order]
cuts = np.where(np.diff(ts) > 1_000_000)[0] + 1
for c0, c1 in zip(np.r_[0, cuts], np.r_[cuts, len(ts)]):
    a, b = ts[c0:c1], te[c0:c1]
    if len(a) < 8:
        continue
    off = a - a.min()
    late = off > 10_000
    print(f"cluster of {len(a):4d} waves: span {b.max() - a.min():7d} ticks; started late (> 10 k ticks after the first): "
          f"{late.sum():3d}, their start offsets p50 {np.median(off[late]) if late.any() else 0:.0f}, lifetimes p50 "
          f"{np.median((b - a)[late]) if late.any() else 0:.0f}; on-time lifetimes p50 {np.median((b - a)[~late]):.0f} max {(b - a)[~late].max()}")
