"""One-off stress: default dispatch at awkward fleet sizes (block tails, several blocks per CU) vs the oracle on a subset."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
worst = 0.0
for name in ("config2", "config3"):
    _, desc = getattr(Cf, name)()
    eng = Engine(desc, 0)
    for R in (1, 2, 5, 15, 17, 4097, 12289, 20480, 20481, 30001):
        rng = np.random.default_rng(R)
        s = Cf.sample_panda_states(rng, R)
        sph = Cf.sample_spheres(rng)
        sph[:, 2] += 1.2
        obs = eng.obstacles(spheres=torch.from_numpy(sph)) if name == "config3" else None
        st = torch.zeros(R, dtype=torch.int32, device="cuda")
        out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), obstacles=obs, status=st)
        torch.cuda.synchronize()
        sub = np.unique(np.concatenate([np.arange(min(R, 40)), np.arange(max(0, R - 40), R), rng.integers(0, R, 300)]))
        kw = dict(spheres=sph) if name == "config3" else {}
        ref = O.step(desc, s["q"][sub], s["qd"][sub], s["goal"][sub], **kw)
        got = out.cpu().numpy()[sub]
        err = np.abs(got - ref["qdd64"]).max(axis=1) / np.maximum(1.0, np.abs(ref["qdd64"]).max(axis=1))
        fin = np.isfinite(ref["qdd64"]).all(axis=1)
        worst = max(worst, float(err[fin].max()))
        flag = "" if err[fin].max() < 2e-5 else "  <-- OUT OF TOLERANCE"
        print(f"{name} R={R}: worst scaled error {err[fin].max():.2e}, status!=0: {int((st != 0).sum())}{flag}")
print("worst", worst)
