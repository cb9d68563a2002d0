"""Which of the two strict resolves is off: certified elimination vs all-Jacobi, against numpy on the exported (M, f)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riemannian_motion_policies_amd import configs as Cf, descriptor as D
from riemannian_motion_policies_amd.engine import Engine
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = Cf.sample_panda_states(np.random.default_rng(1), R)
sph = torch.from_numpy(Cf.sample_spheres(np.random.default_rng(7)))
q, qd, goal = (torch.from_numpy(s[k]).cuda() for k in ("q", "qd", "goal"))
_, desc = Cf.config3("pinv")
eng = Engine(desc, 0)
os.environ["RMP2_STRICT_CERTIFY"] = "0"
jac = Engine(desc, 0)
del os.environ["RMP2_STRICT_CERTIFY"]
st = torch.zeros(R, dtype=torch.int32, device="cuda")
M = torch.zeros(R, 9, 9, dtype=torch.float64, device="cuda"); f = torch.zeros(R, 9, dtype=torch.float64, device="cuda")
M2 = torch.zeros_like(M); f2 = torch.zeros_like(f)
g = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=sph), status=st).clone()
g_dbg = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=sph), M=M, f=f).clone()
w = jac.step(q, qd, goal, obstacles=jac.obstacles(spheres=sph)).clone()
w_dbg = jac.step(q, qd, goal, obstacles=jac.obstacles(spheres=sph), M=M2, f=f2).clone()
torch.cuda.synchronize()
print("kernels:", eng.last_kernel(), "|", jac.last_kernel())
print("M equal:", torch.equal(M, M2), "f equal:", torch.equal(f, f2), "g==g_dbg", torch.equal(g, g_dbg), "w==w_dbg", torch.equal(w, w_dbg))
g, w, Mn, fn = g.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), M.cpu().numpy(), f.cpu().numpy()
rel = np.abs(g - w).max(axis=1) / np.abs(w).max(axis=1)
for i in np.argsort(rel)[-4:]:
    x = np.linalg.solve(Mn[i], fn[i])
    print(f"robot {i}: rel diff {rel[i]:.2e} |qdd| {np.abs(w[i]).max():.3e} cond {np.linalg.cond(Mn[i]):.2e} status {int(st[i])}"
          f" | certified vs numpy {np.abs(g[i]-x).max()/np.abs(x).max():.2e} | jacobi vs numpy {np.abs(w[i]-x).max()/np.abs(x).max():.2e}"
          f" | eig min {np.linalg.eigvalsh(0.5*(Mn[i]+Mn[i].T))[0]:.3e} asym {np.abs(Mn[i]-Mn[i].T).max():.1e}")
