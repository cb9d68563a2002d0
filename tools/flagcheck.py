import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from riemannian_motion_policies_amd import configs as Cf
from riemannian_motion_policies_amd.engine import Engine
for kern in ("hex", "quad"):
    os.environ["RMP2_KERNEL"] = kern
    _, desc = Cf.config2()
    eng = Engine(desc, 0)
    s = Cf.sample_panda_states(np.random.default_rng(1), 4096)
    st = torch.zeros(4096, dtype=torch.int32, device="cuda")
    out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]), status=st)
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    print(kern, "status nonzero:", int((st != 0).sum()), "pinv path:", int(((st & 4) != 0).sum()), "nonfinite:", int(((st & 1) != 0).sum()))
