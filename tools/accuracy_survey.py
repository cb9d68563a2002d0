"""Distribution of the accuracy verdicts (oracle.accuracy_gate) over the unrestricted perf fleets the full-size tests and
bench.py check: how many robots pass the north-star bound (A), how many only the backward bound (B), how many neither."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402
from riemannian_motion_policies_amd.fleet import MixedFleetShard  # noqa: E402


def report(name, got, ref, spread=None):
    g = O.accuracy_gate(got, ref, spread=spread)
    bad = ~g["ok"]
    na = ~g["a"]
    row = dict(name=name, robots=len(got), A=int(g["a"].sum()), B=int(g["b"].sum()), C=int(g["c"].sum()), nan=int(g["both_nan"].sum()), fail=int(bad.sum()),
               omega_max_nonA=float(g["omega"][na].max()) if na.any() else 0.0,
               omega_pcts_nonA=[float(x) for x in np.percentile(g["omega"][na], [50, 90, 99, 100])] if na.any() else [],
               cond_pcts_nonA=[float(x) for x in np.percentile(g["cond"][na], [50, 90, 100])] if na.any() else [],
               worst_fail=[dict(omega=float(g["omega"][i]), cond=float(g["cond"][i]), err=float(g["err_inf"][i]),
                                ref=float(np.abs(ref["qdd64"][i]).max())) for i in np.where(bad)[0][:6]])
    print(json.dumps(row), flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    for rank in (0, 7):
        shard = MixedFleetShard.synthetic(262144, 8, rank, 0)
        shard.step()
        torch.cuda.synchronize()
        for key, part in shard.parts.items():
            q, qd, goal, _ = part["keep"]
            h = part["host"]
            m = min(n, part["n"])
            off = h["csr_offset"][: m + 1]
            ref = O.step(part["desc"], q[:m].cpu().numpy(), qd[:m].cpu().numpy(), goal[:m].cpu().numpy(),
                         spheres=h["spheres"], csr_offset=off, csr_index=h["csr_index"][: off[-1]])
            report(f"config5 rank {rank} {key}", part["out"][:m].cpu().numpy(), ref)
        del shard
    for solve in ("auto", "pinv"):
        _, desc = Cf.config3(solve)
        eng = Engine(desc, 0)
        s = Cf.sample_panda_states(np.random.default_rng(1), 65536)
        sph = Cf.sample_spheres(np.random.default_rng(7), Cf.N_SPHERES)
        out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]),
                       obstacles=eng.obstacles(spheres=torch.from_numpy(sph)))
        torch.cuda.synchronize()
        ref = O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=sph)
        report(f"config3 perf inputs solve={solve} ({eng.last_kernel()[:40]})", out[:n].cpu().numpy(), ref)
        if solve == "auto":
            caps = Cf.sample_capsules(np.random.default_rng(7), Cf.N_SPHERES)
            out = eng.step(torch.from_numpy(s["q"]), torch.from_numpy(s["qd"]), torch.from_numpy(s["goal"]),
                           obstacles=eng.obstacles(spheres=torch.from_numpy(caps)))
            torch.cuda.synchronize()
            sp = O.fp32_resolution(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=caps)
            report("config3c capsules", out[:n].cpu().numpy(), O.step(desc, s["q"][:n], s["qd"][:n], s["goal"][:n], spheres=caps), sp)


if __name__ == "__main__":
    main()
