"""Accuracy survey of the BASELINE perf fleets (round 5; profiles/r05_accuracy_survey.txt): for the first 2 048 robots of every
perf fleet -- the fleets of tests/golden/make_perf_envelope.py: configs 2, 3, 3b (config 3 through explicit pairs), 3c, 5 (both robot
types) -- the engine's error against the fp64 evaluation next to the errors of the two fp32 evaluations of the reference's graph
that exist here:

    err_engine   |engine - fp64|_inf          (fp64: the C oracle's double build on the interface the engine was fed)
    err_c32      |C oracle, fp32 leaves - fp64|_inf on the same interface
    err_ref32    |torch-autograd fp32 restatement - fp64 of the same explicit pairs|_inf   (tests/golden/perf_envelope.npz)

and EVERY clause of oracle.accuracy_gate counted on its own (non-exclusively): A north star, B backward error (with the omega
distribution, so that eta can be read off the data), C fp32 resolution of the inputs, D fp32 resolution of the system.  The ratio
column is err_engine / max(err_ref32, err_c32) over the robots outside A: what separates error the KERNEL adds from error any fp32
evaluation of the algorithm has.  Raw per-robot arrays go to <out>.npz for offline analysis.

    python tools/accuracy_survey.py [n] [out_prefix]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle as O  # noqa: E402
import make_perf_envelope as E  # noqa: E402
from riemannian_motion_policies_amd import configs as Cf  # noqa: E402
from riemannian_motion_policies_amd.engine import Engine  # noqa: E402
from riemannian_motion_policies_amd.fleet import MixedFleetShard  # noqa: E402


def pct(a, qs=(50, 90, 99, 100)):
    a = np.asarray(a)
    a = a[np.isfinite(a)]
    return [float(f"{x:.3g}") for x in np.percentile(a, qs)] if a.size else []


def envelope_without_the_plain_draw(desc, q, qd, goal, kw, base, samples=16, seed=1):
    """oracle.fp32_envelope's perturbed draws only (another seed): the envelope the PLAIN fp32 evaluation is held against in the
    control column -- built without it, as the engine's envelope is built without the engine."""
    rng = np.random.default_rng(seed)
    eps = np.float64(2.0 ** -23)

    def jiggle(a):
        a = np.ascontiguousarray(a, dtype=np.float32).astype(np.float64)
        return (a + rng.choice(np.array([-1.0, 1.0]), a.shape) * eps * np.maximum(np.abs(a), 1.0)).astype(np.float32)
    env = np.zeros(len(q))
    with np.errstate(invalid="ignore"):
        for _ in range(samples):
            kk = {k: (jiggle(v) if k in ("spheres", "p_link", "p_obs", "dist") else v) for k, v in kw.items()}
            r = O.step(desc, jiggle(q), jiggle(qd), jiggle(goal), precision="f32", **kk)["qdd64"]
            env = np.fmax(env, np.abs(r - base).max(axis=1))
    return env


def survey_row(name, got, kw, fleet, desc, err_ref32, raw):
    """One line of the survey + the raw arrays.  kw: the oracle's obstacle arguments of the interface the engine was fed."""
    q, qd, goal = fleet["q"], fleet["qd"], fleet["goal"]
    n = len(q)
    r64 = O.step(desc, q, qd, goal, precision="f64", **kw)
    c32 = O.step(desc, q, qd, goal, precision="f32", **kw)
    spread = O.ulp_spread(desc, q, qd, goal, **kw)
    sysres = O.system_resolution(c32)
    truth = r64["qdd64"]
    scale = np.maximum(1.0, np.abs(truth).max(axis=1))
    with np.errstate(invalid="ignore"):
        e_eng = np.abs(got.astype(np.float64) - truth).max(axis=1)
        e_c32 = np.abs(c32["qdd64"] - truth).max(axis=1)
    e_ref = np.zeros(n) if err_ref32 is None else np.asarray(err_ref32)[:n]
    env = np.maximum(e_ref, e_c32)
    # the fp32 envelope the gate uses (clause E): 17 fp32 evaluations on inputs moved by an fp32 rounding, and the restatement's own
    # error; `control`: the PLAIN fp32 evaluation against an envelope built from 16 OTHER draws -- what a faithful evaluation's ratio
    # looks like, for the engine's to be read against
    env17 = np.maximum(O.fp32_envelope(desc, q, qd, goal, **kw), e_ref)
    env_ctl = np.maximum(envelope_without_the_plain_draw(desc, q, qd, goal, kw, r64["qdd64"]), e_ref)
    # the gate as the tests and bench.py call it (reference = the reference-precision oracle), every clause on its own
    g = O.accuracy_gate(got, c32, spread=np.fmax(spread, e_c32), system_spread=sysres)
    A = g["a"]
    # accuracy_gate returns b / c / d EXCLUSIVE of the earlier clauses; here each clause is counted on its own (the forward and
    # minimum-norm riders of B only matter for rank-dropping systems, which these fleets do not hold)
    omega, fin = g["omega"], np.isfinite(got).all(axis=1)
    # the backward error against the EXACT system (fp64 evaluation): what clause B is judged on since the kernels form 1 - sigmoid
    # without the cancellation of the reference's fp32 `1. - tf.sigmoid(z)` -- the fp32 oracle's own system carries that noise (its
    # answer's backward error against the exact system is printed beside the engine's)
    omega_x = O.accuracy_gate(got, r64)["omega"]
    omega_c32_x = O.accuracy_gate(c32["qdd64"], r64)["omega"]
    B_incl = fin & (omega <= 1e-4)
    C_incl = fin & (g["err_inf"] <= 8.0 * np.fmax(spread, e_c32))
    D_incl = fin & (g["err_inf"] <= 8.0 * sysres)
    A64 = e_eng <= 1e-5 * scale
    nonA = ~(A | A64)
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(env > 0, e_eng / env, np.inf)
        ratio_c = np.where(e_c32 > 0, e_eng / e_c32, np.inf)
        ratio17 = np.where(env17 > 0, e_eng / env17, np.inf)
        ratio_ctl = np.where(env_ctl > 0, e_c32 / env_ctl, np.inf)
    non_c = e_c32 > 1e-5 * scale
    row = dict(fleet=name, robots=n, A_vs_c32=int(A.sum()), A_vs_f64=int(A64.sum()), B_omega_le_1e4=int(B_incl.sum()),
               B_omega_le_2e5=int((fin & (omega_x <= O.ETA)).sum()), B_vs_fp32_oracle_system=int((fin & (omega <= O.ETA)).sum()),
               omega_exact_max=float(np.nanmax(omega_x)), omega_exact_max_of_the_fp32_oracle=float(np.nanmax(omega_c32_x)), C_x8=int(C_incl.sum()), D_x8=int(D_incl.sum()),
               passes_gate=int(g["ok"].sum()), outside_A=int(nonA.sum()),
               omega_pcts_outside_A=pct(omega[nonA]), omega_max_all=float(np.nanmax(omega)),
               rel_err_engine_pcts_outside_A=pct((e_eng / scale)[nonA]), rel_err_c32_pcts_outside_A=pct((e_c32 / scale)[nonA]),
               rel_err_ref32_pcts_outside_A=pct((e_ref / scale)[nonA]),
               ratio_engine_over_envelope_pcts=pct(ratio[nonA]), ratio_gt_2=int((ratio[nonA] > 2).sum()), ratio_gt_8=int((ratio[nonA] > 8).sum()),
               ratio_engine_over_c32_pcts=pct(ratio_c[nonA]),
               E_ratio_engine_over_envelope17_pcts=pct(ratio17[nonA]), E_ratio_gt_1=int((ratio17[nonA] > 1).sum()),
               E_ratio_gt_2=int((ratio17[nonA] > 2).sum()),
               control_ratio_c32_over_envelope16_pcts=pct(ratio_ctl[non_c]), control_gt_1=int((ratio_ctl[non_c] > 1).sum()),
               control_gt_2=int((ratio_ctl[non_c] > 2).sum()),
               rel_err_all_robots_pcts_engine=pct(e_eng / scale, (50, 90, 99, 99.9)), rel_err_all_robots_pcts_c32=pct(e_c32 / scale, (50, 90, 99, 99.9)),
               c32_outside_A_of_f64=int((e_c32 > 1e-5 * scale).sum()), ref32_outside_A=int((e_ref > 1e-5 * scale).sum()))
    print(json.dumps(row), flush=True)
    for k, v in dict(got=got, truth=truth, c32=c32["qdd64"], e_ref32=e_ref, spread=spread, sysres=sysres, omega=omega, omega_exact=omega_x, cond=g["cond"], env17=env17).items():
        raw[f"{name}__{k}"] = v
    return row


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    out_prefix = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r05", "accuracy_survey")
    env = np.load(os.path.join(ROOT, "tests", "golden", "perf_envelope.npz"))
    fleets = E.perf_fleets(n)
    raw, rows = {}, []
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    for solve in ("pinv", "auto"):
        # ---- config 2 (4 096 robots) ----
        fl = fleets["config2"]
        _, desc = Cf.config2(solve)
        s = Cf.sample_panda_states(np.random.default_rng(1), 4096)
        eng = Engine(desc, 0)
        out = eng.step(t(s["q"]), t(s["qd"]), t(s["goal"]))
        torch.cuda.synchronize()
        rows.append(survey_row(f"config2 solve={solve}", out[:n].cpu().numpy(), {}, fl, desc, env["config2_err_ref32"], raw))
        # ---- config 3 / 3c (65 536 robots, shared table) and 3b (explicit pairs of the first n robots' states over the fleet) ----
        s = Cf.sample_panda_states(np.random.default_rng(1), 65536)
        _, desc = Cf.config3(solve)
        eng = Engine(desc, 0)
        q, qd, goal = t(s["q"]), t(s["qd"]), t(s["goal"])
        for name in ("config3", "config3c"):
            fl = fleets[name]
            out = eng.step(q, qd, goal, obstacles=eng.obstacles(spheres=t(fl["table"])))
            torch.cuda.synchronize()
            rows.append(survey_row(f"{name} solve={solve} [{eng.last_kernel()[:28]}]", out[:n].cpu().numpy(), dict(spheres=fl["table"]), fl, desc,
                                   env[name + "_err_ref32"], raw))
        # interface B: the explicit pairs the restatement read (fp64 kinematics rounded to fp32), tiled over a full-size fleet so that
        # the kernel that runs is the perf fleet's
        fl = fleets["config3"]
        pl, po = E.pair_arrays(fl)
        reps = 65536 // n
        qb, qdb, gb = (t(np.tile(fl[k], (reps, 1))) for k in ("q", "qd", "goal"))
        out = eng.step(qb, qdb, gb, obstacles=eng.obstacles(p_link=t(np.tile(pl, (reps, 1, 1))), p_obs=t(np.tile(po, (reps, 1, 1)))))
        torch.cuda.synchronize()
        rows.append(survey_row(f"config3b solve={solve} [{eng.last_kernel()[:28]}]", out[:n].cpu().numpy(), dict(p_link=pl, p_obs=po), fl, desc,
                               env["config3_err_ref32"], raw))
        del eng
        # ---- config 5: rank 0 (TwoJoint) and rank 7 (Panda) of the 8-rank cut ----
        for rank, key, name in ((0, "two_joint", "config5_two_joint"), (7, "panda", "config5_panda")):
            shard = MixedFleetShard.synthetic(262144, 8, rank, 0, solve=solve, cost=E._fixture_cut())
            shard.step()
            torch.cuda.synchronize()
            part = shard.parts[key]
            fl = fleets[name]
            m = len(fl["q"])
            assert np.array_equal(part["keep"][0][:m].cpu().numpy(), fl["q"])
            rows.append(survey_row(f"{name} solve={solve} [{part['engine'].last_kernel()[:28]}]", part["out"][:m].cpu().numpy(),
                                   E.obstacle_kwargs(fl), fl, part["desc"], env[name + "_err_ref32"], raw))
            del shard
    np.savez_compressed(out_prefix + "_raw.npz", **raw)
    json.dump(rows, open(out_prefix + ".json", "w"), indent=1)
    # ---- the table (profiles/r05_accuracy_survey.txt keeps the JSON rows above AND this) ----
    f3 = lambda v: "/".join(f"{x:.3g}" for x in v) if v else "-"
    print("\n# 2 048 robots per fleet; errors against the fp64 evaluation of the reference's formulae (C oracle, double build).")
    print("# A: |err| <= 1e-5 max(1, |qdd|)   B: backward error omega <= 2e-5 (oracle.ETA)   E: |err| <= 2 x fp32 envelope (17 fp32 oracle")
    print("# draws + the autograd fp32 restatement's own error); every clause counted ON ITS OWN.  ratio = err_engine / envelope over the robots")
    print("# outside A (percentiles 50/90/99/100); control = the plain fp32 oracle evaluation against an envelope of 16 OTHER draws.")
    print("# omega (B, omega max): against the EXACT system (M, f) of the fp64 evaluation; `fp32 oracle`: the fp32-leaf oracle's answer against the")
    print("# same system; `vs fp32 sys`: the engine against the fp32-leaf oracle's OWN system (which holds the 1e-7 / (1 - sigmoid) noise of the")
    print("# reference's `1. - tf.sigmoid(z)`; the kernels form that gate without the cancellation).")
    print(f"{'fleet':44s} {'A':>5s} {'B':>5s} {'E':>5s} {'gate':>5s} {'out-A':>5s} | {'ratio engine/envelope':>24s} {'>1':>4s} {'>2':>4s} | "
          f"{'control':>22s} {'>1':>4s} {'>2':>4s} | {'rel err p90/p99 engine':>24s} {'fp32 oracle':>20s} {'restatement out-A':>18s} {'omega max':>10s} {'fp32 oracle':>11s} {'vs fp32 sys':>11s}")
    for r in rows:
        n_e = r["robots"] - r["outside_A"] + (r["outside_A"] - r["E_ratio_gt_2"])
        print(f"{r['fleet'][:44]:44s} {r['A_vs_f64']:5d} {r['B_omega_le_2e5']:5d} {n_e:5d} {r['passes_gate']:5d} {r['outside_A']:5d} | "
              f"{f3(r['E_ratio_engine_over_envelope17_pcts']):>24s} {r['E_ratio_gt_1']:4d} {r['E_ratio_gt_2']:4d} | "
              f"{f3(r['control_ratio_c32_over_envelope16_pcts']):>22s} {r['control_gt_1']:4d} {r['control_gt_2']:4d} | "
              f"{f3(r['rel_err_all_robots_pcts_engine'][1:3]):>24s} {f3(r['rel_err_all_robots_pcts_c32'][1:3]):>20s} {r['ref32_outside_A']:18d} {r['omega_exact_max']:10.2e} {r['omega_exact_max_of_the_fp32_oracle']:11.2e} {r['omega_max_all']:11.2e}")


if __name__ == "__main__":
    main()
