"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) around bench.py into
profiles/traffic_<workload>.json (HBM bytes per launch of the control-step kernel).
usage: pmc_traffic.py <fetch_dir> <write_dir> <workload> <robots> <out.json> [commit]
(the GPU box has no .git: pass the commit the snapshot was taken at, or fill it in afterwards)"""
import csv, glob, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
def per_launch(d, counter):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if r["Counter_Name"] == counter and "rmp2_step" in r["Kernel_Name"]]
    return statistics.median(vals), len(vals)
f, nf = per_launch(sys.argv[1], "FETCH_SIZE")
w, nw = per_launch(sys.argv[2], "WRITE_SIZE")
robots = int(sys.argv[4])
# MI355X_MICROARCH.md section HBM: counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads
# HALF the bytes of a coalesced stream -- measured for 16 B per lane there and for THIS engine's 4 B per lane (and
# WRITE_SIZE exact for both) by tools/fetch_calib.hip on a known 512 MiB: profiles/r02_fetch_write_calibration.txt.
# Correction applied: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
out = {"workload": sys.argv[3], "robots": robots, "launches": min(nf, nw),
       "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
       "hbm_bytes_per_launch": (2 * f + w) * 1024.0,
       "hbm_bytes_per_launch_uncorrected": (f + w) * 1024.0,
       "algorithmic_bytes_per_launch": (6264 if sys.argv[3] == "config3b" else 120) * robots,
       "kernel_src_hash": ge.kernel_src_hash(),
       "commit": (sys.argv[6] if len(sys.argv) > 6 else
                  subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None),
       "note": "FETCH_SIZE/WRITE_SIZE from separate --pmc passes, gfx950 correction FETCH x2 (calibrated on this access "
               "pattern); memory-side (fabric) requests, Infinity-Cache hits included"}
if out["hbm_bytes_per_launch"] > 2 * out["algorithmic_bytes_per_launch"]:
    # the 128-register (four waves per SIMD) build of the quad kernel spills: its private-segment traffic is what shows here
    out["note"] += ("; above the algorithmic bytes: register-spill (scratch) traffic of the register-capped kernel build that "
                    "ran -- DESIGN.md section 5 gives the per-wave figure and the time it buys")
json.dump(out, open(sys.argv[5], "w"), indent=1)
print(out)
