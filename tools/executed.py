"""profiles/executed_<workload>.json: what the control-step kernel EXECUTED per launch (bench.py prints it next to the
algorithmic roofline figures).  Inputs: rocprofv3 --pmc passes around `bench.py --steps 100 --no-cpu-baseline --no-secondary`
(SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS ... / GRBM_GUI_ACTIVE) and the stamps tool's output.
usage: executed.py <pmc_dir_insts> <pmc_dir_cycles> <stamps.txt> <workload> <robots> <out.json> [commit]"""
import csv, glob, json, os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def med(d):
    path = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = {}
    for r in csv.DictReader(open(path)):
        if "rmp2_step_quad" in r["Kernel_Name"] or "rmp2_step" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v) for k, v in vals.items()}

a, b = med(sys.argv[1]), med(sys.argv[2])
st = open(sys.argv[3]).read()
m = re.search(r"pair loop: ([\d.]+) trips per wave-step .*?([\d.]+) in-range pairs per robot-step", st)
waves = a["SQ_WAVES"]
valu_w = a["SQ_INSTS_VALU"] / waves
n_simd = 1024
busy = b.get("GRBM_GUI_ACTIVE")
out = {"workload": sys.argv[4], "robots": int(sys.argv[5]),
       "waves_per_launch": waves,
       "valu_insts_per_wave": valu_w,
       "salu_insts_per_wave": a.get("SQ_INSTS_SALU", 0) / waves,
       "lds_insts_per_wave": a.get("SQ_INSTS_LDS", 0) / waves,
       "vmem_insts_per_wave": (a.get("SQ_INSTS_VMEM_RD", 0) + a.get("SQ_INSTS_VMEM_WR", 0)) / waves if "SQ_INSTS_VMEM_RD" in a else None,
       "wave_lifetime_cycles": 4.0 * a["SQ_WAVE_CYCLES"] / waves,
       "issue_slot_occupancy": (a["SQ_INSTS_VALU"] / n_simd * 2.0) / (4.0 * a["SQ_WAVE_CYCLES"] / waves),
       "issue_slot_occupancy_note": "VALU instructions per SIMD x 2 cycles (wave64 fp32 issue) / mean wave lifetime in cycles "
                                    "(4 x SQ_WAVE_CYCLES / SQ_WAVES: the counter ticks every fourth cycle; all waves of the launch are "
                                    "co-resident at this fleet size, so a wave's lifetime is the kernel's duration)",
       "pair_trips_per_wave_step": float(m.group(1)) if m else None,
       "in_range_pairs_per_robot_step": float(m.group(2)) if m else None,
       "in_range_pair_fraction": float(m.group(2)) / 256.0 if m else None,
       "source": "rocprofv3 --pmc (two passes) around bench.py + tools/stamps.py (diagnostic -DRMP2_STAMPS build)",
       "kernel_src_hash": ge.kernel_src_hash(),
       "commit": sys.argv[7] if len(sys.argv) > 7 else None}
json.dump(out, open(sys.argv[6], "w"), indent=1)
print(json.dumps(out, indent=1))
