"""Per-ablation-group medians of SQ counters from a rocprofv3 --pmc run of tools/phase_timing.py."""
import csv, glob, statistics, sys, collections
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if "rmp2_step" in r["Kernel_Name"]]
by_disp = collections.defaultdict(dict)
for r in rows:
    by_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by_disp)
n = 120
names = sorted({k for d in by_disp.values() for k in d})
print("group " + " ".join(f"{k[3:]:>16s}" for k in names))
for g in range(len(ids) // n):
    grp = ids[g * n + 20:(g + 1) * n]
    waves = statistics.median(by_disp[i].get("SQ_WAVES", 1) for i in grp)
    print(f"{g:5d} " + " ".join(f"{statistics.median(by_disp[i][k] for i in grp) / max(waves, 1):16.0f}" for k in names) + "   (per wave)")
