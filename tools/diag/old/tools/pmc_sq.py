"""Median per-dispatch SQ counters of the step kernel from a rocprofv3 --pmc run directory."""
import csv, glob, statistics, sys, collections
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
vals = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "rmp2_step" in r["Kernel_Name"]:
        vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:28s} median {statistics.median(v):14.0f}  (n={len(v)})")
