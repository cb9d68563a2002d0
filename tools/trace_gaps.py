import csv, glob, statistics, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
q = [r for r in rows if "rmp2_step_quad" in r["Kernel_Name"]]
other = collections = {}
for r in rows:
    if "rmp2_step" not in r["Kernel_Name"]:
        other[r["Kernel_Name"][:60]] = other.get(r["Kernel_Name"][:60], 0) + 1
q = q[len(q)//2:]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in q]
gap = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(q, q[1:])]
per = [(int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a, b in zip(q, q[1:])]
print(sys.argv[1], "n", len(q), "kernel dur median %.1f us" % statistics.median(dur), "gap median %.1f" % statistics.median(gap), "period median %.1f" % statistics.median(per))
print({k: v for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:6]})
