"""Summarise a rocprofv3 kernel-trace CSV produced around tools/phase_timing.py: the step
kernel's dispatches appear in groups of 120 per ablation (20 warm-up + 100 timed); print
the median duration of the last 100 of each group.  usage: trace_summary.py <dir> <labels...>"""
import csv, glob, statistics, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if "rmp2_step" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
n = 120
for gidx in range(len(dur) // n):
    grp = dur[gidx * n + 20:(gidx + 1) * n]
    r = rows[gidx * n + 20]
    print(f"group {gidx:2d}: median {statistics.median(grp):8.2f} us  min {min(grp):8.2f}  grid {r['Grid_Size_X']:>8s} vgpr {r['VGPR_Count']} scratch {r['Scratch_Size']}  {r['Kernel_Name'][:60]}")
