"""Print a window of the kernel timeline of a rocprofv3 --kernel-trace run: name, queue, start / end relative to the
window's first kernel, gap to the previous kernel of the same name.  usage: trace_timeline.py <dir> [first] [count]"""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
t0 = int(rows[first]["Start_Timestamp"])
prev_end = {}
for r in rows[first:first + count]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"][:48]
    print(f"{s/1e3:9.2f} -> {e/1e3:9.2f} us  ({(e-s)/1e3:7.2f})  queue {r.get('Queue_Id','?'):>3s}  {name}")
