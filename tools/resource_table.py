"""Turn hipcc -Rpass-analysis=kernel-resource-usage output into one line per kernel.
usage: resource_table.py <remarks.txt> [name-filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize \[bytes/lane\]"), g("VGPR Spill"), g("SGPR Spill"), g("Occupancy \[waves/SIMD\]"), g("LDS Size \[bytes/block\]")))
print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scr':>6} {'vspill':>6} {'sspill':>6} {'occ':>4}  kernel")
for r in rows:
    if flt in r[0]:
        print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>6} {r[5]:>6} {r[6]:>6} {r[7]:>4}  {r[0][:150]}")
