"""Per-basic-block instruction statistics of one kernel from hipcc's assembly: where the VALU work, the transcendentals,
the DPP moves and the register spills (scratch_load / scratch_store) sit.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o /tmp/k.s <file.hip>
  python tools/asm_blocks.py /tmp/k.s <mangled-name-prefix> [--all]"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
show_all = "--all" in sys.argv
lines, on = [], False
for l in open(path):
    if l.startswith(prefix):
        on = True
    if on:
        lines.append(l.rstrip("\n"))
        if "s_endpgm" in l:
            break
blocks, cur = [], ("entry", [])
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
    else:
        cur[1].append(l)
blocks.append(cur)
tot = dict(n=0, valu=0, trans=0, f64=0, dpp=0, sld=0, sst=0)
for name, b in blocks:
    ins = [x.strip().split()[0] for x in b if x.startswith("\t") and not x.strip().startswith((".", ";"))]
    if not ins:
        continue
    st = dict(n=len(ins), valu=sum(i.startswith("v_") for i in ins),
              trans=sum(i.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")) for i in ins),
              f64=sum("f64" in i for i in ins), dpp=sum(("quad_perm" in x or "row_" in x) for x in b),
              sld=sum(i.startswith("scratch_load") for i in ins), sst=sum(i.startswith("scratch_store") for i in ins))
    for k in tot:
        tot[k] += st[k]
    br = [x.strip() for x in b if "s_cbranch" in x or "s_branch" in x]
    if show_all or st["sld"] + st["sst"] > 0 or st["n"] > 60:
        print(f"{name:12s} " + " ".join(f"{k}={v:4d}" for k, v in st.items()) + f"  {br[-1] if br else ''}")
print("TOTAL (static) " + " ".join(f"{k}={v}" for k, v in tot.items()))
