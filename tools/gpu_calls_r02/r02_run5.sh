#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py tests/test_gpu_capsules.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -5 $O/pytest_gpu_quad.txt
timeout -k 10 120 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3.json'));print('c3 65536 split',j['ms_per_step'],j['roofline']['kernel_ms'],j['roofline']['valu']['frac'],'| c2',j['secondary']['ms_per_step'])"
RMP2_QUAD_SPLIT=0 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary > $O/bench_c3_nosplit.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_nosplit.json'));print('c3 65536 whole',j['ms_per_step'])"
for R in 32768 131072; do timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/bench_c3_$R.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_$R.json'));print('c3 $R split',j['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline --no-secondary > /dev/null 2>&1
find $O/kt3 -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-60,200-400
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 bench.py --steps 100 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python - <<'PY'
import csv, glob, statistics, collections
path = glob.glob("gpurun_out/r02e/a/**/*counter_collection.csv", recursive=True)[0]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if "rmp2_step" in r["Kernel_Name"]:
        part = "PART1" if "false, 1>" in r["Kernel_Name"] else ("PART2" if "false, 2>" in r["Kernel_Name"] else "other")
        vals[part][r["Counter_Name"]].append(float(r["Counter_Value"]))
for part in vals:
    for k, v in sorted(vals[part].items()):
        print(f"{part} {k:24s} median {statistics.median(v):14.0f} (n={len(v)})")
PY
rm -rf $O/a $O/kt3/*/*.db
