#!/bin/bash
# round 2, GPU call 2: quad kernel after the LDS / register diet (4 waves per SIMD) -- parity, then A/B against MINW=2
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest_gpu_quad.txt 2>&1; tail -5 $O/pytest_gpu_quad.txt
for R in 36864 65536 73728; do
  timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/bench_c3_$R.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_$R.json'));print($R,'minw4',j['ms_per_step'],j['roofline']['kernel_ms'])"
  RMP2_QUAD_MINW=2 timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/bench_c3_${R}_minw2.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_c3_${R}_minw2.json'));print($R,'minw2',j['ms_per_step'],j['roofline']['kernel_ms'])"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt3 -- python3 bench.py --steps 200 --no-cpu-baseline --no-secondary > /dev/null 2>&1
find $O/kt3 -name "*kernel_stats.csv" -exec head -3 {} \;
find $O/kt3 -name "*kernel_trace.csv" -exec python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'quad' in r['Kernel_Name']]
r=rows[-1]; print({k:r[k] for k in r if k in ('VGPR_Count','Accum_VGPR_Count','SGPR_Count','LDS_Block_Size','Scratch_Size','Grid_Size_X','Workgroup_Size_X')})" {} \;
rm -rf $O/kt3/*/*.db; du -sh $O
