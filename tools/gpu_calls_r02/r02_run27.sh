#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02y; mkdir -p $O
RMP2_KERNEL=quad timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py tests/test_gpu_random_robots.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -2
for R in 24576 32768 40960 49152 57344 65536 81920 98304 131072 196608 262144; do
line="c3 R=$R b=$(python -c "print($R/16384.)")"
for w in 2 3 4; do
RMP2_QUAD_MINW=$w timeout -k 10 120 python bench.py --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null
line="$line | W$w $(python -c "import json;j=json.load(open('$O/b.json'));print('%.2f' % (j['ms_per_step']*1e3))")"
done
echo "$line"
done
