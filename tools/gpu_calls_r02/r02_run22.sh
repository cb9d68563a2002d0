#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
for cfg in "config3 8192" "config3 16384" "config3 20480" "config2 16384" "config2 32768" "config2 40960"; do
set -- $cfg
timeout -k 10 120 python bench.py --workload $1 --robots $2 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('$1 $2 auto',j['ms_per_step'], j['roofline']['kernel'][:30])"
done
timeout -k 10 120 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null; python -c "import json;j=json.load(open('$O/bench_config5.json'));print('c5',j['ms_per_step'],j['value']/1e6,j['roofline']['kernel'][:40])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
