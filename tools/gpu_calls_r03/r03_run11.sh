#!/bin/bash
# round 3, GPU call 11: stamps of the plain four-wave build after the ds_read fix; explicit-pair A/B
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
RMP2_LIB=$GRAFT_REPO_ROOT/tools/diag/librmp2_stamps.so RMP2_KERNEL=quad python tools/stamps.py 65536 > $O/stamps.txt 2> $O/stamps.err; cat $O/stamps.txt
for R in 32768 65536; do timeout -k 10 300 python bench.py --workload config3b --no-cpu-baseline --steps 500 --robots $R > $O/b.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/b.json')); r=d['roofline']; print('config3b R $R', d['ms_per_step']*1e3, 'us; hbm frac', r['frac'])"; done
timeout -k 10 300 python bench.py --workload config5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null; cut -c1-250 $O/bench_config5.json
timeout -k 10 300 python bench.py --workload config4 --no-cpu-baseline --no-secondary > $O/bench_config4.json 2>/dev/null; cut -c1-250 $O/bench_config4.json
