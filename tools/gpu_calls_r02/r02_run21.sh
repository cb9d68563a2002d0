#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
for wl in config2; do
for R in 9216 40960 49152 65536 81920 98304 131072 196608 262144; do
line="$wl R=$R"
for k in quad lane; do
RMP2_KERNEL=$k timeout -k 10 120 python bench.py --workload $wl --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null
line="$line | $k $(python -c "import json;j=json.load(open('$O/b.json'));print('%.2f' % (j['ms_per_step']*1e3))")"
done
echo "$line"
done; done
for R in 9216; do
line="config3 R=$R"
for k in hex quad; do
RMP2_KERNEL=$k timeout -k 10 120 python bench.py --workload config3 --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null
line="$line | $k $(python -c "import json;j=json.load(open('$O/b.json'));print('%.2f' % (j['ms_per_step']*1e3))")"
done
echo "$line"
done
RMP2_KERNEL=hex timeout -k 10 120 python bench.py --workload config2 --robots 9216 --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null; python -c "import json;j=json.load(open('$O/b.json'));print('config2 9216 hex %.2f' % (j['ms_per_step']*1e3))"
