#!/bin/bash
# dispatch thresholds: hex vs quad (vs lane for config 2) at mid-size fleets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
for wl in config3 config2; do
for R in 4096 6144 8192 10240 12288 16384 20480 24576 32768; do
line="$wl R=$R"
for k in hex quad lane; do
if [ $wl = config3 ] && [ $k = lane ]; then continue; fi
RMP2_KERNEL=$k timeout -k 10 120 python bench.py --workload $wl --robots $R --no-cpu-baseline --no-secondary > $O/b.json 2>/dev/null
line="$line | $k $(python -c "import json;j=json.load(open('$O/b.json'));print('%.2f' % (j['ms_per_step']*1e3))")"
done
echo "$line"
done; done
